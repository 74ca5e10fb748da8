// parameters.hpp -- facade header carrying the reference's name.  The caller (apps/counter.cpp) constructs
// Parameters(k, m, b) and reads k, m, b, m_reduc, allocated_bytes, compacted_size, mask_large_minimizer and dede
// (brisk/parameters.hpp:11-18), so those names and types are the contract; everything else is this repository's.
#ifndef BRISK_AMD_PARAMETERS_HPP
#define BRISK_AMD_PARAMETERS_HPP
#include <cstdint>

#include "Decycling.h"

typedef unsigned int uint;

// Derived sizes of a (k, m, b) triple, in nucleotides unless said otherwise.
struct BriskGeometry {
    static uint8_t minimizer_nts_outside_bucket(uint8_t m, uint8_t b) { return (uint8_t)(m - b); }  // parameters.hpp:27
    static uint8_t compacted_kmer_nts(uint8_t k, uint8_t b) { return (uint8_t)(k - b); }             // parameters.hpp:30
    // bytes of a compacted super-k-mer of maximal length: 2k - m - b nucleotides, four to a byte, rounded up (:31)
    static uint superkmer_bytes(uint8_t k, uint8_t m, uint8_t b) { return (uint)((2u * k - m - b + 3u) / 4u); }
    static uint64_t low_bits(unsigned n) { return n >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1); }
};

class Parameters {
  public:
    uint8_t k, m, b;                // k-mer size; minimizer size (odd, < k); bucket order of magnitude in [1, m]
    uint8_t m_reduc;                // minimizer nucleotides that are not part of the bucket id
    uint allocated_bytes;           // see BriskGeometry::superkmer_bytes
    uint8_t compacted_size;         // nucleotides of a k-mer once its bucket nucleotides are removed
    uint64_t mask_large_minimizer;  // 2m ones
    DecyclingSet* dede;             // owned by nobody, as in the reference (brisk/parameters.hpp:32-33)

    // An invalid triple (b > m underflows m_reduc in the reference and crashes there) is reported when the index is
    // built: Brisk's constructor throws std::invalid_argument.
    Parameters(uint8_t k_, uint8_t m_, uint8_t b_)
        : k(k_),
          m(m_),
          b(b_),
          m_reduc(BriskGeometry::minimizer_nts_outside_bucket(m_, b_)),
          allocated_bytes(BriskGeometry::superkmer_bytes(k_, m_, b_)),
          compacted_size(BriskGeometry::compacted_kmer_nts(k_, b_)),
          mask_large_minimizer(BriskGeometry::low_bits(2u * m_)),
          dede(new DecyclingSet(m_ ? m_ : 1)) {}
};

#endif
