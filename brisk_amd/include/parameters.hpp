// parameters.hpp -- facade header with the reference's name (brisk/parameters.hpp:9-35).
#ifndef BRISK_AMD_PARAMETERS_HPP
#define BRISK_AMD_PARAMETERS_HPP
#include <cmath>
#include <cstdint>

#include "Decycling.h"

typedef unsigned int uint;

class Parameters {
  public:
    uint8_t k;
    uint8_t m;
    uint8_t b;
    uint8_t m_reduc;
    uint allocated_bytes;
    uint8_t compacted_size;
    uint64_t mask_large_minimizer;
    DecyclingSet* dede;
    /** @param k k-mer size, @param m minimizer size (odd, < k), @param b bucket order of magnitude in [1, m].
     *  The derived fields are the reference's (parameters.hpp:24-34).  Unlike the reference, an invalid
     *  triple (e.g. b > m, which underflows m_reduc there and segfaults) is reported when the index is
     *  built: Brisk's constructor throws std::invalid_argument. */
    Parameters(uint8_t k, uint8_t m, uint8_t b) {
        this->k = k;
        this->m = m;
        this->b = b;
        this->m_reduc = m - b;
        this->mask_large_minimizer = m >= 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * m)) - 1);
        this->compacted_size = k - b;
        this->allocated_bytes = (uint)std::ceil(((double)(2 * k - m - b)) / 4);
        this->dede = new DecyclingSet(m ? m : 1);  // owned by nobody, as in the reference (parameters.hpp:32-33)
    }
};

#endif
