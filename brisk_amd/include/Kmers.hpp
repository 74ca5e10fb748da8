// Kmers.hpp -- facade header with the reference's name (brisk/Kmers.hpp:26-136): kint, kmer_full,
// SuperKmerEnumerator and the k-mer <-> string helpers.  SuperKmerEnumerator runs on the GPU:
// the first next() scans the whole sequence through brisk_hip_scan_sequence and the following
// calls hand the vectors out one by one.
#ifndef BRISK_AMD_KMERS_HPP
#define BRISK_AMD_KMERS_HPP
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hashing.hpp"
#include "parameters.hpp"

typedef __uint128_t kint;
typedef __uint128_t skint;

class kmer_full {
  public:
    kint kmer_s;
    kint minimizer;
    uint8_t minimizer_idx;
    std::vector<int8_t> interleaved;
    DecyclingSet* dede;
    // brisk/Kmers.cpp:17-25
    kmer_full(kint value, uint8_t minimizer_idx, uint8_t minimizer_size, DecyclingSet* dede)
        : kmer_s(value), minimizer((value >> (2 * minimizer_idx)) & (((kint)1 << (2 * minimizer_size)) - 1)), minimizer_idx(minimizer_idx), dede(dede) {}
    kmer_full() : kmer_s(0), minimizer(0), minimizer_idx(0), dede(nullptr) {}
    kmer_full(kmer_full&&) = default;
    kmer_full(const kmer_full&) = default;
    kmer_full& operator=(kmer_full&&) = default;
    kmer_full& operator=(const kmer_full&) = default;
    void copy(const kmer_full& kmer) { *this = kmer; }  // brisk/Kmers.cpp:53-59
    void compute_mini(uint8_t mini_size) {              // brisk/Kmers.cpp:74-77
        minimizer = (kmer_s >> (2 * minimizer_idx)) & (((kint)1 << (2 * mini_size)) - 1);
    }
    uint8_t suffix_size() const { return minimizer_idx; }
    uint8_t prefix_size(const uint8_t k, const uint8_t m) const { return k - m - minimizer_idx; }
};

class SuperKmerEnumerator {
  public:
    SuperKmerEnumerator(std::string& s, const uint8_t k, const uint8_t m, DecyclingSet* dede)
        : seq(s), dede(dede), k(k), m(m), scanned_(false), cursor_(0), kmer_at_(0) {
        if (dede == nullptr || dede->m() != m) throw std::invalid_argument("SuperKmerEnumerator: dede does not match m");
    }
    // Appends the next vector of k-mers that share a minimizer and returns that minimizer's value;
    // an untouched `kmers` (size unchanged) means the sequence is exhausted (brisk/Kmers.cpp:522-603).
    kint next(std::vector<kmer_full>& kmers) {
        if (!scanned_) scan();
        if (cursor_ >= ret_.size()) return (kint)0;
        const uint32_t n = n_[cursor_];
        const uint64_t ret = ret_[cursor_];
        for (uint32_t j = 0; j < n; j++, kmer_at_++) {
            kmer_full km(((kint)hi_[kmer_at_] << 64) | lo_[kmer_at_], idx_[kmer_at_], m, dede);
            km.minimizer = ret;  // brisk/Kmers.cpp:579-583
            kmers.push_back(std::move(km));
        }
        cursor_++;
        return (kint)ret;
    }
    std::string& seq;
    DecyclingSet* dede;
    uint8_t k;
    uint8_t m;

  private:
    void scan() {
        scanned_ = true;
        if (seq.size() < k) return;
        const uint64_t nk = seq.size() - k + 1;
        ret_.resize(nk);
        n_.resize(nk);
        lo_.resize(nk);
        hi_.resize(nk);
        idx_.resize(nk);
        uint64_t n_skm = 0;
        std::lock_guard<std::recursive_mutex> g(dede->mutex());
        brisk_hip_index* h = dede->scan_handle(k);
        const int rc = brisk_hip_scan_sequence(h, seq.data(), seq.size(), nk, ret_.data(), n_.data(), lo_.data(), hi_.data(), idx_.data(), &n_skm);
        if (rc != BRISK_HIP_OK) throw std::runtime_error(std::string("brisk_hip_scan_sequence: ") + brisk_hip_last_error(h));
        ret_.resize(n_skm);
        n_.resize(n_skm);
    }
    bool scanned_;
    size_t cursor_, kmer_at_;
    std::vector<uint64_t> ret_, lo_, hi_;
    std::vector<uint32_t> n_;
    std::vector<uint8_t> idx_;
};

// ----- k-mer <-> text (brisk/Kmers.hpp:94-125, brisk/Kmers.cpp:218-253); encoding A0 C1 T2 G3 -----
template <typename T>
void print_kmer(T num, uint8_t n) {
    static const char letters[4] = {'A', 'C', 'T', 'G'};
    for (int i = n - 1; i >= 0; i--) std::cerr << letters[(unsigned)((num >> (2 * i)) & 3)];
    std::cerr << std::endl;
}
inline std::string kmer2str(kint num, unsigned k) {
    static const char letters[4] = {'A', 'C', 'T', 'G'};
    std::string res;
    for (int i = (int)k - 1; i >= 0; i--) res += letters[(unsigned)((num >> (2 * i)) & 3)];
    return res;
}
inline kint str2num(const std::string& str) {
    kint res = 0;
    for (char c : str) res = (res << 2) + ((c >> 1) & 3);
    return res;
}

#endif
