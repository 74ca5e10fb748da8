// Decycling.h -- facade header with the reference's name (brisk/Decycling.h:14-27).
// The coefficient table is built on the HOST with libm exactly as the reference builds it
// (brisk/Decycling.cpp:7-13) and handed to the device as bits; classification itself runs on
// the GPU (there is no host implementation of the path in this repo's product code).
#ifndef BRISK_AMD_DECYCLING_H
#define BRISK_AMD_DECYCLING_H
#define _USE_MATH_DEFINES
#include <cmath>
#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "brisk_hip.h"

class DecyclingSet {
  public:
    explicit DecyclingSet(unsigned m) : m_(m), coef_(4 * m, 0.0) {
        const double unit = 2 * M_PI / m;
        for (unsigned j = 1; j < m; j++) {
            coef_[4 * j + 1] = std::sin(unit * j);
            coef_[4 * j + 2] = 2 * coef_[4 * j + 1];
            coef_[4 * j + 3] = 3 * coef_[4 * j + 1];
        }
    }
    ~DecyclingSet() {
        for (auto& kv : scan_) brisk_hip_destroy(kv.second);
    }
    DecyclingSet(const DecyclingSet&) = delete;
    DecyclingSet& operator=(const DecyclingSet&) = delete;

    unsigned m() const { return m_; }
    const double* coef() const { return coef_.data(); }

    // bfc_hash_64 of an m-mer (brisk/hashing.cpp:8-19): class << 62 | mixer, evaluated on the device
    uint64_t order_key(uint64_t mmer) {
        std::lock_guard<std::recursive_mutex> g(mu_);
        uint64_t key = 0;
        if (brisk_hip_debug_order_keys(scan_handle(m_ < 62 ? m_ + 2 : 63), &mmer, 1, /*exact=*/1, &key) != BRISK_HIP_OK)
            throw std::runtime_error("brisk_hip_debug_order_keys failed");
        return key;
    }
    // class 0/1/2 of an m-mer (brisk/Decycling.cpp:38-52)
    unsigned memDouble(uint64_t seq) { return (unsigned)(order_key(seq) >> 62); }
    bool mem(uint64_t seq) { return memDouble(seq) == 0; }  // brisk/Decycling.cpp:28-34

    // a scan-only index for SuperKmerEnumerator(k, m): no bucket is ever filled.
    // Calls on the returned handle must hold mutex().
    brisk_hip_index* scan_handle(unsigned k) {
        std::lock_guard<std::recursive_mutex> g(mu_);
        auto it = scan_.find(k);
        if (it != scan_.end()) return it->second;
        brisk_hip_options o{};
        o.struct_size = sizeof o;
        o.part_bits = 2;  // the library grows it to the smallest directory whose entry key fits
        o.entry_ids = 1;
        brisk_hip_index* h = nullptr;
        const unsigned b = m_ < 14 ? m_ : 14;
        const int rc = brisk_hip_create(&h, (uint8_t)k, (uint8_t)m_, (uint8_t)b, 1, coef_.data(), &o);
        if (rc != BRISK_HIP_OK)
            throw std::invalid_argument("brisk_hip_create(k=" + std::to_string(k) + ", m=" + std::to_string(m_) + ") failed with status " +
                                        std::to_string(rc) + " (bad parameters or no gfx950 device)");
        scan_[k] = h;
        return h;
    }
    std::recursive_mutex& mutex() { return mu_; }

  private:
    const unsigned m_;
    std::vector<double> coef_;
    std::map<unsigned, brisk_hip_index*> scan_;
    std::recursive_mutex mu_;
};

#endif
