// hashing.hpp -- facade header with the reference's name (brisk/hashing.hpp:9-10).
#ifndef BRISK_AMD_HASHING_HPP
#define BRISK_AMD_HASHING_HPP
#include <cstdint>

#include "Decycling.h"

// order key of an m-mer (brisk/hashing.cpp:8-19), evaluated on the device.  `mask` (2m ones) is
// implied by dede.  bfc_hash_64_inv (hashing.cpp:23-48) has no caller outside the index and is
// not part of this facade: the device applies it when it hands k-mers back (Brisk::next).
inline uint64_t bfc_hash_64(uint64_t key, uint64_t mask, DecyclingSet* dede) {
    (void)mask;
    return dede->order_key(key);
}

#endif
