// buckets.hpp -- facade header with the reference's name.  The reference's Bucket<DATA>/SKL storage
// engine (brisk/buckets.hpp, brisk/SuperKmerLight.hpp) is replaced by the device index behind
// include/brisk_hip.h; apps/counter.cpp includes this header but uses nothing
// from it.
#ifndef BRISK_AMD_BUCKETS_HPP
#define BRISK_AMD_BUCKETS_HPP
#include "Kmers.hpp"
#include "parameters.hpp"
#endif
