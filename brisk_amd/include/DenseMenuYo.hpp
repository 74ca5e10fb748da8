// DenseMenuYo.hpp -- facade header with the reference's name.  `Brisk<DATA>::menu` is an opaque
// owner of the device index handle (the reference's 4^b directory, lock stripes and bucket matrix,
// brisk/DenseMenuYo.hpp:32-99, live on the GPU).
#ifndef BRISK_AMD_DENSEMENUYO_HPP
#define BRISK_AMD_DENSEMENUYO_HPP
#include <stdexcept>
#include <string>

#include "brisk_hip.h"
#include "parameters.hpp"

template <class DATA>
class DenseMenuYo {
  public:
    brisk_hip_index* handle;
    Parameters params;
    explicit DenseMenuYo(Parameters& parameters) : handle(nullptr), params(parameters) {
        brisk_hip_options o{};
        o.struct_size = sizeof o;
        o.entry_ids = 1;  // DATA lives on the host, indexed by the entry ids the device assigns
        const int rc = brisk_hip_create(&handle, params.k, params.m, params.b, (uint32_t)sizeof(DATA), params.dede->coef(), &o);
        if (rc == BRISK_HIP_EINVAL)
            throw std::invalid_argument("Brisk: invalid Parameters(k=" + std::to_string(params.k) + ", m=" + std::to_string(params.m) +
                                        ", b=" + std::to_string(params.b) + "): need 1 <= b <= m < k <= 63, m odd");
        if (rc != BRISK_HIP_OK) throw std::runtime_error("Brisk: brisk_hip_create failed with status " + std::to_string(rc));
    }
    ~DenseMenuYo() {
        if (handle) brisk_hip_destroy(handle);
    }
    DenseMenuYo(const DenseMenuYo&) = delete;
    DenseMenuYo& operator=(const DenseMenuYo&) = delete;
};

#endif
