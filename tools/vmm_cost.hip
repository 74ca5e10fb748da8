// What device memory costs to get (GPU box): hipMemCreate / hipMemMap / hipMemSetAccess timed for pieces of P GiB up to T GiB.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_cost tools/vmm_cost.hip && /tmp/vmm_cost P T [1: access set from the base, 0: per piece]
// Round 3, one MI355X box: hipMemCreate ~30 ms per GiB whatever the piece size (96 GiB: 2.9-3.4 s), hipMemMap 18-150 ms, hipMemSetAccess < 2 ms in all.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t piece = (size_t)(argc > 1 ? atoi(argv[1]) : 8) << 30, total = (size_t)(argc > 2 ? atoi(argv[2]) : 96) << 30;
    const int whole_range = argc > 3 ? atoi(argv[3]) : 1;
    hipSetDevice(0);
    hipFree(0);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* base = nullptr;
    double t0 = now();
    if (hipMemAddressReserve(&base, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    printf("reserve %.1f ms\n", (now() - t0) * 1e3);
    size_t mapped = 0;
    double tc = 0, tm = 0, ta = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    while (mapped < total) {
        hipMemGenericAllocationHandle_t hd;
        double a = now();
        if (hipMemCreate(&hd, piece, &prop, 0) != hipSuccess) { printf("create failed at %zu GiB\n", mapped >> 30); break; }
        double b = now();
        if (hipMemMap((char*)base + mapped, piece, 0, hd, 0) != hipSuccess) { printf("map failed\n"); break; }
        double c = now();
        hipError_t e = whole_range ? hipMemSetAccess(base, mapped + piece, &acc, 1) : hipMemSetAccess((char*)base + mapped, piece, &acc, 1);
        double d = now();
        if (e != hipSuccess) { printf("setaccess failed: %s\n", hipGetErrorString(e)); break; }
        tc += b - a; tm += c - b; ta += d - c;
        hs.push_back(hd);
        mapped += piece;
    }
    printf("piece %zu GiB, %zu GiB mapped (%s): create %.1f ms, map %.1f ms, setaccess %.1f ms\n", piece >> 30, mapped >> 30, whole_range ? "access from base" : "access per piece", tc * 1e3, tm * 1e3, ta * 1e3);
    double t1 = now();
    hipMemset(base, 1, mapped);
    hipDeviceSynchronize();
    printf("first memset of it all: %.1f ms\n", (now() - t1) * 1e3);
    return 0;
}
