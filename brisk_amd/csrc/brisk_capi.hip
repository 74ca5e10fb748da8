// brisk_capi.hip -- host side of libbrisk_hip.so: the C-ABI of include/brisk_hip.h.
// Memory management, batching and kernel sequencing; all arithmetic of the path
// is in brisk_kernels.hip.  There is deliberately NO CPU fallback: without a
// gfx950 device brisk_hip_create fails with BRISK_HIP_ENODEVICE.
#include "brisk_kernels.hip"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/brisk_hip.h"

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#define BRISK_HOST_AVX2 1
#endif

namespace {

enum Slot { S_PACK, S_SYNTH, S_COUNT, S_SCAN, S_HIST, S_PSUM, S_TOUCHED, S_SCATTER, S_INSERT, S_QUERY, S_ENUM, S_LOOKUP, S_UPLOAD, S_NSLOTS };
// (the last slot is host wall time, not a kernel: a host batch's bytes from the caller's memory to the packed stream on the device)
const char* const kSlotNames[S_NSLOTS] = {"k_pack_ascii", "k_synth", "k_count_kmers", "k_scan", "k_part_hist", "k_psum", "k_touched_need",
                                          "k_scatter", "k_insert", "k_query", "k_enumerate", "k_lookup", "host_upload_and_pack_wall"};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// A device buffer that grows in place: a large virtual range is reserved once and physical
// memory is mapped behind it on demand (HIP virtual memory management), so growing the k-mer
// arena never copies it and never holds two generations at once.
struct VmBuf {
    char* base = nullptr;
    size_t reserved = 0, mapped = 0, gran = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};

constexpr size_t kPinBytes = 8192;  // 255 k-mers x 22 bytes + the tail words of the per-call API's staging
struct PendingEvent {
    int slot;
    hipEvent_t a, b;
};

}  // namespace

struct brisk_hip_index {
    BriskParams P{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::recursive_mutex call_mu;  // one call at a time per handle: the batch-granular lock of include/brisk_hip.h, "Threads"
    u64 n_parts = 0, n_buckets = 0;
    u64 max_batch_reads = 0;

    double* d_coef = nullptr;
    double* d_tabs = nullptr;   // coef[128] + packed fixed-point decycling chunk tables (u64 bits), staged to LDS by k_scan2
    ScanCfg scfg{};
    u32 insert_waves = 4096;    // most persistent k_insert waves any instantiation keeps resident (<= INSERT_SLOTS): sizes the arena reserve
    u32 scan_waves = 0;         // waves per k_scan2 block
    size_t scan_lds = 0;
    bool scan_v1 = false;       // BRISK_SCAN_V1=1: the plain restatement kernel
    bool entry_ids = false;
    bool use_vmm = false;
    bool scan_hist_valid = false;  // d_hist holds the per-partition histogram of the last brisk_hip_scan_packed
    u64 arena_limit = 0;        // BRISK_ARENA_LIMIT (entries): artificial ceiling, for tests of the out-of-memory path
    VmBuf vm_keys, vm_counts, vm_ids;
    unsigned long long* d_id_counter = nullptr;
    DevBuf seq_buf;
    // the index
    u64 arena_cap = 0;
    u64 arena_used_host = 0;
    IndexDev ix{};
    // scratch
    DevBuf route_buf;
    DevBuf bins;  // binned layout: n_parts bins of bin_cap records
    DevBuf huge;  // [0] number, [1..HUGE_LIST_CAP] descriptor indices of the batch's huge partitions (k_insert_huge)
    // Small insert batches are scanned at once and inserted later (flush_pending): their records collect here, their per-partition
    // counts in d_hist, until there are enough of them for the insert to work at its density, or a call needs the index.
    DevBuf pend;
    u64 n_pend = 0;
    bool pend_hist_ok = true;   // d_hist counts exactly the pending records
    bool defer = true;          // brisk_hip_options.immediate_inserts == 0 and BRISK_DEFER != 0
    bool verify = false;        // BRISK_VERIFY=1 at create: every stage hand-over of the host paths is checked (verify_upload, verify_records)
    bool trace = false;         // BRISK_TRACE=1 at create: one stderr line per batch saying which host path and which insert kernel took it
    DevBuf staging, parted, desc, chunk_buf, tags_a, tags_b, packed_tmp, bases_tmp, starts_tmp, sums_tmp, enum_out, lookup_buf;
    DevBuf packed_tmp2, starts_tmp2;  // the second set of insert_reads_pipelined: one sub-batch is scanned while the next one arrives
    u32* d_ovf_cnt = nullptr;              // OVF_REGIONS counters of the binned scan's overflow area
    std::vector<u64> owner_cut;            // sharded index: owner o holds partitions [owner_cut[o], owner_cut[o + 1]) (n_owners + 1 entries)
    u32* d_owner_cut = nullptr;            // the same on the device once brisk_hip_set_owner_cuts has replaced the equal ranges (else null)
    unsigned long long* d_hist = nullptr;  // n_parts + 1
    u32* d_off = nullptr;                  // n_parts + 1
    u32* d_cur32 = nullptr;                // n_parts
    u32* d_touched = nullptr;              // n_parts
    u32* d_block_sums = nullptr;
    u32 n_scan_blocks = 0;
    unsigned long long* d_small = nullptr;  // [0] n_rec [1] overflow(u32) [2] n_touched(u32) [3] need [4] kmers bound
    unsigned long long* h_small = nullptr;  // pinned mirror
    char* h_pin = nullptr;                  // pinned staging of the per-call API (one vector's k-mers in, ids out: one copy each way)
    u64 nb_skmers = 0;
    std::vector<u32> h_dir_cnt;  // enumeration snapshot
    bool dir_snapshot_valid = false;
    // profiling
    bool profiling = false;
    std::vector<PendingEvent> pending;
    uint64_t prof_launches[S_NSLOTS] = {};
    double prof_ms[S_NSLOTS] = {};
};

namespace {

#define HIPCHK(h, call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
            return e_ == hipErrorOutOfMemory ? BRISK_HIP_ENOMEM : BRISK_HIP_EHIP;                       \
        }                                                                                               \
    } while (0)

int fail(brisk_hip_index* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

bool pool_drain();

int ensure(brisk_hip_index* h, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return BRISK_HIP_OK;
    if (b.p) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e == hipErrorOutOfMemory && pool_drain()) {  // the arenas of destroyed indexes may be what is in the way
        (void)hipGetLastError();
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the failed allocation must not surface later as some kernel's launch error
        b.p = nullptr;
        return fail(h, e == hipErrorOutOfMemory ? BRISK_HIP_ENOMEM : BRISK_HIP_EHIP, std::string("device allocation of ") + std::to_string(want >> 20) + " MiB: " + hipGetErrorString(e));
    }
    b.bytes = want;
    return BRISK_HIP_OK;
}

struct ProfScope {
    brisk_hip_index* h;
    int slot;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(brisk_hip_index* h_, int slot_) : h(h_), slot(slot_) {
        if (h->profiling) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, h->stream);
        }
    }
    ~ProfScope() {
        if (a) {
            hipEventRecord(b, h->stream);
            h->pending.push_back({slot, a, b});
        }
    }
};

inline u32 nblocks(u64 n, u32 per) { return (u32)((n + per - 1) / per); }

int launch_check(brisk_hip_index* h, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, BRISK_HIP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    static const bool sync_launches = getenv("BRISK_SYNC_LAUNCHES") != nullptr;  // debugging: name the kernel a device fault belongs to
    if (sync_launches) {
        fprintf(stderr, "[brisk_hip] %s ...", what);
        e = hipStreamSynchronize(h->stream);
        fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
        if (e != hipSuccess) return fail(h, BRISK_HIP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    return BRISK_HIP_OK;
}

int vm_reserve(brisk_hip_index* h, VmBuf& b, size_t bytes) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    if (hipMemGetAllocationGranularity(&b.gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || b.gran == 0) return BRISK_HIP_EHIP;
    const size_t step = std::max<size_t>(b.gran, (size_t)1 << 30);  // map in >= 1 GiB pieces
    b.gran = (step + b.gran - 1) / b.gran * b.gran;
    bytes = (bytes + b.gran - 1) / b.gran * b.gran;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, bytes, 0, nullptr, 0) != hipSuccess) return BRISK_HIP_EHIP;
    b.base = (char*)p;
    b.reserved = bytes;
    return BRISK_HIP_OK;
}
int vm_grow(brisk_hip_index* h, VmBuf& b, size_t bytes) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    while (b.mapped < bytes) {
        // one physical allocation per piece of at most 8 GiB
        size_t add = (bytes - b.mapped + b.gran - 1) / b.gran * b.gran;
        const size_t piece_max = std::max<size_t>(b.gran, (size_t)8 << 30) / b.gran * b.gran;
        add = std::min(add, piece_max);
        if (b.mapped + add > b.reserved) return fail(h, BRISK_HIP_ENOMEM, "arena: virtual reservation exhausted");
        hipMemGenericAllocationHandle_t hd;
        hipError_t e = hipMemCreate(&hd, add, &prop, 0);
        if (e != hipSuccess && pool_drain()) {  // the arenas of destroyed indexes may be what is in the way
            (void)hipGetLastError();
            e = hipMemCreate(&hd, add, &prop, 0);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, BRISK_HIP_ENOMEM, std::string("arena growth: hipMemCreate: ") + hipGetErrorString(e));
        }
        e = hipMemMap(b.base + b.mapped, add, 0, hd, 0);
        if (e != hipSuccess) {
            hipMemRelease(hd);
            return fail(h, BRISK_HIP_EHIP, std::string("arena growth: hipMemMap: ") + hipGetErrorString(e));
        }
        // ROCm 7.2 rejects hipMemSetAccess on some sub-ranges that start past earlier mappings
        // ("invalid argument") but accepts the whole mapped range: always set access from the base
        e = hipMemSetAccess(b.base, b.mapped + add, &acc, 1);
        if (e != hipSuccess) {
            hipMemUnmap(b.base + b.mapped, add);
            hipMemRelease(hd);
            return fail(h, e == hipErrorOutOfMemory ? BRISK_HIP_ENOMEM : BRISK_HIP_EHIP,
                        std::string("arena growth: hipMemSetAccess: ") + hipGetErrorString(e) + " (mapped " + std::to_string(b.mapped >> 20) + " MiB, adding " +
                            std::to_string(add >> 20) + " MiB)");
        }
        b.handles.push_back(hd);
        b.sizes.push_back(add);
        b.mapped += add;
        static const bool dbg = getenv("BRISK_DEBUG_VMM") != nullptr;
        if (dbg) fprintf(stderr, "[brisk_hip] vmm: base %p reserved %zu MiB: mapped +%zu MiB at %p -> %zu MiB\n", (void*)b.base, b.reserved >> 20, add >> 20,
                         (void*)(b.base + b.mapped - add), b.mapped >> 20);
    }
    return BRISK_HIP_OK;
}
void vm_free(VmBuf& b) {  // a reservation nothing was ever mapped into may go back to the runtime; anything else is retired (below)
    if (b.base && b.handles.empty()) hipMemAddressFree(b.base, b.reserved);
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); i++) {
        hipMemUnmap(b.base + off, b.sizes[i]);
        hipMemRelease(b.handles[i]);
        off += b.sizes[i];
    }
    b = VmBuf{};
}

// A virtual address that has been unmapped must never be mapped again in this process.  What was seen (ROCm 7.2, gfx950,
// BRISK_DEBUG_VMM=1, round 1's gpurun_out/fs.log; never reproduced in isolation, so the cause below is an attribution):
//   counts arena, reservation 0x74eca7000000 + 17 GiB:  hipMemCreate/hipMemMap 1 GiB at base; + 1 GiB at 0x74ece7000000;
//   the index is destroyed and its arena trimmed: hipMemUnmap(0x74ece7000000, 1 GiB) + hipMemRelease;
//   the next index grows: hipMemCreate(6 GiB) + hipMemMap(0x74ece7000000, 6 GiB) + hipMemSetAccess(base, 7 GiB)
//   (always from the base: this stack answers "invalid argument" to hipMemSetAccess on some sub-ranges that start past
//   earlier mappings); k_insert then stores counts at 0x74ed00008000 -- 400 MiB into the re-mapped piece, inside the range
//   that had been mapped and unmapped before -- and the GPU reports a memory access fault at exactly that address.
// The same was seen with hipMemAddressFree + hipMemAddressReserve handing the same range out again.  Consistent with a
// translation for the unmapped range surviving the unmap; nothing in this library touches the range in between.  So:
//  * arenas of destroyed indexes go to a small pool WITH their mappings and are handed to the next index on the same
//    device, whatever their size (a bench-size arena -- 100 GB mapped -- is reused, not retired); the pool gives its
//    physical memory back when an allocation of this library fails for lack of memory (pool_drain), so what it holds
//    is never what makes a new index fail;
//  * an arena that is not pooled (pool full: the smallest one goes) is retired: its physical memory is released, its
//    virtual range stays reserved for the life of the process (address space only; g_retired_va counts it,
//    brisk_hip_memory_info reports it) so that nothing is mapped there again.
struct VmSet {
    int device = -1;
    VmBuf keys, counts, ids;
    size_t mapped() const { return keys.mapped + counts.mapped + ids.mapped; }
};
std::mutex g_pool_mu;
std::vector<VmSet> g_pool;
constexpr size_t kPoolMaxArenas = 4;
std::atomic<unsigned long long> g_retired_va{0};  // bytes of address space retired arenas keep reserved

void vm_retire(VmBuf& b) {  // give the physical memory back, keep the address range out of circulation
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); i++) {
        hipMemUnmap(b.base + off, b.sizes[i]);
        hipMemRelease(b.handles[i]);
        off += b.sizes[i];
    }
    if (b.base) g_retired_va += b.reserved;
    b = VmBuf{};
}
bool pool_take(int device, bool want_ids, VmBuf& keys, VmBuf& counts, VmBuf& ids) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) {
        if (g_pool[i].device != device) continue;
        keys = g_pool[i].keys;
        counts = g_pool[i].counts;
        ids = g_pool[i].ids;
        g_pool.erase(g_pool.begin() + i);
        if (!want_ids) vm_retire(ids);
        return true;
    }
    return false;
}
void pool_give(int device, VmBuf& keys, VmBuf& counts, VmBuf& ids) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!keys.base || !counts.base) {
        vm_retire(keys);
        vm_retire(counts);
        vm_retire(ids);
        return;
    }
    VmSet s;
    s.device = device;
    s.keys = keys;
    s.counts = counts;
    s.ids = ids;
    g_pool.push_back(s);
    keys = counts = ids = VmBuf{};
    while (g_pool.size() > kPoolMaxArenas) {  // full: the arena with the least memory behind it is the cheapest to lose
        size_t least = 0;
        for (size_t i = 1; i < g_pool.size(); i++)
            if (g_pool[i].mapped() < g_pool[least].mapped()) least = i;
        vm_retire(g_pool[least].keys);
        vm_retire(g_pool[least].counts);
        vm_retire(g_pool[least].ids);
        g_pool.erase(g_pool.begin() + least);
    }
}
// an allocation failed for lack of device memory: give back what the pool holds; true if that freed anything
bool pool_drain() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    bool any = false;
    for (VmSet& s : g_pool) {
        any = any || s.mapped() > 0;
        vm_retire(s.keys);
        vm_retire(s.counts);
        vm_retire(s.ids);
    }
    g_pool.clear();
    return any;
}
size_t pool_bytes() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t held = 0;
    for (const VmSet& s : g_pool) held += s.mapped();
    return held;
}

// arena growth.  With virtual memory management: map more physical memory behind the reserved
// ranges (no copy).  Without it (reservation failed at create): offsets are bump-allocated, so
// the used prefix moves verbatim into a larger allocation.
int ensure_arena(brisk_hip_index* h, u64 need_entries) {
    if (h->arena_used_host + need_entries <= h->arena_cap) return BRISK_HIP_OK;
    const u64 target = h->arena_used_host + need_entries;
    if (h->arena_limit && target > h->arena_limit) return fail(h, BRISK_HIP_ENOMEM, "arena: BRISK_ARENA_LIMIT reached");
    if (h->use_vmm) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        u64 ncap = std::max<u64>(target, 1u << 16);
        int rc;
        const u64 kb = 8ull * h->ix.key_words;  // bytes of a stored key
        if ((rc = vm_grow(h, h->vm_keys, ncap * kb))) return rc;
        if ((rc = vm_grow(h, h->vm_counts, ncap))) return rc;
        if (h->entry_ids && (rc = vm_grow(h, h->vm_ids, ncap * 4))) return rc;
        ncap = std::min<u64>(h->vm_keys.mapped / kb, h->vm_counts.mapped);
        if (h->entry_ids) ncap = std::min<u64>(ncap, h->vm_ids.mapped / 4);
        h->ix.keys = (u64*)h->vm_keys.base;
        h->ix.counts = (uint8_t*)h->vm_counts.base;
        h->ix.ids = h->entry_ids ? (u32*)h->vm_ids.base : nullptr;
        h->arena_cap = ncap;
        h->ix.arena_cap = ncap;
        return BRISK_HIP_OK;
    }
    u64 ncap = std::max<u64>(h->arena_cap + h->arena_cap / 4, target);
    ncap = std::max<u64>(ncap, 1u << 16);
    u64* nk = nullptr;
    uint8_t* nc = nullptr;
    u32* ni = nullptr;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u64 kb = 8ull * h->ix.key_words;
    hipError_t e = hipMalloc((void**)&nk, ncap * kb);
    if (e == hipSuccess) e = hipMalloc((void**)&nc, ncap);
    if (e == hipSuccess && h->entry_ids) e = hipMalloc((void**)&ni, ncap * 4);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the failed allocation must not surface later as some kernel's launch error
        if (nk) hipFree(nk);
        if (nc) hipFree(nc);
        return fail(h, BRISK_HIP_ENOMEM, "arena growth: out of device memory");
    }
    if (h->arena_used_host) {
        HIPCHK(h, hipMemcpyAsync(nk, h->ix.keys, h->arena_used_host * kb, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(nc, h->ix.counts, h->arena_used_host, hipMemcpyDeviceToDevice, h->stream));
        if (ni) HIPCHK(h, hipMemcpyAsync(ni, h->ix.ids, h->arena_used_host * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (h->ix.keys) hipFree(h->ix.keys);
    if (h->ix.counts) hipFree(h->ix.counts);
    if (h->ix.ids) hipFree(h->ix.ids);
    h->ix.keys = nk;
    h->ix.counts = nc;
    h->ix.ids = ni;
    h->arena_cap = ncap;
    h->ix.arena_cap = ncap;
    return BRISK_HIP_OK;
}

// exclusive prefix of the record histogram -> d_off (n_parts+1), d_cur32 seeded
// The partitions an index can hold records of: all of them, or its owner's range of a sharded job.  Everything that walks the
// partitions of a batch (prefix sums, the touched list) walks this range only: with N owners the directory has N times the
// partitions a single index needs, and each owner sees records in one N-th of them.
struct PartRange {
    u64 lo, len;
};
// first partition of owner o's range: the smallest p with p * N >> part_bits == o
static u64 equal_range_first(const BriskParams& P, u32 o) { return (((u64)o << P.part_bits) + P.n_owners - 1) / P.n_owners; }
static u64 owner_first_partition(const brisk_hip_index* h, u32 o) { return h->owner_cut.empty() ? equal_range_first(h->P, o) : h->owner_cut[o]; }
static PartRange own_partitions(const brisk_hip_index* h) {
    if (h->P.n_owners <= 1) return PartRange{0, h->n_parts};
    const u64 lo = owner_first_partition(h, h->P.owner_rank);
    return PartRange{lo, owner_first_partition(h, h->P.owner_rank + 1) - lo};
}
// exclusive prefix of the histogram's record counts over the index's own partitions -> d_off, d_cur32 (d_off[lo + len] = total)
int prefix_partitions(brisk_hip_index* h, u32 sub = 0) {  // sub: over the records beyond `sub` per partition
    ProfScope ps(h, S_PSUM);
    const PartRange r = own_partitions(h);
    const u32 nb = nblocks(r.len, 256 * SCAN_ITEMS);
    hipLaunchKernelGGL(k_psum_block, dim3(nb), dim3(256), 0, h->stream, h->d_hist + r.lo, r.len, h->d_block_sums, sub);
    hipLaunchKernelGGL(k_psum_top, dim3(1), dim3(1024), 0, h->stream, h->d_block_sums, nb);
    hipLaunchKernelGGL(k_psum_apply, dim3(nb), dim3(256), 0, h->stream, h->d_hist + r.lo, r.len, h->d_block_sums, h->d_off + r.lo, h->d_cur32 + r.lo, sub);
    return launch_check(h, "prefix_partitions");
}
// the own partitions that hold records of this batch -> d_touched, their number -> *d_n
int list_touched(brisk_hip_index* h, u32* d_n) {
    const PartRange r = own_partitions(h);
    hipLaunchKernelGGL(k_touched, dim3(nblocks(r.len, 1024 * TOUCHED_ITEMS)), dim3(1024), 0, h->stream, h->d_hist + r.lo, r.len, (u32)r.lo, h->d_touched, d_n);
    return launch_check(h, "k_touched");
}

// records (unordered, all owned by this index) -> index.  If have_hist, d_hist
// already holds this batch's per-partition histogram (the scan filled it).
// `bl` (binned layout): the scan wrote the records into per-partition bins (bl->bins, bin_cap records each) and the
// n_ovf records beyond them into bl->ovf; d_hist holds the histogram.  Null: d_rec holds the records, in any order.
int verify_records(brisk_hip_index* h, const u64* d_rec, u64 n_rec, u64 want, const char* what);
int verify_hist(brisk_hip_index* h, u64 want_rec, u64 want_inst, const char* what);
struct BinLayout {
    u64* bins;
    u32 bin_cap;
    u64* ovf;
    u64 n_ovf;
    const u32* ovf_cnt;   // the overflow area's region counters (ScanOut::ovf_cnt) and the slots per region
    u32 ovf_region_cap;
    const u32* tags;  // query mode: the read of every record, laid out like the records: [n_parts * bin_cap] binned, then the overflow records'
};
int insert_records_once(brisk_hip_index* h, const u64* d_rec, u64 n_rec, bool have_hist, const BinLayout* bl = nullptr);
#define HUGE_LIST_CAP 65536u   // huge partitions of one batch (k_insert_huge)

// Records in any split are still a valid batch: when the single-pass arena reserve for a batch does not
// fit the device (BRISK_HIP_ENOMEM is raised before anything is written), insert it as two halves.
int insert_records_impl(brisk_hip_index* h, const u64* d_rec, u64 n_rec, bool have_hist) {
    int rc = insert_records_once(h, d_rec, n_rec, have_hist);
    if (rc != BRISK_HIP_ENOMEM || n_rec < 2) return rc;
    const u64 half = n_rec / 2;
    if ((rc = insert_records_impl(h, d_rec, half, false))) return rc;
    return insert_records_impl(h, d_rec + half * h->P.stride, n_rec - half, false);
}

int insert_records_once(brisk_hip_index* h, const u64* d_rec, u64 n_rec, bool have_hist, const BinLayout* bl) {
    h->scan_hist_valid = false;  // d_hist is this batch's from here on
    if (n_rec == 0) return BRISK_HIP_OK;
    if (n_rec >= (1ull << 32)) return fail(h, BRISK_HIP_EINVAL, "more than 2^32-1 records in one batch");
    const BriskParams& P = h->P;
    int rc;
    if (!have_hist) {
        HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
        ProfScope ps(h, S_HIST);
        hipLaunchKernelGGL(k_part_hist, dim3(nblocks(n_rec, 256)), dim3(256), 0, h->stream, P, d_rec, n_rec, h->d_hist);
        if ((rc = launch_check(h, "k_part_hist"))) return rc;
    }
    // classic layout: all records go to partition order; binned layout: only the few beyond their bins do
    const u64 n_move = bl ? bl->n_ovf : n_rec;
    if (n_move && (rc = prefix_partitions(h, bl ? bl->bin_cap : 0u))) return rc;
    if (P.n_owners > 1 && !have_hist) {
        // records counted here, not by their scan: every one of them must lie in this owner's range (the prefix ran over it alone,
        // and a record of another owner would be scattered by a cursor nobody set)
        const PartRange r = own_partitions(h);
        u32 in_range = 0;
        HIPCHK(h, hipMemcpyAsync(&in_range, h->d_off + r.lo + r.len, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (in_range != n_rec) return fail(h, BRISK_HIP_EINVAL, "insert_records: " + std::to_string(n_rec - in_range) + " records belong to other owners (route_records first)");
    }
    HIPCHK(h, hipMemsetAsync(h->d_small + 2, 0, 16, h->stream));
    {
        ProfScope ps(h, S_TOUCHED);
        if ((rc = list_touched(h, (u32*)(h->d_small + 2)))) return rc;
    }
    if (n_move) {
        if ((rc = ensure(h, h->parted, n_move * P.stride * 8))) return rc;
        ProfScope ps(h, S_SCATTER);
        const u64 n_slots = bl ? (u64)bl->ovf_region_cap * OVF_REGIONS : n_move;  // (the overflow area is scattered slot by slot: its regions are filled to different levels)
        hipLaunchKernelGGL(k_scatter, dim3(nblocks(n_slots, 256)), dim3(256), 0, h->stream, P, bl ? (const u64*)bl->ovf : d_rec, n_move, h->d_cur32, (u64*)h->parted.p, 0,
                           (const u32*)nullptr, (u32*)nullptr, h->ix.err, bl ? bl->ovf_cnt : (const u32*)nullptr, bl ? bl->ovf_region_cap : 0u);
        if ((rc = launch_check(h, "k_scatter"))) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small + 2, h->d_small + 2, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u32 n_touched = (u32)h->h_small[2];
    if (n_touched == 0) return BRISK_HIP_OK;
    // partitions of more k-mer instances than this go to k_insert_huge, a workgroup each (0: none; entry-id indexes keep ids per entry,
    // which that kernel does not)
    static const long huge_env = getenv("BRISK_HUGE_AT") ? atol(getenv("BRISK_HUGE_AT")) : -1;  // tests: a small threshold sends ordinary partitions there
    const u32 huge_at = h->entry_ids ? 0u : huge_env >= 0 ? (u32)huge_env : 16384u;
    {
        ProfScope ps(h, S_TOUCHED);
        if ((rc = ensure(h, h->desc, (size_t)n_touched * sizeof(PartDesc)))) return rc;
        if (huge_at) {
            if ((rc = ensure(h, h->huge, (size_t)(HUGE_LIST_CAP + 1) * 4))) return rc;
            HIPCHK(h, hipMemsetAsync(h->huge.p, 0, 4, h->stream));
        }
        hipLaunchKernelGGL(k_need, dim3(std::min<u32>(nblocks(n_touched, 256), 2048)), dim3(256), 0, h->stream, h->d_hist, (bl && !n_move) ? (const u32*)nullptr : h->d_off,
                           h->d_touched, n_touched, h->ix.dir, (PartDesc*)h->desc.p, h->d_small + 3, bl ? bl->bin_cap : 0u, huge_at, huge_at ? (u32*)h->huge.p : (u32*)nullptr,
                           (u32)HUGE_LIST_CAP);
        if (int lrc = launch_check(h, "k_need")) return lrc;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small + 3, h->d_small + 3, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_small + 5, h->ix.cursor, 8, hipMemcpyDeviceToHost, h->stream));
    h->h_small[4] = 0;
    if (huge_at) HIPCHK(h, hipMemcpyAsync(h->h_small + 4, h->huge.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->arena_used_host = h->h_small[5];
    // (more huge partitions than the list holds: none is diverted, the wave kernels take them all)
    const u32 n_huge = (u32)h->h_small[4] <= HUGE_LIST_CAP ? (u32)h->h_small[4] : 0u;
    h->ix.huge_at = n_huge ? huge_at : 0u;
    // worst case: every slice the batch may need, the tail a private chunk strands at each refill (< 1/8 of it), and
    // one partly used chunk per persistent wave
    if ((rc = ensure_arena(h, h->h_small[3] + h->h_small[3] / 7 + (u64)h->insert_waves * ARENA_CHUNK))) return rc;
    {
        ProfScope ps(h, S_INSERT);
        HIPCHK(h, hipMemsetAsync(h->d_small + 6, 0, 8, h->stream));
        // partitions of many records (few distinct minimizers: m <= 11) take the 512-instance kernel: half as many chunks, and
        // with them half as many passes over a partition's entries, outweigh its 2 waves per SIMD (k31/m11/b11: 48 -> 36 ms)
        const u32 batches = (n_touched + WI_BATCH - 1) / WI_BATCH;
        static const u64 big_at = getenv("BRISK_INSERT_BIG_AT") ? (u64)atoll(getenv("BRISK_INSERT_BIG_AT")) : 64ull;  // experiments
        const bool big = !bl && n_rec / n_touched > big_at;  // (its in-place collapse needs the classic layout)
        const RecSrc src{bl ? bl->bins : (u64*)h->parted.p, (const u64*)h->parted.p, bl ? bl->bin_cap : 0u};
        // persistent waves: as many as the device keeps resident (the kernel's time follows their number almost linearly)
        static const u32 wave_env = getenv("BRISK_INSERT_WAVES") ? (u32)atoi(getenv("BRISK_INSERT_WAVES")) : 0u;  // experiments
        auto resident = [&](const void* fn) -> u32 {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) cus = 256;
            const u32 w = std::min<u32>((u32)per_cu * (u32)cus, h->insert_waves);
            static const bool dbg = getenv("BRISK_DEBUG_INSERT") != nullptr;
            if (dbg) fprintf(stderr, "[brisk_hip] k_insert: %d waves per CU x %d CUs resident, %u launched\n", per_cu, cus, wave_env ? std::min<u32>(w, wave_env) : w);
            return wave_env ? std::min<u32>(w, wave_env) : w;
        };
        // instantiations with the record geometry (nw, k - b, routing-id bits kept in the key) as constants, for the
        // common parameter sets under the default partition layout; anything else takes the generic body
#define LAUNCH_INSERT(NW, KB, SH)                                                                                                                                   \
    {                                                                                                                                                               \
        if (big) {                                                                                                                                                  \
            const dim3 grid(std::min<u32>(batches, resident((const void*)k_insert_big<NW, KB, SH>)));                                                             \
            hipLaunchKernelGGL((k_insert_big<NW, KB, SH>), grid, dim3(64), 0, h->stream, P, src, (const PartDesc*)h->desc.p, n_touched, h->ix,                    \
                               (u32*)(h->d_small + 6));                                                                                                             \
        } else {                                                                                                                                                    \
            const dim3 grid(std::min<u32>(batches, resident((const void*)k_insert<NW, KB, SH>)));                                                                 \
            hipLaunchKernelGGL((k_insert<NW, KB, SH>), grid, dim3(64), 0, h->stream, P, src, (const PartDesc*)h->desc.p, n_touched, h->ix,                        \
                               (u32*)(h->d_small + 6));                                                                                                             \
        }                                                                                                                                                           \
    }
#define LAUNCH_INSERT_FAST(NW, KB, SH)                                                                                                                              \
    {                                                                                                                                                               \
        if (big) LAUNCH_INSERT(NW, KB, SH)                                                                                                                          \
        else {                                                                                                                                                      \
            const dim3 grid(std::min<u32>(batches, resident((const void*)k_insert_fast<NW, KB, SH>)));                                                            \
            hipLaunchKernelGGL((k_insert_fast<NW, KB, SH>), grid, dim3(64), 0, h->stream, P, src, (const PartDesc*)h->desc.p, n_touched, h->ix,                   \
                               (u32*)(h->d_small + 6));                                                                                                             \
        }                                                                                                                                                           \
    }
        static const bool generic_only = getenv("BRISK_INSERT_GENERIC") != nullptr;  // A/B and tests: force the run-time body
        if (!generic_only && P.nw == 3 && P.kb == 49 && P.shift == 4) LAUNCH_INSERT_FAST(3, 49, 4)        // k63 m21 b14
        else if (!generic_only && P.nw == 3 && P.kb == 49 && P.shift == 3) LAUNCH_INSERT_FAST(3, 49, 3)  // the same with 2^25..2^27 partitions:
        else if (!generic_only && P.nw == 3 && P.kb == 49 && P.shift == 2) LAUNCH_INSERT_FAST(3, 49, 2)  // jobs of 2..8 x 50 M reads per batch over
        else if (!generic_only && P.nw == 3 && P.kb == 49 && P.shift == 1) LAUNCH_INSERT_FAST(3, 49, 1)  // as many owners (brisk_hip_options.part_bits)
        else if (!generic_only && P.nw == 2 && P.kb == 17 && P.shift == 4) LAUNCH_INSERT_FAST(2, 17, 4)  // k31 m15 b14 (apps/counter.cpp:355)
        else if (!generic_only && P.nw == 2 && P.kb == 17 && P.shift == 5) LAUNCH_INSERT_FAST(2, 17, 5)  // (the same with 2^23 / 2^22 partitions: one batch of ~20 M reads,
        else if (!generic_only && P.nw == 2 && P.kb == 17 && P.shift == 6) LAUNCH_INSERT_FAST(2, 17, 6)  // brisk_hip_part_bits_for_batch)
        else if (!generic_only && P.nw == 2 && P.kb == 20 && P.shift == 0) LAUNCH_INSERT_FAST(2, 20, 0)  // k31 m11 b11
        else LAUNCH_INSERT(0, 0, 0)
#undef LAUNCH_INSERT_FAST
#undef LAUNCH_INSERT
        if ((rc = launch_check(h, big ? "k_insert_big" : "k_insert"))) return rc;
        if (h->trace) fprintf(stderr, "[brisk_hip] path: insert of %llu records (%s layout, histogram %s) into %u partitions: %s, %u partitions to k_insert_huge\n", (unsigned long long)n_rec,
                              bl ? "binned" : "classic", have_hist ? "from the scan" : "counted here", n_touched, big ? "k_insert_big (in-place collapse for partitions of > 128 records)" : "k_insert", n_huge);
        if (n_huge) {
            hipLaunchKernelGGL(k_insert_huge, dim3(std::min<u32>(n_huge, 256u)), dim3(HG_THREADS), 0, h->stream, P, src, (const PartDesc*)h->desc.p, (const u32*)h->huge.p + 1,
                               (const u32*)h->huge.p, h->ix);
            if ((rc = launch_check(h, "k_insert_huge"))) return rc;
        }
    }
    h->nb_skmers += n_rec;
    h->dir_snapshot_valid = false;
    return BRISK_HIP_OK;
}

// scan reads -> records in d_rec (cap records).  n_rec_out on host after a sync.
// one scan launch over n_items reads (or virtual reads when cc.vreads is set); counters are NOT reset here
int launch_scan(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_items, const ScanOut& out, bool query_mode, bool plain,
                const ChunkCtl& cc) {
    ProfScope ps(h, S_SCAN);
    if (plain) {  // sequence mode needs the minimizer values: the plain kernel carries them
        BriskParams P = h->P;
        if (out.ret) {  // and whole vectors: see brisk_hip_scan_sequence
            P.ext_bits -= P.cls_bits;
            P.cls_bits = 0;
        }
        hipLaunchKernelGGL(k_scan, dim3(nblocks(n_items, SCAN_BLOCK)), dim3(SCAN_BLOCK), 0, h->stream, P, d_packed, d_starts, n_items, h->d_coef,
                           out, query_mode ? 1 : 0);
    } else {
        const u32 bt = h->scan_waves * 64;
        const dim3 grid(nblocks(n_items, bt)), block(bt);
        // instantiations: {k and m compile-time for the common parameter sets (the class tables' layout folds into the
        // instructions), or from P with the chunk count unrolled (6: m 27..29, 4: m 17..19, 3: m 11..15 at CLS_W 5; else a loop)} x mode
#define LAUNCH_SCAN2(NCH, MODE, KK, MM) \
    hipLaunchKernelGGL((k_scan2<NCH, MODE, KK, MM>), grid, block, h->scan_lds, h->stream, h->P, h->scfg, d_packed, d_starts, n_items, h->d_tabs, out, cc)
#define LAUNCH_SCAN2_MODES(NCH, KK, MM)                    \
    {                                                      \
        if (cc.vreads) LAUNCH_SCAN2(NCH, 2, KK, MM);       \
        else if (query_mode) LAUNCH_SCAN2(NCH, 1, KK, MM); \
        else LAUNCH_SCAN2(NCH, 0, KK, MM);                 \
    }
        const u32 k = h->P.k, m = h->P.m;
        if (k == 63 && m == 21) LAUNCH_SCAN2_MODES(0, 63, 21)
        else if (k == 31 && m == 15) LAUNCH_SCAN2_MODES(0, 31, 15)  // the reference's default parameters (apps/counter.cpp:355)
        else if (k == 31 && m == 11) LAUNCH_SCAN2_MODES(0, 31, 11)
        else if (h->scfg.nch == 6) LAUNCH_SCAN2_MODES(6, 0, 0)
        else if (h->scfg.nch == 4) LAUNCH_SCAN2_MODES(4, 0, 0)
        else if (h->scfg.nch == 3) LAUNCH_SCAN2_MODES(3, 0, 0)
        else LAUNCH_SCAN2_MODES(0, 0, 0)
#undef LAUNCH_SCAN2_MODES
#undef LAUNCH_SCAN2
    }
    return launch_check(h, "k_scan");
}

// Scan a batch into d_rec.  Long sequences (insert mode only) are scanned as chunks, checked at the
// seams and, if a seam does not match, re-scanned whole; *hist_valid tells whether d_hist still
// describes exactly the records in d_rec.
int scan_impl(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, u64* d_rec, u64 cap, bool with_hist,
              bool query_mode, u32* d_tags, u64* n_rec_out, u64* d_ret = nullptr, u64 kmer_bound = 0, bool* hist_valid = nullptr, bool keep_hist = false) {
    if (hist_valid) *hist_valid = with_hist;
    if (with_hist) {
        h->scan_hist_valid = false;
        if (!keep_hist) HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));  // (keep_hist: add to the pending records' counts)
    }
    HIPCHK(h, hipMemsetAsync(h->d_small, 0, 16, h->stream));
    ScanOut out{d_rec, cap, h->d_small, with_hist ? h->d_hist : nullptr, (u32*)(h->d_small + 1), d_tags, d_ret, nullptr, 0u, nullptr, 0u, nullptr};
    const bool plain = h->scan_v1 || d_ret;
    int rc;
    u32 n_vr = 0;
    // long sequences are chunked in insert mode (no tags) and in query mode (read tags); not in sequence mode (plain)
    const bool may_chunk = !plain && kmer_bound > SCAN_LONG && (query_mode ? d_tags != nullptr : !d_tags);
    VRead *d_vr = nullptr, *d_rerun = nullptr;
    ChunkState *d_spec = nullptr, *d_truth = nullptr;
    u32 *d_status = nullptr, *d_cursor = nullptr, *d_stop = nullptr;
    u64 cap_vr = 0;
    // chunk length: long enough to amortise the warm-up, short enough to give the device lanes to fill
    u32 chunk = 4096;
    while (chunk > 1024 && kmer_bound / chunk < (1u << 18)) chunk >>= 1;
    if (may_chunk) {
        cap_vr = kmer_bound / chunk + kmer_bound / SCAN_LONG + 2;
        const u64 cap_long = kmer_bound / SCAN_LONG + 1;
        const size_t bytes = 2 * cap_vr * sizeof(VRead) + (2 * cap_vr + 1) * sizeof(ChunkState) + 3 * cap_vr * 4 + cap_long * sizeof(LongRead);
        if ((rc = ensure(h, h->chunk_buf, bytes))) return rc;
        d_vr = (VRead*)h->chunk_buf.p;
        d_rerun = d_vr + cap_vr;
        d_spec = (ChunkState*)(d_rerun + cap_vr);
        d_truth = d_spec + cap_vr;
        d_status = (u32*)(d_truth + cap_vr + 1);
        d_cursor = d_status + cap_vr;
        d_stop = d_cursor + cap_vr;
        LongRead* d_long = (LongRead*)(d_stop + cap_vr + (cap_vr & 1));
        HIPCHK(h, hipMemsetAsync(h->d_small + 6, 0, 16, h->stream));
        hipLaunchKernelGGL(k_plan_chunks, dim3(nblocks(n_reads, 256)), dim3(256), 0, h->stream, d_starts, n_reads, h->P.k, chunk, (u32)cap_vr,
                           (u32*)(h->d_small + 7), d_long, (u32*)(h->d_small + 6));
        if (int lrc = launch_check(h, "k_plan_chunks")) return lrc;
        HIPCHK(h, hipMemcpyAsync(h->h_small + 6, h->d_small + 6, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const u32 n_long = (u32)h->h_small[6];
        if (n_long) {
            hipLaunchKernelGGL(k_fill_chunks, dim3(std::min<u32>(n_long, 65535u)), dim3(256), 0, h->stream, d_starts, (u32)h->P.k, (u32)h->P.w, chunk, d_long, n_long, d_vr);
            if (int lrc = launch_check(h, "k_fill_chunks")) return lrc;
        }
        n_vr = (u32)h->h_small[7];
        if (n_vr > cap_vr) return fail(h, BRISK_HIP_EHIP, "chunk plan exceeds its bound");
    }
    ChunkCtl cc{nullptr, nullptr, nullptr, n_vr ? SCAN_LONG : 0u};
    if ((rc = launch_scan(h, d_packed, d_starts, n_reads, out, query_mode, plain, cc))) return rc;
    if (n_vr) {
        // records of the short reads are in [0, n1); the chunked launch appends after them and tags its records with
        // their chunk; seeded re-scans append after those
        HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const u64 n1 = std::min<u64>(h->h_small[0], cap);
        u32* d_ctags = d_tags;  // chunk index per record of the chunked launches (query mode: in the caller's tag array, read indices later)
        u64* d_qret = nullptr;  // query mode: where a record's vector starts | its minimizer is 0 << 63
        if (!query_mode) {
            if ((rc = ensure(h, h->tags_a, cap * 4))) return rc;
            d_ctags = (u32*)h->tags_a.p;
        } else {
            if ((rc = ensure(h, h->seq_buf, cap * 8 + cap_vr * 8))) return rc;
            d_qret = (u64*)h->seq_buf.p;
        }
        HIPCHK(h, hipMemsetAsync(d_spec, 0, (2 * cap_vr + 1) * sizeof(ChunkState) + 2 * cap_vr * 4, h->stream));  // states, status, cursor
        HIPCHK(h, hipMemsetAsync(d_stop, 0xff, cap_vr * 4, h->stream));
        ScanOut out2 = out;
        out2.tag = d_ctags;
        out2.ret = d_qret;
        ChunkCtl c2{d_vr, d_spec, d_truth, 0u};
        if ((rc = launch_scan(h, d_packed, d_starts, n_vr, out2, false, false, c2))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const u64 n2 = std::min<u64>(h->h_small[0], cap);
        ScanOut out3 = out2;
        out3.hist = nullptr;  // once a chunk is re-scanned the histogram is rebuilt from the final records
        u64 total_rerun = 0;
        u32 rounds = 0;
        for (;; rounds++) {  // one round per mismatch along a sequence; no mismatch: one round
            HIPCHK(h, hipMemsetAsync(h->d_small + 7, 0, 8, h->stream));
            hipLaunchKernelGGL(k_chunk_match, dim3(nblocks(n_vr, 256)), dim3(256), 0, h->stream, d_vr, d_spec, d_truth, n_vr, d_cursor, d_stop);
            hipLaunchKernelGGL(k_chunk_commit, dim3(nblocks(n_vr, 256)), dim3(256), 0, h->stream, d_vr, n_vr, chunk, (u32)h->P.k, (u32)h->P.w, d_starts, d_cursor,
                               d_stop, d_status, d_rerun, (u32*)(h->d_small + 7));
            hipLaunchKernelGGL(k_chunk_next, dim3(nblocks(n_vr, 256)), dim3(256), 0, h->stream, d_vr, n_vr, d_cursor, d_stop);
            if (int lrc = launch_check(h, "k_chunk_match/commit/next")) return lrc;
            HIPCHK(h, hipMemcpyAsync(h->h_small + 7, h->d_small + 7, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const u32 n_rerun = (u32)h->h_small[7];
            if (!n_rerun) break;
            total_rerun += n_rerun;
            ChunkCtl c3{d_rerun, d_spec, d_truth, 0u};
            if ((rc = launch_scan(h, d_packed, d_starts, n_rerun, out3, false, false, c3))) return rc;
        }
        static const bool dbg_chunks = getenv("BRISK_DEBUG_CHUNKS") != nullptr;
        if (dbg_chunks) fprintf(stderr, "[brisk_hip] chunked scan: %u chunks of %u steps, %llu seeded re-scans in %u rounds\n", n_vr, chunk,
                                (unsigned long long)total_rerun, rounds);
        if (query_mode) {
            // stop every sequence where query_sequence stops it, drop the void records, tag the rest with their read
            HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (!(u32)h->h_small[1]) {
                const u64 n3 = std::min<u64>(h->h_small[0], cap);
                unsigned long long* d_brk = (unsigned long long*)(d_qret + cap);
                if ((rc = ensure(h, h->parted, (n3 - n1 + 1) * (h->P.stride * 8 + 4)))) return rc;
                u64* stage = (u64*)h->parted.p;
                u32* stage_tags = (u32*)(stage + (n3 - n1 + 1) * h->P.stride);
                HIPCHK(h, hipMemsetAsync(d_brk, 0xff, cap_vr * 8, h->stream));
                HIPCHK(h, hipMemsetAsync(h->d_small + 6, 0, 8, h->stream));
                if (n3 > n1) {
                    hipLaunchKernelGGL(k_query_break, dim3(nblocks(n3 - n1, 256)), dim3(256), 0, h->stream, d_qret, d_ctags, n1, n2, n3, d_vr, d_status, d_starts, d_brk);
                    hipLaunchKernelGGL(k_query_filter, dim3(nblocks(n3 - n1, 256)), dim3(256), 0, h->stream, h->P, d_rec, d_qret, d_ctags, n1, n2, n3, d_vr, d_status,
                                       d_brk, stage, stage_tags, h->d_small + 6);
                    if (int lrc = launch_check(h, "k_query_break/filter")) return lrc;
                }
                HIPCHK(h, hipMemcpyAsync(h->h_small + 6, h->d_small + 6, 8, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                const u64 kept = h->h_small[6];
                if (kept) {
                    HIPCHK(h, hipMemcpyAsync(d_rec + n1 * h->P.stride, stage, kept * h->P.stride * 8, hipMemcpyDeviceToDevice, h->stream));
                    HIPCHK(h, hipMemcpyAsync(d_ctags + n1, stage_tags, kept * 4, hipMemcpyDeviceToDevice, h->stream));
                }
                h->h_small[6] = n1 + kept;  // pinned: stays untouched until the copy below has run
                HIPCHK(h, hipMemcpyAsync(h->d_small, h->h_small + 6, 8, hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
            if (hist_valid) *hist_valid = false;
        } else if (total_rerun) {
            HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (!(u32)h->h_small[1]) {
                // drop what the re-scanned chunks emitted speculatively ([n1, n2)); what the seeded scans emitted ([n2, n3)) stays
                const u64 n3 = std::min<u64>(h->h_small[0], cap);
                if ((rc = ensure(h, h->parted, (n3 - n1 + 1) * h->P.stride * 8))) return rc;
                HIPCHK(h, hipMemsetAsync(h->d_small + 6, 0, 8, h->stream));
                hipLaunchKernelGGL(k_filter_records, dim3(nblocks(n2 - n1, 256)), dim3(256), 0, h->stream, h->P, d_rec, (const u32*)h->tags_a.p, n1, n2, d_status,
                                   (u64*)h->parted.p, h->d_small + 6);
                if (int lrc = launch_check(h, "k_filter_records")) return lrc;
                HIPCHK(h, hipMemcpyAsync(h->h_small + 6, h->d_small + 6, 8, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                const u64 kept = h->h_small[6];
                char* stage = (char*)h->parted.p;
                if (n3 > n2) HIPCHK(h, hipMemcpyAsync(stage + kept * h->P.stride * 8, d_rec + n2 * h->P.stride, (n3 - n2) * h->P.stride * 8, hipMemcpyDeviceToDevice, h->stream));
                if (kept + n3 - n2) HIPCHK(h, hipMemcpyAsync(d_rec + n1 * h->P.stride, stage, (kept + n3 - n2) * h->P.stride * 8, hipMemcpyDeviceToDevice, h->stream));
                h->h_small[6] = n1 + kept + (n3 - n2);  // pinned: stays untouched until the copy below has run
                HIPCHK(h, hipMemcpyAsync(h->d_small, h->h_small + 6, 8, hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
            if (hist_valid) *hist_valid = false;
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_rec_out = h->h_small[0];
    if ((u32)h->h_small[1]) return BRISK_HIP_ECAPACITY;
    return BRISK_HIP_OK;
}

int count_kmers(brisk_hip_index* h, const u64* d_starts, u64 n_reads, u64* out, u64* out_long = nullptr) {
    HIPCHK(h, hipMemsetAsync(h->d_small + 4, 0, 16, h->stream));
    {
        ProfScope ps(h, S_COUNT);
        const u32 grid = std::min<u32>(nblocks(n_reads, 256), 4096);
        hipLaunchKernelGGL(k_count_kmers, dim3(grid ? grid : 1), dim3(256), 0, h->stream, d_starts, n_reads, h->P.k, h->d_small + 4);
        if (int lrc = launch_check(h, "k_count_kmers")) return lrc;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small + 4, h->d_small + 4, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->h_small[5] >> 63) return fail(h, BRISK_HIP_EINVAL, "read offsets do not ascend (offsets[i + 1] < offsets[i]): not a read table, or the buffer was overwritten while the call ran");
    *out = h->h_small[4];
    if (out_long) *out_long = h->h_small[5];
    return BRISK_HIP_OK;
}

// first guess at the records `bound` k-mers of n_reads reads become: one per ~(w+2)/2 k-mers (a quarter more, for slack), the one
// that ends every read, and with minimizer_idx classes a cut every cls_width k-mers
static u64 records_estimate(const brisk_hip_index* h, u64 bound, u64 n_reads) {
    u64 est = 5 * bound / (2 * (h->P.w + 2)) + 2 * n_reads + 4096;
    if (h->P.cls_bits) est += 5 * bound / (4 * h->P.cls_width);
    return std::min<u64>(bound, est);
}

// scan a batch into the staging buffer, retrying once with the exact bound
int scan_to_staging(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, bool with_hist, bool query_mode,
                    u64* n_rec_out, bool* hist_valid = nullptr) {
    int rc;
    u64 bound = 0, in_long = 0;
    if ((rc = count_kmers(h, d_starts, n_reads, &bound, &in_long))) return rc;
    if (bound == 0) {
        *n_rec_out = 0;
        return BRISK_HIP_OK;
    }
    (void)in_long;
    u64 cap = records_estimate(h, bound, n_reads);
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = ensure(h, h->staging, cap * h->P.stride * 8))) return rc;
        u32* tags = nullptr;
        if (query_mode) {
            if ((rc = ensure(h, h->tags_a, cap * 4))) return rc;
            tags = (u32*)h->tags_a.p;
        }
        rc = scan_impl(h, d_packed, d_starts, n_reads, (u64*)h->staging.p, cap, with_hist, query_mode, tags, n_rec_out, nullptr, bound, hist_valid);
        if (rc == BRISK_HIP_OK && h->verify && !query_mode) return verify_records(h, (const u64*)h->staging.p, *n_rec_out, bound, "scan to staging");
        if (rc != BRISK_HIP_ECAPACITY) return rc;
        cap = bound;
    }
    return fail(h, BRISK_HIP_EHIP, "scan overflowed its exact bound");
}

// the parameter sets k_insert_fast / k_query_fast are instantiated for (the only kernels that read the binned layout in query mode)
static bool has_fast_geometry(const BriskParams& P) {
    return (P.nw == 3 && P.kb == 49 && P.shift >= 1 && P.shift <= 4) || (P.nw == 2 && P.kb == 17 && P.shift >= 4 && P.shift <= 6) || (P.nw == 2 && P.kb == 20 && P.shift == 0);
}

// Scan a batch straight into per-partition bins (insert or query mode).  *applied = false: the batch does not qualify, or its
// records did not fit (nothing of the index has been touched): the classic path takes it.
int scan_binned(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, bool query_mode, bool* applied, BinLayout* bl, u64* n_rec_out) {
    *applied = false;
    static const long forced = getenv("BRISK_BINS") ? atol(getenv("BRISK_BINS")) : -1;  // 0: never; S > 0: always, with bins of S records (tests)
    if (forced == 0 || h->entry_ids || h->scan_v1 || h->P.n_owners > 1) return BRISK_HIP_OK;
    // minimizers short enough for class bits are few and unevenly used: a tenth of the partitions hold everything, and bins sized for
    // the mean overflow
    if (forced < 0 && h->P.cls_bits) return BRISK_HIP_OK;
    static const bool query_generic = getenv("BRISK_QUERY_GENERIC") != nullptr;
    if (query_mode && (query_generic || !has_fast_geometry(h->P))) return BRISK_HIP_OK;
    int rc;
    u64 bound = 0, in_long = 0;
    if ((rc = count_kmers(h, d_starts, n_reads, &bound, &in_long))) return rc;
    if (bound == 0 || in_long) return BRISK_HIP_OK;  // long sequences are scanned in chunks whose records are filtered afterwards
    const u64 est = records_estimate(h, bound, n_reads);  // the staging path's first guess: ~1.3x the records
    u64 cap = forced > 0 ? (u64)forced : (2 * est / h->n_parts + 8 + 3) / 4 * 4;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const u64 bytes = h->n_parts * cap * h->P.stride * 8;
    // (bins of up to 256 records: fewer, bigger partitions -- options.part_bits fitted to the batch -- keep the rank-is-the-slot layout)
    if (forced <= 0 && (est < 2 * h->n_parts || cap > 256 || bytes > total_b / 4)) return BRISK_HIP_OK;  // sparse batch, big partitions, or too much memory
    if (h->n_parts * cap >= (1ull << 32)) return BRISK_HIP_OK;
    // (tests force tiny bins: most records lie beyond them, and the regions -- filled by the low bits of the partition -- are uneven when partitions are few)
    const u32 ovf_region_cap = (u32)(((forced > 0 ? 4 * est : est / 8) + 65536 + OVF_REGIONS - 1) / OVF_REGIONS);
    const u64 ovf_cap = (u64)ovf_region_cap * OVF_REGIONS;
    if ((rc = ensure(h, h->bins, bytes))) return rc == BRISK_HIP_ENOMEM ? (h->err.clear(), BRISK_HIP_OK) : rc;
    if ((rc = ensure(h, h->staging, ovf_cap * h->P.stride * 8))) return rc;
    u32* tags = nullptr;
    if (query_mode) {
        if ((rc = ensure(h, h->tags_a, (h->n_parts * cap + ovf_cap) * 4))) return rc == BRISK_HIP_ENOMEM ? (h->err.clear(), BRISK_HIP_OK) : rc;
        tags = (u32*)h->tags_a.p;
    }
    h->scan_hist_valid = false;
    HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_small, 0, 16, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_small + 7, 0, 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_ovf_cnt, 0, OVF_REGIONS * 4, h->stream));
    ScanOut out{nullptr, 0, h->d_small, h->d_hist, (u32*)(h->d_small + 1), tags, nullptr, (u64*)h->bins.p, (u32)cap, (u64*)h->staging.p, ovf_region_cap, h->d_ovf_cnt};
    ChunkCtl cc{nullptr, nullptr, nullptr, 0u};
    if ((rc = launch_scan(h, d_packed, d_starts, n_reads, out, query_mode, false, cc))) return rc;
    hipLaunchKernelGGL(k_sum_regions, dim3(1), dim3(1024), 0, h->stream, h->d_ovf_cnt, ovf_region_cap, h->d_small + 7);
    if ((rc = launch_check(h, "k_sum_regions"))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_small + 7, h->d_small + 7, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((u32)h->h_small[1]) return BRISK_HIP_OK;  // more records beyond the bins than the overflow buffer holds: the classic path takes the batch
    *n_rec_out = h->h_small[0];
    if (h->trace) fprintf(stderr, "[brisk_hip] path: binned scan of %llu reads: %llu records, bins of %llu, %llu records beyond their bins\n", (unsigned long long)n_reads,
                          (unsigned long long)h->h_small[0], (unsigned long long)cap, (unsigned long long)h->h_small[7]);
    *bl = BinLayout{(u64*)h->bins.p, (u32)cap, (u64*)h->staging.p, h->h_small[7], h->d_ovf_cnt, ovf_region_cap, tags};
    if (h->verify && !query_mode && (rc = verify_hist(h, h->h_small[0], bound, "binned scan"))) return rc;
    *applied = true;
    return BRISK_HIP_OK;
}

// The binned insert of one batch: the scan writes every record straight into its partition's bin --
// the rank the histogram atomic returns is the slot -- so no staging copy and no k_scatter pass exist; the few records
// beyond a bin (bin_cap is about twice the expected mean) are moved by the classic scatter.  *applied = false when the
// batch does not qualify (nothing has been touched then) and the classic path must take it.
int insert_packed_binned(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, bool* applied) {
    BinLayout bl{};
    u64 n_rec = 0;
    int rc = scan_binned(h, d_packed, d_starts, n_reads, false, applied, &bl, &n_rec);
    if (rc || !*applied) return rc;
    *applied = false;
    rc = insert_records_once(h, nullptr, n_rec, true, &bl);
    if (rc == BRISK_HIP_ENOMEM) {  // the single-pass arena reserve does not fit: the classic path can split the batch (nothing was written)
        h->err.clear();
        return BRISK_HIP_OK;
    }
    *applied = rc == BRISK_HIP_OK;
    return rc;
}

// ---- deferred inserts -------------------------------------------------------------------------------------------------------
// The insert works on partitions: its time follows the number of partitions a batch touches, and only with 10-20 records in
// each is that time spent on k-mers (50 M reads in one batch: 31 ms; in 25 batches of 2 M: 180 ms, on an index of 2^24
// partitions).  The scan has no such threshold.  So a batch that would leave the partitions nearly empty is scanned right away
// -- its records appended to h->pend, their per-partition counts added to d_hist -- and inserted together with the batches that
// follow it, as soon as there are DEFER_FLUSH_AT records per partition or any call other than an insert needs the index or the
// scratch the pending state lives in (enter()).  Results are those of inserting every batch at once: counts add up and wrap the
// same way in any grouping (tests run the same inputs with BRISK_DEFER=0 and 1).
#define DEFER_DIRECT_AT 10u   // batches estimated at this many records per partition or more are inserted directly
#define DEFER_FLUSH_AT 12u    // pending records per partition at which they are inserted

int flush_pending(brisk_hip_index* h) {
    if (!h->n_pend) return BRISK_HIP_OK;
    const u64 n = h->n_pend;
    const bool hist_ok = h->pend_hist_ok;
    if (h->trace) fprintf(stderr, "[brisk_hip] path: flush of %llu pending records\n", (unsigned long long)n);
    h->n_pend = 0;  // whatever happens, the records are consumed (a failed insert is a failed insert)
    h->pend_hist_ok = true;
    return insert_records_impl(h, (const u64*)h->pend.p, n, hist_ok);
}

// room for `more` records behind the pending ones (contents kept)
static int pend_reserve(brisk_hip_index* h, u64 more) {
    const size_t rec_bytes = h->P.stride * 8, need = (h->n_pend + more) * rec_bytes;
    if (h->pend.bytes >= need) return BRISK_HIP_OK;
    // a stream of real batches gets all it will ever need at once (the flush threshold plus the largest deferrable batch: 12 GB on
    // 2^24 partitions at k = 63) -- growing there means copying gigabytes; tiny indexes grow by doubling
    const size_t full = (size_t)(DEFER_FLUSH_AT + DEFER_DIRECT_AT) * h->n_parts * rec_bytes;
    DevBuf bigger;
    int rc = ensure(h, bigger, need > ((size_t)256 << 20) ? std::max(need, full) : std::max<size_t>(need, 2 * h->pend.bytes));
    if (rc) return rc;
    if (h->n_pend) HIPCHK(h, hipMemcpyAsync(bigger.p, h->pend.p, h->n_pend * rec_bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->pend.p) (void)hipFree(h->pend.p);
    h->pend = bigger;
    return BRISK_HIP_OK;
}

// scan a small batch into the pending records; *deferred = false: the batch is dense enough (or the index not of the kind) to be inserted directly
int defer_batch(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, bool* deferred) {
    *deferred = false;
    if (!h->defer || h->entry_ids || h->scan_v1 || h->P.n_owners > 1) return BRISK_HIP_OK;
    int rc;
    u64 bound = 0;
    if ((rc = count_kmers(h, d_starts, n_reads, &bound))) return rc;
    if (bound == 0) {
        *deferred = true;  // nothing to insert
        return BRISK_HIP_OK;
    }
    const u64 est = records_estimate(h, bound, n_reads);
    if (est >= (u64)DEFER_DIRECT_AT * h->n_parts) return BRISK_HIP_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        const u64 cap = attempt ? bound : est;
        if ((rc = pend_reserve(h, cap))) {
            if (rc != BRISK_HIP_ENOMEM) return rc;
            // no room for the pending buffer: what is pending goes in now, and this batch takes the direct path, which needs none
            h->err.clear();
            return flush_pending(h);
        }
        u64 n_rec = 0;
        bool hist_ok = true;
        const bool with_hist = h->pend_hist_ok;
        rc = scan_impl(h, d_packed, d_starts, n_reads, (u64*)h->pend.p + h->n_pend * h->P.stride, cap, with_hist, false, nullptr, &n_rec, nullptr, bound, &hist_ok,
                       /*keep_hist=*/h->n_pend > 0);
        if (rc == BRISK_HIP_ECAPACITY) {  // the first guess was too small: what it counted before it stopped is in d_hist
            h->pend_hist_ok = false;
            h->err.clear();
            continue;
        }
        if (rc) return rc;
        if (h->verify && (rc = verify_records(h, (const u64*)h->pend.p + h->n_pend * h->P.stride, n_rec, bound, "deferred scan"))) return rc;
        if (!with_hist || !hist_ok) h->pend_hist_ok = false;
        if (h->trace) fprintf(stderr, "[brisk_hip] path: deferred scan of %llu reads (attempt %d): %llu records behind %llu pending ones, histogram %s\n", (unsigned long long)n_reads, attempt,
                              (unsigned long long)n_rec, (unsigned long long)h->n_pend, h->pend_hist_ok ? "kept" : "to be recounted");
        h->n_pend += n_rec;
        *deferred = true;
        if (h->n_pend >= (u64)DEFER_FLUSH_AT * h->n_parts) return flush_pending(h);
        return BRISK_HIP_OK;
    }
    return fail(h, BRISK_HIP_EHIP, "scan overflowed its exact bound");
}

// what every entry point other than the inserts does first: the index as of all completed insert calls
static int enter(brisk_hip_index* h) { return flush_pending(h); }

int insert_packed_impl(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads) {
    for (u64 r0 = 0; r0 < n_reads; r0 += h->max_batch_reads) {
        const u64 nb = std::min<u64>(h->max_batch_reads, n_reads - r0);
        u64 n_rec = 0;
        int rc;
        bool done = false;
        if ((rc = defer_batch(h, d_packed, d_starts + r0, nb, &done))) return rc;
        if (done) continue;
        if ((rc = flush_pending(h))) return rc;  // (the direct paths use d_hist)
        if ((rc = insert_packed_binned(h, d_packed, d_starts + r0, nb, &done))) return rc;
        if (h->trace) fprintf(stderr, "[brisk_hip] path: direct insert of %llu reads, %s\n", (unsigned long long)nb, done ? "records binned by the scan" : "records to staging, then k_scatter");
        if (done) continue;
        bool hist_ok = true;
        if ((rc = scan_to_staging(h, d_packed, d_starts + r0, nb, true, false, &n_rec, &hist_ok))) return rc;
        if ((rc = insert_records_impl(h, (const u64*)h->staging.p, n_rec, hist_ok))) return rc;
    }
    return BRISK_HIP_OK;
}

// records (with a tag each) -> d_sums[tag] += sum of the counts of the record's k-mers that are present.
// d_hist holds the records' per-partition histogram; d_sums must be zeroed by the caller.
int query_records_impl(brisk_hip_index* h, const u64* d_rec, const u32* d_tags, u64 n_rec, unsigned long long* d_sums, const BinLayout* bl = nullptr);

int query_packed_impl(brisk_hip_index* h, const u32* d_packed, const u64* d_starts, u64 n_reads, unsigned long long* d_sums) {
    // d_sums[n_reads] must be zeroed by the caller
    u64 n_rec = 0;
    int rc;
    {  // one pass over the records, as in the insert: the scan bins them (and their reads' indices) by partition
        BinLayout bl{};
        bool binned = false;
        if ((rc = scan_binned(h, d_packed, d_starts, n_reads, true, &binned, &bl, &n_rec))) return rc;
        if (binned) return n_rec ? query_records_impl(h, nullptr, nullptr, n_rec, d_sums, &bl) : BRISK_HIP_OK;
    }
    bool hist_ok = true;
    if ((rc = scan_to_staging(h, d_packed, d_starts, n_reads, true, true, &n_rec, &hist_ok))) return rc;
    if (n_rec == 0) return BRISK_HIP_OK;
    if (!hist_ok) {  // long sequences were scanned in chunks and cut where the query stops: count the records that are left
        HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
        hipLaunchKernelGGL(k_part_hist, dim3(nblocks(n_rec, 256)), dim3(256), 0, h->stream, h->P, (const u64*)h->staging.p, n_rec, h->d_hist);
        if (int lrc = launch_check(h, "k_part_hist")) return lrc;
    }
    return query_records_impl(h, (const u64*)h->staging.p, (const u32*)h->tags_a.p, n_rec, d_sums);
}

// `bl`: the records (and bl->tags) lie in per-partition bins, the ones beyond them in bl->ovf (fast geometries only); else d_rec / d_tags
int query_records_impl(brisk_hip_index* h, const u64* d_rec, const u32* d_tags, u64 n_rec, unsigned long long* d_sums, const BinLayout* bl) {
    const BriskParams& P = h->P;
    h->scan_hist_valid = false;
    int rc;
    const u64 n_move = bl ? bl->n_ovf : n_rec;
    if (n_move && (rc = prefix_partitions(h, bl ? bl->bin_cap : 0u))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_small + 2, 0, 8, h->stream));
    if ((rc = list_touched(h, (u32*)(h->d_small + 2)))) return rc;
    if (n_move) {
        ProfScope ps(h, S_SCATTER);
        if ((rc = ensure(h, h->parted, n_move * P.stride * 8))) return rc;
        if ((rc = ensure(h, h->tags_b, n_move * 4))) return rc;
        const u64 n_slots = bl ? (u64)bl->ovf_region_cap * OVF_REGIONS : n_move;
        hipLaunchKernelGGL(k_scatter, dim3(nblocks(n_slots, 256)), dim3(256), 0, h->stream, P, bl ? (const u64*)bl->ovf : d_rec, n_move, h->d_cur32, (u64*)h->parted.p, 0,
                           bl ? bl->tags + h->n_parts * bl->bin_cap : d_tags, (u32*)h->tags_b.p, h->ix.err, bl ? bl->ovf_cnt : (const u32*)nullptr, bl ? bl->ovf_region_cap : 0u);
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small + 2, h->d_small + 2, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_small + 5, h->ix.cursor, 8, hipMemcpyDeviceToHost, h->stream));  // arena slots handed out so far
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u32 n_touched = (u32)h->h_small[2];
    h->arena_used_host = h->h_small[5];
    if (n_touched == 0) return BRISK_HIP_OK;
    if ((rc = ensure(h, h->desc, (size_t)n_touched * sizeof(PartDesc)))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_small + 3, 0, 8, h->stream));
    // partitions whose entries x instances would keep one wave busy for long go to k_query_huge, a workgroup each
    static const long hq_env = getenv("BRISK_HUGE_QUERY_AT") ? atol(getenv("BRISK_HUGE_QUERY_AT")) : -1;  // entries; tests: 0 sends (nearly) everything there
    const u32 hq_at = hq_env >= 0 ? (u32)hq_env : 4096u;
    if ((rc = ensure(h, h->huge, (size_t)(HUGE_LIST_CAP + 1) * 4))) return rc;
    HIPCHK(h, hipMemsetAsync(h->huge.p, 0, 4, h->stream));
    {
        ProfScope ps(h, S_TOUCHED);
        hipLaunchKernelGGL(k_need, dim3(std::min<u32>(nblocks(n_touched, 256), 2048)), dim3(256), 0, h->stream, h->d_hist, (bl && !n_move) ? (const u32*)nullptr : h->d_off,
                           h->d_touched, n_touched, h->ix.dir, (PartDesc*)h->desc.p, h->d_small + 3, bl ? bl->bin_cap : 0u, 0u, (u32*)h->huge.p, (u32)HUGE_LIST_CAP, hq_at,
                           hq_env >= 0 ? 0ull : 1ull << 26);
        if (int lrc = launch_check(h, "k_need")) return lrc;
    }
    {
        ProfScope ps(h, S_QUERY);
        HIPCHK(h, hipMemsetAsync(h->d_small + 6, 0, 8, h->stream));
        const u32 batches = (n_touched + WI_BATCH - 1) / WI_BATCH;
        const RecSrc src{bl ? bl->bins : (u64*)h->parted.p, (const u64*)h->parted.p, bl ? bl->bin_cap : 0u};
        const u32* tags_binned = bl ? bl->tags : nullptr;
        auto resident = [&](const void* fn) -> u32 {  // persistent waves: as many as the device keeps resident
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) cus = 256;
            return std::min<u32>((u32)per_cu * (u32)cus, INSERT_SLOTS);
        };
        // table chunks of 128 entries while the partitions average at most ~100 (a partition's arena slice is its entries + 25 %
        // + 8: 135 slots per partition), which keeps twice the waves resident; else 256
        static const long ent_env = getenv("BRISK_QUERY_ENT") ? atol(getenv("BRISK_QUERY_ENT")) : 0;  // experiments and tests: 128 | 256
        const bool small = ent_env ? ent_env == 128 : h->arena_used_host / own_partitions(h).len <= 135;
#define LAUNCH_QUERY_FAST_ENT(NW, KB, SH, ENT)                                                                                                                    \
    hipLaunchKernelGGL((k_query_fast<NW, KB, SH, ENT>), dim3(std::min<u32>(batches, resident((const void*)k_query_fast<NW, KB, SH, ENT>))), dim3(64), 0, h->stream, P, \
                       src, tags_binned, (const u32*)h->tags_b.p, (const PartDesc*)h->desc.p, n_touched, h->ix, d_sums, (u32*)(h->d_small + 6))
#define LAUNCH_QUERY_FAST(NW, KB, SH)                     \
    {                                                     \
        if (small) LAUNCH_QUERY_FAST_ENT(NW, KB, SH, 128); \
        else LAUNCH_QUERY_FAST_ENT(NW, KB, SH, 256);       \
    }
        static const bool generic_only = getenv("BRISK_QUERY_GENERIC") != nullptr;  // A/B and tests: force the run-time body
        const bool fast = bl || !generic_only;
        if (fast && P.nw == 3 && P.kb == 49 && P.shift == 4) LAUNCH_QUERY_FAST(3, 49, 4)       // k63 m21 b14
        else if (fast && P.nw == 3 && P.kb == 49 && P.shift == 3) LAUNCH_QUERY_FAST(3, 49, 3)  // (sharded, 2^25..2^27 partitions)
        else if (fast && P.nw == 3 && P.kb == 49 && P.shift == 2) LAUNCH_QUERY_FAST(3, 49, 2)
        else if (fast && P.nw == 3 && P.kb == 49 && P.shift == 1) LAUNCH_QUERY_FAST(3, 49, 1)
        else if (fast && P.nw == 2 && P.kb == 17 && P.shift == 4) LAUNCH_QUERY_FAST(2, 17, 4)  // k31 m15 b14 (apps/counter.cpp:355)
        else if (fast && P.nw == 2 && P.kb == 17 && P.shift == 5) LAUNCH_QUERY_FAST(2, 17, 5)
        else if (fast && P.nw == 2 && P.kb == 17 && P.shift == 6) LAUNCH_QUERY_FAST(2, 17, 6)
        else if (fast && P.nw == 2 && P.kb == 20 && P.shift == 0) LAUNCH_QUERY_FAST(2, 20, 0)  // k31 m11 b11
        else if (bl) return fail(h, BRISK_HIP_EHIP, "binned query without a kernel for this geometry");
        else
            hipLaunchKernelGGL(k_query, dim3(std::min<u32>(batches, INSERT_SLOTS)), dim3(64), 0, h->stream, P, (const u64*)h->parted.p, (const u32*)h->tags_b.p,
                               (const PartDesc*)h->desc.p, n_touched, h->ix, d_sums, (u32*)(h->d_small + 6));
#undef LAUNCH_QUERY_FAST
#undef LAUNCH_QUERY_FAST_ENT
        if (int lrc = launch_check(h, "k_query")) return lrc;
        // the listed partitions (none, as a rule: the kernel reads the list's length on the device and returns)
        hipLaunchKernelGGL(k_query_huge, dim3(256), dim3(HG_THREADS), 0, h->stream, P, src, tags_binned, (const u32*)h->tags_b.p, (const PartDesc*)h->desc.p, (const u32*)h->huge.p + 1,
                           (const u32*)h->huge.p, h->ix, d_sums);
    }
    return launch_check(h, "k_query_huge");
}

// nuc2int (Kmers.cpp:442-444) in bulk on the host: n ASCII bytes -> (n + 15) / 16 words of the packed stream, first nucleotide of a
// word in its top bits, the last word zero padded (the layout k_pack_ascii writes).  The format conversion at the boundary -- the
// upload threads do it in the pass over the caller's bytes that stages them into pinned memory, so that a quarter of the bytes
// cross PCIe -- not part of the path's arithmetic: everything from the packed stream on is device code.
static void host_pack_scalar(const char* s, u64 n, u32* out) {
    const u64 n_words = (n + 15) / 16;
    for (u64 w = 0; w < n_words; w++) {
        const u64 left = n - w * 16;
        u32 v = 0;
        for (u64 i = 0; i < 16; i++) v = (v << 2) | (i < left ? (((uint8_t)s[w * 16 + i] >> 1) & 3u) : 0u);
        out[w] = v;
    }
}
#ifdef BRISK_HOST_AVX2
__attribute__((target("avx2"))) static void host_pack_avx2(const char* s, u64 n, u32* out) {
    const u64 n32 = n / 32;  // 32 bytes -> two words
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i mul41 = _mm256_set1_epi16(0x0104);      // bytes (4, 1): n0 * 4 + n1 per byte pair
    const __m256i mul161 = _mm256_set1_epi32(0x00010010);  // 16-bit (16, 1): t0 * 16 + t1 = the byte of four nucleotides
    const __m256i pick = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    for (u64 i = 0; i < n32; i++) {
        __m256i v = _mm256_loadu_si256((const __m256i*)(s + 32 * i));
        v = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
        const __m256i t = _mm256_maddubs_epi16(v, mul41);
        const __m256i b = _mm256_madd_epi16(t, mul161);
        const __m256i w = _mm256_shuffle_epi8(b, pick);
        out[2 * i] = (u32)_mm256_extract_epi32(w, 0);
        out[2 * i + 1] = (u32)_mm256_extract_epi32(w, 4);
    }
    if (n32 * 32 < n) host_pack_scalar(s + n32 * 32, n - n32 * 32, out + 2 * n32);
}
#endif
static void host_pack(const char* s, u64 n, u32* out) {
#ifdef BRISK_HOST_AVX2
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return host_pack_avx2(s, n, out);
#endif
    host_pack_scalar(s, n, out);
}

// Host ASCII -> packed 2-bit stream on the device.  The caller's memory is pageable: a plain hipMemcpy moves it at
// 17-19 GB/s through the runtime's single staging path.  Here a few host threads take chunks into their own pinned
// buffers (two each, so a thread fills one while the copy engine drains the other); chunks are whole 16-nt words of the
// packed stream, so they are independent.  [r3] The threads pack while they stage (host_pack: the pass over the caller's
// bytes that used to be a memcpy) and the copy engine moves the packed words straight to their place; BRISK_HOST_PACK=0
// keeps the older route (ASCII over PCIe, k_pack_ascii where it lands).  Process-wide, one upload at a time.
struct UploadLane {
    hipStream_t st = nullptr;
    char* pin[2] = {nullptr, nullptr};
    char* dev[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
constexpr size_t kUploadChunk = (size_t)16 << 20;
static unsigned upload_lanes() {  // BRISK_UPLOAD_LANES: host threads of an upload (default 12 packing, 6 on the older route: more only fought over PCIe there)
    static const unsigned n = [] {
        const char* e = getenv("BRISK_UPLOAD_LANES");
        const long v = e ? atol(e) : 0;
        const bool hp = !(getenv("BRISK_HOST_PACK") && atoi(getenv("BRISK_HOST_PACK")) == 0);
        return (unsigned)(v >= 1 && v <= 64 ? v : hp ? 12 : 6);
    }();
    return n;
}
#define kUploadLanes upload_lanes()
std::mutex g_upload_mu;
std::vector<UploadLane> g_upload;
int g_upload_device = -1;

// `pipelined` (insert_reads_pipelined's uploader thread): the destination is free by the caller's bookkeeping and the handle's own
// stream belongs to the caller's thread -- only the lanes' streams are touched here, and an error text goes to *err_out, not h->err.
int upload_and_pack(brisk_hip_index* h, const char* src, u64 nb, u32* d_packed, u64 n_words, bool pipelined = false, std::string* err_out = nullptr) {
    const u64 n_chunks = (nb + kUploadChunk - 1) / kUploadChunk;
    static const bool host_packs = !(getenv("BRISK_HOST_PACK") && atoi(getenv("BRISK_HOST_PACK")) == 0);
    auto in_one_piece = [&]() -> int {  // plain copy of the whole input, then one pack launch
        int rc;
        if ((rc = ensure(h, h->bases_tmp, nb + 16))) return rc;
        if (nb) HIPCHK(h, hipMemcpyAsync(h->bases_tmp.p, src, nb, hipMemcpyHostToDevice, h->stream));
        if (n_words) {
            ProfScope ps(h, S_PACK);
            hipLaunchKernelGGL(k_pack_ascii, dim3(nblocks(n_words, 256)), dim3(256), 0, h->stream, (const uint8_t*)h->bases_tmp.p, nb, d_packed, n_words);
        }
        return launch_check(h, "k_pack_ascii");
    };
    if (n_chunks < 4 && !pipelined) return in_one_piece();  // small input: not worth the threads
    std::lock_guard<std::mutex> lk(g_upload_mu);
    if (g_upload_device != h->device) {  // first use on this device: the lanes stay for the life of the process
        for (UploadLane& l : g_upload)
            for (int i = 0; i < 2; i++) {
                if (l.pin[i]) hipHostFree(l.pin[i]);
                if (l.dev[i]) hipFree(l.dev[i]);
                if (l.ev[i]) hipEventDestroy(l.ev[i]);
            }
        for (UploadLane& l : g_upload)
            if (l.st) hipStreamDestroy(l.st);
        g_upload.assign(kUploadLanes, UploadLane{});
        g_upload_device = -1;
        bool ok = true;
        for (UploadLane& l : g_upload) {
            ok = ok && hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 2 && ok; i++)
                ok = hipHostMalloc((void**)&l.pin[i], host_packs ? kUploadChunk / 4 : kUploadChunk) == hipSuccess &&
                     (host_packs || hipMalloc((void**)&l.dev[i], kUploadChunk) == hipSuccess) && hipEventCreateWithFlags(&l.ev[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {  // no room for the pinned lanes (192 MiB of host, as much of device memory): the plain path still works
            (void)hipGetLastError();
            if (pipelined) {
                if (err_out) *err_out = "upload: no pinned lanes";
                return BRISK_HIP_ENOMEM;
            }
            return in_one_piece();
        }
        g_upload_device = h->device;
    }
    if (!pipelined) HIPCHK(h, hipStreamSynchronize(h->stream));  // d_packed may still be read by earlier work of this index
    std::vector<hipError_t> err(kUploadLanes, hipSuccess);
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < kUploadLanes; t++)
        workers.emplace_back([&, t]() {
            hipError_t e = hipSetDevice(h->device);
            UploadLane& l = g_upload[t];
            u32 turn = 0;
            for (u64 c = t; c < n_chunks && e == hipSuccess; c += kUploadLanes, turn ^= 1) {
                const u64 off = c * kUploadChunk, len = std::min<u64>(kUploadChunk, nb - off);
                e = hipEventSynchronize(l.ev[turn]);  // the copy that used this buffer two chunks ago has drained it
                if (e != hipSuccess) break;
                const u64 words = (len + 15) / 16;
                if (host_packs) {
                    host_pack(src + off, len, (u32*)l.pin[turn]);
                    e = hipMemcpyAsync(d_packed + off / 16, l.pin[turn], words * 4, hipMemcpyHostToDevice, l.st);
                    if (e != hipSuccess) break;
                } else {
                    memcpy(l.pin[turn], src + off, len);
                    e = hipMemcpyAsync(l.dev[turn], l.pin[turn], len, hipMemcpyHostToDevice, l.st);
                    if (e != hipSuccess) break;
                    hipLaunchKernelGGL(k_pack_ascii, dim3(nblocks(words, 256)), dim3(256), 0, l.st, (const uint8_t*)l.dev[turn], len, d_packed + off / 16, words);
                }
                e = hipEventRecord(l.ev[turn], l.st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(l.st);
            err[t] = e;
        });
    for (std::thread& w : workers) w.join();
    for (hipError_t e : err)
        if (e != hipSuccess) {
            if (err_out) {
                *err_out = std::string("upload: ") + hipGetErrorString(e);
                return BRISK_HIP_EHIP;
            }
            return fail(h, BRISK_HIP_EHIP, std::string("upload: ") + hipGetErrorString(e));
        }
    return BRISK_HIP_OK;
}

// ---- BRISK_VERIFY=1: stage checks of the host input path ---------------------------------------------------------------------
// Round 2's parity soak saw four wrong indexes in ~49,000 cases.  The one that was recorded (tests/fuzz_parity.py as it ran then,
// seed 102 case 739; tools/replay_fuzz_case.py rebuilds its inputs bit for bit on the CPU) is reproduced EXACTLY -- entry count,
// bucket count, every reported entry -- by changing ONE nucleotide of ONE read of the input (read 292, offset 843, T -> A: bit 25
// of word 4839 of the packed stream cleared, or the ASCII byte behind it) and by no change downstream of the scan: the 26 wrong
// k-mers are all the k-mers of that read that cover the nucleotide, spread over three super-k-mers, so the scan's step loop and
// its record builder both read the same wrong value.  The fault therefore lies between the caller's buffer and the packed stream
// the scan reads (host -> device copy, the ASCII staging buffer, k_pack_ascii's store, the packed words until the scan is done),
// not in the scan, the record hand-over or the insert.  These checks name the sub-stage should it happen again:
//   verify_upload   the packed stream on the device against the caller's bytes packed on the host; on a mismatch the ASCII staging
//                   buffer is compared with the caller's bytes too (copy or staging memory vs pack kernel or packed memory), and
//                   the word is read a second time (a value that changes between two reads was never stored wrong);
//                   run before the scan and again after the body (the stream must not change while it is being scanned)
//   verify_records  sum over a scan's records of their k-mer counts == the batch's k-mer instances (insert-mode scans)
static u32 host_pack16(const char* s, u64 n) {  // nuc2int (Kmers.cpp:442-444) over up to 16 bytes, first nt in the top bits, zero padded
    u32 v = 0;
    for (u64 i = 0; i < 16; i++) v = (v << 2) | (i < n ? (((uint8_t)s[i] >> 1) & 3u) : 0u);
    return v;
}
int verify_upload(brisk_hip_index* h, const char* src, u64 nb, const u32* d_packed, u64 n_words, const char* when) {
    if (!n_words) return BRISK_HIP_OK;
    std::vector<u32> dev(n_words);
    HIPCHK(h, hipMemcpy(dev.data(), d_packed, n_words * 4, hipMemcpyDeviceToHost));
    for (u64 w = 0; w < n_words; w++) {
        const u32 want = host_pack16(src + w * 16, nb - w * 16);
        if (dev[w] == want) continue;
        u32 again = 0;
        HIPCHK(h, hipMemcpy(&again, d_packed + w, 4, hipMemcpyDeviceToHost));
        std::string msg = std::string("BRISK_VERIFY: packed stream differs from the caller's bytes ") + when + ": word " + std::to_string(w) + " of " + std::to_string(n_words) +
                          ": host " + std::to_string(want) + ", device " + std::to_string(dev[w]) + ", device read again " + std::to_string(again);
        // the ASCII staging buffer still holds this batch when the input went in one piece (upload_and_pack); the lanes' chunk buffers do not
        if (h->bases_tmp.p && h->bases_tmp.bytes >= nb && (nb + kUploadChunk - 1) / kUploadChunk < 4) {
            const u64 b0 = w * 16, bn = std::min<u64>(16, nb - b0);
            char a[17] = {0};
            HIPCHK(h, hipMemcpy(a, (const char*)h->bases_tmp.p + b0, bn, hipMemcpyDeviceToHost));
            const bool ascii_ok = memcmp(a, src + b0, bn) == 0;
            msg += std::string("; ASCII staging bytes ") + (ascii_ok ? "EQUAL the caller's: the pack kernel's store or the packed words are at fault"
                                                                     : "DIFFER from the caller's: the host->device copy or the staging memory is at fault") +
                   " (device \"" + std::string(a, bn) + "\", host \"" + std::string(src + b0, bn) + "\")";
        }
        u64 n_bad = 0;
        for (u64 x = w; x < n_words; x++) n_bad += dev[x] != host_pack16(src + x * 16, nb - x * 16);
        msg += "; " + std::to_string(n_bad) + " words differ in all";
        fprintf(stderr, "[brisk_hip] %s\n", msg.c_str());
        return fail(h, BRISK_HIP_EHIP, msg);
    }
    return BRISK_HIP_OK;
}
// sum of hdr_n over n_rec records laid out back to back == want (insert-mode scans: every k-mer instance of the batch is in exactly one record)
int verify_records(brisk_hip_index* h, const u64* d_rec, u64 n_rec, u64 want, const char* what) {
    std::vector<u64> hdr(n_rec);
    if (n_rec) HIPCHK(h, hipMemcpy2D(hdr.data(), 8, d_rec + h->P.nw, h->P.stride * 8, 8, n_rec, hipMemcpyDeviceToHost));
    u64 sum = 0;
    for (u64 i = 0; i < n_rec; i++) sum += (hdr[i] >> 32) & 0xff;
    if (sum == want) return BRISK_HIP_OK;
    const std::string msg = std::string("BRISK_VERIFY: ") + what + ": the records hold " + std::to_string(sum) + " k-mers, the batch has " + std::to_string(want);
    fprintf(stderr, "[brisk_hip] %s\n", msg.c_str());
    return fail(h, BRISK_HIP_EHIP, msg);
}

// the per-partition histogram a scan left in d_hist: records and k-mer instances add up to what the scan reported / the batch holds
int verify_hist(brisk_hip_index* h, u64 want_rec, u64 want_inst, const char* what) {
    std::vector<unsigned long long> hist(h->n_parts);
    HIPCHK(h, hipMemcpy(hist.data(), h->d_hist, h->n_parts * 8, hipMemcpyDeviceToHost));
    u64 rec = 0, inst = 0;
    for (unsigned long long v : hist) {
        rec += v & 0xffffffffull;
        inst += v >> 32;
    }
    if (rec == want_rec && inst == want_inst) return BRISK_HIP_OK;
    const std::string msg = std::string("BRISK_VERIFY: ") + what + ": the histogram counts " + std::to_string(rec) + " records / " + std::to_string(inst) +
                            " k-mers, expected " + std::to_string(want_rec) + " / " + std::to_string(want_inst);
    fprintf(stderr, "[brisk_hip] %s\n", msg.c_str());
    return fail(h, BRISK_HIP_EHIP, msg);
}

// host ASCII reads -> device packed stream + starts, in pieces of at most max_bases
template <class F>
int for_each_host_batch(brisk_hip_index* h, const char* bases, const uint64_t* offsets, uint64_t n_reads, F&& body) {
    const u64 max_bases = 1ull << 33;  // only the packed stream (a quarter of it) and chunk buffers live on the device
    for (u64 r0 = 0; r0 < n_reads;) {
        // the longest run of reads from r0 within max_batch_reads and max_bases (at least one read)
        const u64 r_hi = std::min<u64>(n_reads, r0 + h->max_batch_reads);
        u64 r1 = (u64)(std::upper_bound(offsets + r0, offsets + r_hi + 1, offsets[r0] + max_bases) - offsets) - 1;
        if (r1 <= r0) r1 = r0 + 1;
        const u64 nb = offsets[r1] - offsets[r0];
        const u64 nr = r1 - r0;
        int rc;
        const auto t_0 = std::chrono::steady_clock::now();
        const u64 n_words = (nb + 15) / 16;
        if ((rc = ensure(h, h->packed_tmp, (n_words + 4) * 4))) return rc;
        if ((rc = ensure(h, h->starts_tmp, (nr + 1) * 8))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->starts_tmp.p, offsets + r0, (nr + 1) * 8, hipMemcpyHostToDevice, h->stream));
        if (offsets[r0]) {  // starts are relative to the piece
            hipLaunchKernelGGL(k_rebase, dim3(nblocks(nr + 1, 256)), dim3(256), 0, h->stream, (u64*)h->starts_tmp.p, nr + 1, (u64)offsets[r0]);
            if ((rc = launch_check(h, "k_rebase"))) return rc;
        }
        HIPCHK(h, hipMemsetAsync((char*)h->packed_tmp.p + n_words * 4, 0, 16, h->stream));
        static const bool dbg_up = getenv("BRISK_DEBUG_UPLOAD") != nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        if ((rc = upload_and_pack(h, bases + offsets[r0], nb, (u32*)h->packed_tmp.p, n_words))) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->verify && (rc = verify_upload(h, bases + offsets[r0], nb, (const u32*)h->packed_tmp.p, n_words, "before the scan"))) return rc;
        if (h->profiling) {
            h->prof_ms[S_UPLOAD] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count();
            h->prof_launches[S_UPLOAD]++;
        }
        if (dbg_up) fprintf(stderr, "[brisk_hip] upload: %llu bases in %.1f ms (offsets prepared in %.1f ms)\n", (unsigned long long)nb,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count(),
                            std::chrono::duration<double, std::milli>(t_a - t_0).count());
        if ((rc = body(r0, nr))) return rc;
        if (h->verify) {  // the packed stream must not have changed while it was scanned
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if ((rc = verify_upload(h, bases + offsets[r0], nb, (const u32*)h->packed_tmp.p, n_words, "after the scan"))) return rc;
        }
        r0 = r1;
    }
    return BRISK_HIP_OK;
}

// A big host batch in sub-batches of twelve upload chunks (192 MiB of ASCII, ~1.3 M reads of 150 bp): an uploader thread drives the
// pinned lanes for sub-batch j + 1 while this thread scans sub-batch j (small scans are deferred inserts: their records collect
// and go in together), two packed buffers taking turns.  Round 2 uploaded a whole call's bytes, then computed: 10 M reads took
// 30-32 ms of upload + 17 ms of scan and insert; the upload alone is what the link allows.
constexpr u64 kPipeChunks = 12;
int insert_reads_pipelined(brisk_hip_index* h, const char* bases, const uint64_t* offsets, uint64_t n_reads) {
    struct Sub {
        u64 r0, r1;
    };
    std::vector<Sub> subs;
    // pieces of at least kPipeChunks chunks, and no more than about eight of them: every piece pays the path's fixed costs (a dozen
    // launches and host synchronisations), and large pieces take the binned scan.  BRISK_PIPE_PIECES overrides the eight.
    static const u64 n_pieces = getenv("BRISK_PIPE_PIECES") && atol(getenv("BRISK_PIPE_PIECES")) > 0 ? (u64)atol(getenv("BRISK_PIPE_PIECES")) : 8;
    const u64 total = offsets[n_reads] - offsets[0];
    const u64 want = std::max<u64>(kPipeChunks * kUploadChunk, (total / n_pieces + kUploadChunk - 1) / kUploadChunk * kUploadChunk);
    for (u64 r0 = 0; r0 < n_reads;) {
        u64 r1 = (u64)(std::upper_bound(offsets + r0, offsets + n_reads + 1, offsets[r0] + want) - offsets) - 1;
        if (r1 <= r0) r1 = r0 + 1;
        r1 = std::min<u64>(r1, r0 + h->max_batch_reads);
        if (offsets[n_reads] - offsets[r1] < 4 * kUploadChunk && n_reads - r0 <= h->max_batch_reads) r1 = n_reads;  // (no small last piece: every piece takes the lanes)
        subs.push_back(Sub{r0, r1});
        r0 = r1;
    }
    u64 max_words = 0, max_reads = 0;
    for (const Sub& sb : subs) {
        max_words = std::max<u64>(max_words, (offsets[sb.r1] - offsets[sb.r0] + 15) / 16);
        max_reads = std::max<u64>(max_reads, sb.r1 - sb.r0);
    }
    int rc;
    DevBuf* pk[2] = {&h->packed_tmp, &h->packed_tmp2};
    DevBuf* st[2] = {&h->starts_tmp, &h->starts_tmp2};
    for (int i = 0; i < 2; i++) {
        if ((rc = ensure(h, *pk[i], (max_words + 4) * 4))) return rc;
        if ((rc = ensure(h, *st[i], (max_reads + 1) * 8))) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the buffers may still be read by earlier work of this index
    int up_rc = BRISK_HIP_OK;
    std::string up_err;
    double up_ms = 0.0;
    auto upload = [&](size_t j) {
        const auto t0 = std::chrono::steady_clock::now();
        const Sub sb = subs[j];
        const u64 nb = offsets[sb.r1] - offsets[sb.r0];
        up_rc = hipSetDevice(h->device) == hipSuccess ? upload_and_pack(h, bases + offsets[sb.r0], nb, (u32*)pk[j & 1]->p, (nb + 15) / 16, true, &up_err) : BRISK_HIP_EHIP;
        up_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    std::thread up(upload, (size_t)0);
    for (size_t j = 0; j < subs.size(); j++) {
        up.join();
        if (up_rc) return fail(h, up_rc, up_err.empty() ? "upload failed" : up_err);
        if (j + 1 < subs.size()) up = std::thread(upload, j + 1);  // ... while the device works on sub-batch j
        const Sub sb = subs[j];
        const u64 nr = sb.r1 - sb.r0, nb = offsets[sb.r1] - offsets[sb.r0], n_words = (nb + 15) / 16;
        auto body = [&]() -> int {
            HIPCHK(h, hipMemcpyAsync(st[j & 1]->p, offsets + sb.r0, (nr + 1) * 8, hipMemcpyHostToDevice, h->stream));
            if (offsets[sb.r0]) {
                hipLaunchKernelGGL(k_rebase, dim3(nblocks(nr + 1, 256)), dim3(256), 0, h->stream, (u64*)st[j & 1]->p, nr + 1, (u64)offsets[sb.r0]);
                if (int lrc = launch_check(h, "k_rebase")) return lrc;
            }
            HIPCHK(h, hipMemsetAsync((char*)pk[j & 1]->p + n_words * 4, 0, 16, h->stream));
            int brc;
            if (h->verify && (brc = verify_upload(h, bases + offsets[sb.r0], nb, (const u32*)pk[j & 1]->p, n_words, "before the scan (pipelined)"))) return brc;
            if ((brc = insert_packed_impl(h, (const u32*)pk[j & 1]->p, (const u64*)st[j & 1]->p, nr))) return brc;
            if (h->verify) {
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if ((brc = verify_upload(h, bases + offsets[sb.r0], nb, (const u32*)pk[j & 1]->p, n_words, "after the scan (pipelined)"))) return brc;
            }
            return BRISK_HIP_OK;
        };
        if ((rc = body())) {
            if (up.joinable()) up.join();  // (the uploader reads the caller's memory: it must be done before the call returns)
            return rc;
        }
    }
    if (h->profiling) {
        h->prof_ms[S_UPLOAD] += up_ms;
        h->prof_launches[S_UPLOAD] += subs.size();
    }
    return BRISK_HIP_OK;
}

int drain_profile(brisk_hip_index* h) {
    if (h->pending.empty()) return BRISK_HIP_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& pe : h->pending) {
        float ms = 0;
        hipEventElapsedTime(&ms, pe.a, pe.b);
        h->prof_ms[pe.slot] += ms;
        h->prof_launches[pe.slot]++;
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    h->pending.clear();
    return BRISK_HIP_OK;
}

void free_all(brisk_hip_index* h) {
    hipStreamSynchronize(h->stream);  // nothing of ours may be in flight when the arena is unmapped
    auto fr = [](void* p) { if (p) hipFree(p); };
    for (DevBuf* b : {&h->huge, &h->pend, &h->bins, &h->staging, &h->parted, &h->desc, &h->chunk_buf, &h->route_buf, &h->tags_a, &h->tags_b, &h->packed_tmp, &h->bases_tmp, &h->starts_tmp, &h->sums_tmp, &h->enum_out,
                      &h->lookup_buf, &h->packed_tmp2, &h->starts_tmp2})
        fr(b->p);
    fr(h->d_coef);
    fr(h->d_tabs);
    if (h->use_vmm) {
        pool_give(h->device, h->vm_keys, h->vm_counts, h->vm_ids);
    } else {
        fr(h->ix.keys);
        fr(h->ix.counts);
        fr(h->ix.ids);
    }
    fr(h->d_id_counter);
    fr(h->ix.err);
    fr(h->seq_buf.p);
    fr(h->ix.dir);
    fr(h->ix.cursor);
    fr(h->ix.bucket_bits);
    fr(h->ix.stats);
    fr(h->ix.slot_cur);
    fr(h->ix.slot_end);
    fr(h->d_hist);
    fr(h->d_ovf_cnt);
    fr(h->d_owner_cut);
    fr(h->d_off);
    fr(h->d_cur32);
    fr(h->d_touched);
    fr(h->d_block_sums);
    fr(h->d_small);
    if (h->h_small) hipHostFree(h->h_small);
    if (h->h_pin) hipHostFree(h->h_pin);
    for (auto& pe : h->pending) {
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
}

}  // namespace

// ===========================================================================
extern "C" {

#define BRISK_API __attribute__((visibility("default")))

BRISK_API uint32_t brisk_hip_abi_version(void) { return BRISK_HIP_ABI_VERSION; }

BRISK_API int brisk_hip_create(brisk_hip_index** out, uint8_t k, uint8_t m, uint8_t b, uint32_t data_bytes, const double* coef_table,
                               const brisk_hip_options* opt) {
    if (!out) return BRISK_HIP_EINVAL;
    *out = nullptr;
    // Parameters contract (parameters.hpp:19-22, Brisk.hpp:50-51, counter.cpp:32)
    if (!(b >= 1 && b <= m && m < k && k <= 63 && (m & 1) && m <= 31) || !coef_table) return BRISK_HIP_EINVAL;
    if (b > 16) return BRISK_HIP_EUNSUPPORTED;  // bucket ids are 32-bit, as the reference's uint32_t directory (DenseMenuYo.hpp:37,105)
    if (data_bytes != 1 && !(opt && opt->struct_size >= sizeof(brisk_hip_options) && opt->entry_ids)) return BRISK_HIP_EUNSUPPORTED;
    brisk_hip_options o{};
    if (opt) memcpy(&o, opt, std::min<size_t>(opt->struct_size ? opt->struct_size : sizeof(o), sizeof(o)));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || o.device < 0 || o.device >= ndev) return BRISK_HIP_ENODEVICE;
    brisk_hip_index* h = new (std::nothrow) brisk_hip_index();
    if (!h) return BRISK_HIP_ENOMEM;

    BriskParams& P = h->P;
    P.k = k; P.m = m; P.b = b;
    P.w = k - m;
    P.suff_reduc = (m - b + 1) / 2;
    P.kb = k - b;
    P.nw = (2 * (2 * k - m - b) + 63) / 64;
    P.stride = P.nw + 1;
    // Partitions: by default up to 2^24 of them whatever b is -- a small b leaves few buckets, so records are routed
    // by the bucket id plus ext_bits more bits of the same hashed minimizer (at most what it has: 2m - 2b, and what
    // the record header has room for).  An explicit part_bits keeps plain bucket ranges.
    // A minimizer of fewer than 12 nts has fewer than 24 hash bits to give: the partitions of such an index are few and
    // big (k=31 m=11 b=11: 4 M buckets holding what the default geometry spreads over 16 M partitions, and the most
    // frequent minimizers far more).  minimizer_idx is part of an entry's identity (SuperKmerLight.hpp:209-217), so its
    // class floor(minimizer_idx / cls_width) can extend the routing id like a hash bit does; the scan cuts a
    // super-k-mer's record where the class changes (brisk_scan.hip, emit_record_at).
    P.ext_bits = P.cls_bits = 0;
    P.cls_width = 1;
    if (!o.part_bits && 2u * b < 24u) {
        static const long cls_env = getenv("BRISK_CLS_BITS") ? atol(getenv("BRISK_CLS_BITS")) : -1;
        // One class bit: each one multiplies the records (k=31 m=11 b=11, 10 M reads: 120 M records whole, 230 M cut in two classes,
        // 320 M in four) and scan + scatter grow with them, while the insert has what it needs once the partitions fit its
        // LDS table: 48.3 ms per batch without, 34.9 with one bit, 47.0 with two, 73.0 with three (BRISK_CLS_BITS, profiles/r02_cls_sweep.txt).
        if (2u * m < 24u && P.w + 1 >= 8) P.cls_bits = cls_env >= 0 ? std::min<u32>((u32)cls_env, 3u) : 1u;
        const u32 from_hash = std::min<u32>(std::min<u32>(24u - 2u * b, 2u * (m - b)), 16u - P.cls_bits);
        P.ext_bits = from_hash + P.cls_bits;
        if (P.cls_bits) P.cls_width = (P.w + 1 + (1u << P.cls_bits) - 1) >> P.cls_bits;
    }
    const u32 rbits = 2u * b + P.ext_bits;
    P.part_bits = o.part_bits ? std::min<u32>(o.part_bits, 2u * b) : std::min<u32>(rbits, 24u);
    P.shift = rbits - P.part_bits;
    // the entry key [routing id low bits | compacted | idx'] must fit 128 bits
    while (P.shift + 2 * P.kb + 6 > 128 && P.shift > 0) { P.shift--; P.part_bits++; }
    if (P.shift + 2 * P.kb + 6 > 128 || P.part_bits > 30) {
        delete h;
        return BRISK_HIP_EUNSUPPORTED;
    }
    // an entry key of at most 64 bits is stored in one word (k31 b14: 44 bits, k31 b11: 46): 9 bytes per entry instead of 17
    // (the kernels with compile-time geometry decide by the same formula: insert_body, k_query_fast)
    h->ix.key_words = P.shift + 2 * P.kb + 6 <= 64 ? 1u : 2u;
    P.n_owners = o.n_owners ? o.n_owners : 1;
    P.owner_rank = o.owner_rank;
    if (P.owner_rank >= P.n_owners) { delete h; return BRISK_HIP_EINVAL; }
    if (P.n_owners > ROUTE_MAX_OWNERS) { delete h; return BRISK_HIP_EUNSUPPORTED; }  // routing keeps per-owner cursors in LDS
    P.m_mask = (1ull << (2 * m)) - 1;
    P.bucket_mask = (1ull << (2 * b)) - 1;
    h->n_parts = 1ull << P.part_bits;
    h->n_buckets = 1ull << (2 * b);
    h->max_batch_reads = o.max_batch_reads ? o.max_batch_reads : (1ull << 26);
    h->entry_ids = o.entry_ids != 0;
    static const bool defer_env = !(getenv("BRISK_DEFER") && atoi(getenv("BRISK_DEFER")) == 0);  // BRISK_DEFER=0: every insert call completes before it returns
    h->defer = defer_env && !o.immediate_inserts;
    {   // read per handle, not once per process: a soak switches it case by case
        const char* v = getenv("BRISK_VERIFY");
        h->verify = v && v[0] == '1';
        const char* t = getenv("BRISK_TRACE");
        h->trace = t && t[0] == '1';
    }
    h->device = o.device;

    auto init = [&]() -> int {
        HIPCHK(h, hipSetDevice(h->device));
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(h, BRISK_HIP_ENODEVICE, std::string("not a gfx950 device: ") + prop.gcnArchName);
        if (o.stream) h->stream = (hipStream_t)o.stream;
        else { HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
        HIPCHK(h, hipMalloc((void**)&h->d_coef, 128 * sizeof(double)));
        HIPCHK(h, hipMemsetAsync(h->d_coef, 0, 128 * sizeof(double), h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_coef, coef_table, 4 * m * sizeof(double), hipMemcpyHostToDevice, h->stream));
        {
            // decycling chunk tables (host, from the caller's libm table): cls_nch(m) chunks of cls_width(m, c) nts, both sums
            // of a chunk in one u64 (layout and decoding: decy_class_fast in brisk_scan.hip).
            // R(x)      = sum_{i=0}^{m-2} coef[4(m-1-i) + nt_i(x)]          (Decycling.cpp:17-24)
            // R(rot(x)) = sum_{i=1}^{m-1} coef[4(m-i)   + nt_i(x)]          (rot: Decycling.cpp:41)
            // word = a + b * 2^32 as one signed 64-bit integer, a / b = the chunk's share of R / R(rot) in units of 2^-24
            ScanCfg& c = h->scfg;
            c.nlow = std::min<u32>(32, k);
            c.nlow1 = std::min<u32>(32, k - 1);
            c.nch = cls_nch(m);
            c.qcap = 320;  // > 4 super-k-mers per lane before a mid-read flush (less where the tables need the LDS, below)
            if (c.nch > CLS_MAX_CHUNKS) return fail(h, BRISK_HIP_EUNSUPPORTED, "minimizer too long for the class tables");
            const u32 n_tab = 128 + cls_base(m, c.nch);
            c.n_tab = n_tab;
            std::vector<double> tabs(n_tab, 0.0);
            for (u32 i = 0; i < 4u * m; i++) tabs[i] = coef_table[i];
            for (u32 ch = 0; ch < c.nch; ch++) {
                const u32 off = cls_off(m, ch), wd = cls_width(m, ch), base = cls_base(m, ch);
                c.chunk[ch] = (2 * off) | ((2 * wd) << 6) | (base << 10);
                for (u32 v = 0; v < (1u << (2 * wd)); v++) {
                    double a = 0.0, bsum = 0.0;
                    for (u32 i = off; i < off + wd; i++) {
                        const u32 nt = (v >> (2 * (i - off))) & 3;
                        if (i + 1 < m) a += coef_table[4 * (m - 1 - i) + nt];
                        if (i >= 1) bsum += coef_table[4 * (m - i) + nt];
                    }
                    const int64_t af = llround(std::ldexp(a, 24)), bf = llround(std::ldexp(bsum, 24));
                    const u64 word = (u64)(af + bf * (int64_t(1) << 32));
                    memcpy(&tabs[128 + base + v], &word, 8);
                }
            }
            HIPCHK(h, hipMalloc((void**)&h->d_tabs, n_tab * sizeof(double)));
            HIPCHK(h, hipMemcpyAsync(h->d_tabs, tabs.data(), n_tab * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            // LDS: the class tables once per block, an emit queue per wave.  Eight waves per block, two blocks per CU.
            const size_t fixed = (size_t)(n_tab + 9) * 8;
            const size_t lds_max = 160 * 1024;
            // as many 8-wave blocks per CU as the kernel's registers allow (SCAN_WAVES_PER_EU waves per SIMD), the queues
            // shortened if that is what it takes (not below 224 entries); if the tables are too big for that, fewer blocks
            const bool spec = (k == 63 && m == 21) || (k == 31 && m == 15) || (k == 31 && m == 11);  // launch_scan's instantiations with k and m as constants
            u32 blocks_per_cu = std::max<u32>(1, (u32)(scan_waves_per_eu(spec ? (int)k : 0, spec ? (int)m : 0) * 4) / 8);
            auto lds_of = [&](u32 q, u32 waves) { return fixed + waves * (size_t)(q + 96) * 8; };  // queue + the lanes' read starts and tags
            while (blocks_per_cu > 1 && blocks_per_cu * lds_of(224, 8) > lds_max - 2048) blocks_per_cu--;
            while (c.qcap > 224 && blocks_per_cu * lds_of(c.qcap, 8) > lds_max - 2048) c.qcap -= 32;
            if (const char* e = getenv("BRISK_SCAN_QCAP")) c.qcap = (u32)std::max(128, atoi(e));
            const size_t per_wave = (size_t)(c.qcap + 96) * 8;
            u32 wv = 8;  // 4 waves per SIMD and block; block sizes that are not a multiple of 4 waves place badly (10-wave blocks: 49 ms against 31)
            if (lds_of(c.qcap, 8) > lds_max) wv = (u32)std::min<size_t>(16, (lds_max - fixed) / per_wave);
            if (const char* e = getenv("BRISK_SCAN_WAVES")) wv = (u32)std::min(16, std::max(1, atoi(e)));
            h->scan_waves = wv;
            h->scan_lds = fixed + wv * per_wave;
            // the attribute is per function, not per handle: never lower it for a handle created earlier
            static size_t lds_attr = 0;
            if (h->scan_lds > lds_attr) {
                lds_attr = h->scan_lds;
#define SCAN2_FNS(NCH, KK, MM) (const void*)k_scan2<NCH, 0, KK, MM>, (const void*)k_scan2<NCH, 1, KK, MM>, (const void*)k_scan2<NCH, 2, KK, MM>
                const void* fns[] = {SCAN2_FNS(0, 0, 0), SCAN2_FNS(3, 0, 0), SCAN2_FNS(4, 0, 0), SCAN2_FNS(6, 0, 0), SCAN2_FNS(0, 31, 11), SCAN2_FNS(0, 31, 15), SCAN2_FNS(0, 63, 21),
                                     (const void*)k_debug_keys};
#undef SCAN2_FNS
                for (const void* fn : fns) HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_attr));
            }
            const char* v1 = getenv("BRISK_SCAN_V1");
            h->scan_v1 = v1 && v1[0] == '1';
        }
        {   // the persistent insert kernels: how many waves the device keeps resident, at most
            const void* fns[] = {(const void*)k_insert<0, 0, 0>, (const void*)k_insert_big<0, 0, 0>, (const void*)k_insert_fast<3, 49, 4>, (const void*)k_insert_fast<2, 17, 4>,
                                 (const void*)k_insert_fast<2, 20, 0>, (const void*)k_insert_big<3, 49, 4>, (const void*)k_insert_big<2, 17, 4>, (const void*)k_insert_big<2, 20, 0>};
            int most = 16;
            for (const void* fn : fns) {
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) == hipSuccess && per_cu > most) most = per_cu;
            }
            h->insert_waves = std::min<u32>((u32)most * (u32)prop.multiProcessorCount, INSERT_SLOTS);
        }
        const u64 np = h->n_parts;
        HIPCHK(h, hipMalloc((void**)&h->ix.dir, np * sizeof(DirEnt)));
        HIPCHK(h, hipMalloc((void**)&h->ix.cursor, 8));
        HIPCHK(h, hipMalloc((void**)&h->ix.stats, 64));
        HIPCHK(h, hipMalloc((void**)&h->ix.slot_cur, INSERT_SLOTS * 8));
        HIPCHK(h, hipMalloc((void**)&h->ix.slot_end, INSERT_SLOTS * 8));
        HIPCHK(h, hipMemsetAsync(h->ix.slot_cur, 0, INSERT_SLOTS * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->ix.slot_end, 0, INSERT_SLOTS * 8, h->stream));
        const u64 bit_words = (h->n_buckets + 31) / 32;
        HIPCHK(h, hipMalloc((void**)&h->ix.bucket_bits, bit_words * 4));
        HIPCHK(h, hipMemsetAsync(h->ix.dir, 0, np * sizeof(DirEnt), h->stream));
        HIPCHK(h, hipMemsetAsync(h->ix.cursor, 0, 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->ix.stats, 0, 64, h->stream));
        HIPCHK(h, hipMemsetAsync(h->ix.bucket_bits, 0, bit_words * 4, h->stream));
        HIPCHK(h, hipMalloc((void**)&h->d_hist, (np + 1) * 8));
        HIPCHK(h, hipMalloc((void**)&h->d_ovf_cnt, OVF_REGIONS * 4));
        HIPCHK(h, hipMalloc((void**)&h->d_off, (np + 1) * 4));
        HIPCHK(h, hipMalloc((void**)&h->d_cur32, np * 4));
        HIPCHK(h, hipMalloc((void**)&h->d_touched, np * 4));
        h->n_scan_blocks = nblocks(np, 256 * SCAN_ITEMS);
        HIPCHK(h, hipMalloc((void**)&h->d_block_sums, (size_t)(h->n_scan_blocks + 1) * 4));
        h->ix.bits_check = 2u * b < 20u ? 1u : 0u;
        HIPCHK(h, hipMalloc((void**)&h->ix.err, 8));
        HIPCHK(h, hipMemsetAsync(h->ix.err, 0, 8, h->stream));
        HIPCHK(h, hipMalloc((void**)&h->d_id_counter, 8));
        HIPCHK(h, hipMemsetAsync(h->d_id_counter, 0, 8, h->stream));
        HIPCHK(h, hipMalloc((void**)&h->d_small, 64));
        HIPCHK(h, hipHostMalloc((void**)&h->h_small, 64));
        HIPCHK(h, hipHostMalloc((void**)&h->h_pin, kPinBytes));
        HIPCHK(h, hipMemsetAsync(h->d_small, 0, 64, h->stream));
        {
            // reserve virtual ranges as large as the device's memory; physical pages follow demand
            size_t free_b = 0, total_b = 0;
            if (const char* lim = getenv("BRISK_ARENA_LIMIT")) h->arena_limit = strtoull(lim, nullptr, 10);
            const char* novmm = getenv("BRISK_NO_VMM");
            if (!(novmm && novmm[0] == '1') && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
                const u64 max_entries = total_b / 17;
                const bool pooled = pool_take(h->device, h->entry_ids, h->vm_keys, h->vm_counts, h->vm_ids);
                if ((pooled || (vm_reserve(h, h->vm_keys, max_entries * 16) == BRISK_HIP_OK && vm_reserve(h, h->vm_counts, max_entries) == BRISK_HIP_OK)) &&
                    (!h->entry_ids || h->vm_ids.base || vm_reserve(h, h->vm_ids, max_entries * 4) == BRISK_HIP_OK)) {
                    h->use_vmm = true;
                } else {
                    vm_free(h->vm_keys);
                    vm_free(h->vm_counts);
                    vm_free(h->vm_ids);
                    h->err.clear();
                    fprintf(stderr, "[brisk_hip] create: no virtual range of %llu GiB for the arena (%llu GiB of address space are held by retired arenas): "
                                    "falling back to copy-on-growth allocations\n",
                            (unsigned long long)(max_entries * 17 >> 30), (unsigned long long)(g_retired_va.load() >> 30));
                }
            }
        }
        if (o.arena_entries) {
            int rc = ensure_arena(h, o.arena_entries);
            if (rc) return rc;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return BRISK_HIP_OK;
    };
    int rc = init();
    if (rc != BRISK_HIP_OK) {
        fprintf(stderr, "brisk_hip_create: %s\n", h->err.c_str());
        free_all(h);
        delete h;
        return rc;
    }
    *out = h;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_destroy(brisk_hip_index* h) {
    if (!h) return BRISK_HIP_EINVAL;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_clear(brisk_hip_index* h) {
    if (!h) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    const u64 np = h->n_parts;
    HIPCHK(h, hipMemsetAsync(h->ix.dir, 0, np * sizeof(DirEnt), h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.cursor, 0, 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_id_counter, 0, 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.err, 0, 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.stats, 0, 64, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.slot_cur, 0, INSERT_SLOTS * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.slot_end, 0, INSERT_SLOTS * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->ix.bucket_bits, 0, ((h->n_buckets + 31) / 32) * 4, h->stream));
    h->arena_used_host = 0;
    h->nb_skmers = 0;
    h->dir_snapshot_valid = false;
    h->n_pend = 0;  // records scanned and not yet inserted go with the index
    h->pend_hist_ok = true;
    return BRISK_HIP_OK;
}

BRISK_API const char* brisk_hip_last_error(const brisk_hip_index* h) { return h ? h->err.c_str() : "null handle"; }

static int check_device_flags(brisk_hip_index* h) {
    u32 e = 0;
    HIPCHK(h, hipMemcpyAsync(&e, h->ix.err, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (e) return fail(h, BRISK_HIP_EHIP, "device-side consistency check failed, flags=" + std::to_string(e) +
                                              " (1 scatter slot out of range, 2 arena exhausted, 4 chunk overflow): the index is not valid");
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_sync(brisk_hip_index* h) {
    if (!h) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    return check_device_flags(h);
}

BRISK_API int brisk_hip_get_layout(const brisk_hip_index* h, brisk_hip_layout* out) {
    if (!h || !out) return BRISK_HIP_EINVAL;
    const BriskParams& P = h->P;
    out->k = P.k; out->m = P.m; out->b = P.b;
    out->m_reduc = P.m - P.b;
    out->compacted_size = P.kb;
    out->allocated_bytes = (2 * P.k - P.m - P.b + 3) / 4;  // parameters.hpp:31
    out->record_words = P.stride;
    out->part_bits = P.part_bits;
    out->ext_bits = P.ext_bits;
    out->cls_bits = P.cls_bits;
    out->cls_width = P.cls_width;
    out->n_owners = P.n_owners;
    out->owner_rank = P.owner_rank;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_insert_packed(brisk_hip_index* h, const uint32_t* d_packed, const uint64_t* d_starts, uint64_t n_reads) {
    if (!h || (n_reads && (!d_packed || !d_starts))) return BRISK_HIP_EINVAL;
    if (h->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "insert_packed on a sharded index: use scan/route/insert_records");
    if (h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "bulk count on an entry-id index");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    return insert_packed_impl(h, d_packed, d_starts, n_reads);
}

BRISK_API int brisk_hip_insert_reads(brisk_hip_index* h, const char* bases, const uint64_t* offsets, uint64_t n_reads) {
    if (!h || (n_reads && (!bases || !offsets))) return BRISK_HIP_EINVAL;
    if (h->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "insert_reads on a sharded index: use scan/route/insert_records");
    if (h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "bulk count on an entry-id index");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    static const bool no_pipe = getenv("BRISK_UPLOAD_PIPELINE") && atoi(getenv("BRISK_UPLOAD_PIPELINE")) == 0;  // A/B and tests
    if (!no_pipe && n_reads && offsets[n_reads] - offsets[0] >= (kPipeChunks + 4) * kUploadChunk) return insert_reads_pipelined(h, bases, offsets, n_reads);
    return for_each_host_batch(h, bases, offsets, n_reads, [&](u64, u64 nr) {
        return insert_packed_impl(h, (const u32*)h->packed_tmp.p, (const u64*)h->starts_tmp.p, nr);
    });
}

BRISK_API int brisk_hip_get_reads(brisk_hip_index* h, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t* per_read_sum) {
    if (!h || (n_reads && (!bases || !offsets || !per_read_sum))) return BRISK_HIP_EINVAL;
    if (h->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "get_reads on a sharded index sees one bucket range only: use scan_query / route_tagged / query_records");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    return for_each_host_batch(h, bases, offsets, n_reads, [&](u64 r0, u64 nr) -> int {
        int rc;
        if ((rc = ensure(h, h->sums_tmp, nr * 8))) return rc;
        HIPCHK(h, hipMemsetAsync(h->sums_tmp.p, 0, nr * 8, h->stream));
        if ((rc = query_packed_impl(h, (const u32*)h->packed_tmp.p, (const u64*)h->starts_tmp.p, nr, (unsigned long long*)h->sums_tmp.p))) return rc;
        HIPCHK(h, hipMemcpyAsync(per_read_sum + r0, h->sums_tmp.p, nr * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return BRISK_HIP_OK;
    });
}

BRISK_API int brisk_hip_get_packed(brisk_hip_index* h, const uint32_t* d_packed, const uint64_t* d_starts, uint64_t n_reads, uint64_t* d_per_read_sum) {
    if (!h || (n_reads && (!d_packed || !d_starts || !d_per_read_sum))) return BRISK_HIP_EINVAL;
    if (h->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "get_packed on a sharded index sees one bucket range only: use scan_query / route_tagged / query_records");
    if (!n_reads) return BRISK_HIP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    HIPCHK(h, hipMemsetAsync(d_per_read_sum, 0, n_reads * 8, h->stream));
    for (u64 r0 = 0; r0 < n_reads; r0 += h->max_batch_reads) {  // the record tags of a batch are indices into its own slice of the sums
        const u64 nb = std::min<u64>(h->max_batch_reads, n_reads - r0);
        if (int rc = query_packed_impl(h, d_packed, d_starts + r0, nb, (unsigned long long*)d_per_read_sum + r0)) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_device_flags(h);
}

BRISK_API int brisk_hip_lookup(brisk_hip_index* h, const uint64_t* kmer_lo, const uint64_t* kmer_hi, const uint8_t* minimizer_idx, uint64_t n,
                               uint8_t* out_data, uint8_t* out_found) {
    if (!h || (n && (!kmer_lo || !kmer_hi || !minimizer_idx || !out_data || !out_found))) return BRISK_HIP_EINVAL;
    if (h->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "lookup on a sharded index sees one bucket range only: use scan_query / route_tagged / query_records");
    if (n == 0) return BRISK_HIP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    int rc;
    if ((rc = ensure(h, h->lookup_buf, n * 19 + 64))) return rc;
    char* base = (char*)h->lookup_buf.p;
    u64* d_lo = (u64*)base;
    u64* d_hi = (u64*)(base + n * 8);
    uint8_t* d_idx = (uint8_t*)(base + n * 16);
    uint8_t* d_data = d_idx + n;
    uint8_t* d_found = d_data + n;
    HIPCHK(h, hipMemcpyAsync(d_lo, kmer_lo, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_hi, kmer_hi, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_idx, minimizer_idx, n, hipMemcpyHostToDevice, h->stream));
    {
        ProfScope ps(h, S_LOOKUP);
        hipLaunchKernelGGL(k_lookup, dim3(nblocks(n * 64, 256)), dim3(256), 0, h->stream, h->P, h->ix, d_lo, d_hi, d_idx, n, d_data, d_found, (u32*)nullptr);
        if ((rc = launch_check(h, "k_lookup"))) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(out_data, d_data, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(out_found, d_found, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BRISK_HIP_OK;
}

static int enumerate_impl(brisk_hip_index* h, uint64_t* cursor, uint64_t* out_lo, uint64_t* out_hi, uint8_t* out_minimizer_idx,
                          uint8_t* out_data, uint32_t* out_ids, uint64_t cap, uint64_t* n_out) {
    if (!h || !cursor || !n_out || (cap && (!out_lo || !out_hi || !out_minimizer_idx))) return BRISK_HIP_EINVAL;
    if (out_ids && !h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "not an entry-id index");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    *n_out = 0;
    if (*cursor == 0 || !h->dir_snapshot_valid) {
        h->h_dir_cnt.resize(h->n_parts);
        // d_cur32 is per-batch scratch, free between batches
        hipLaunchKernelGGL(k_dir_counts, dim3(nblocks(h->n_parts, 256)), dim3(256), 0, h->stream, h->ix.dir, h->n_parts, h->d_cur32);
        if (int lrc = launch_check(h, "k_dir_counts")) return lrc;
        HIPCHK(h, hipMemcpyAsync(h->h_dir_cnt.data(), h->d_cur32, h->n_parts * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->dir_snapshot_valid = true;
    }
    u64 p = *cursor;
    while (p < h->n_parts && h->h_dir_cnt[p] == 0) p++;
    if (p >= h->n_parts) { *cursor = h->n_parts; return BRISK_HIP_OK; }
    std::vector<u64> base;
    u64 total = 0, q = p;
    while (q < h->n_parts && total + h->h_dir_cnt[q] <= cap) {
        base.push_back(total);
        total += h->h_dir_cnt[q];
        q++;
    }
    if (q == p) return fail(h, BRISK_HIP_ECAPACITY, "enumerate: cap smaller than one partition");
    const u64 np = q - p;
    int rc;
    if ((rc = ensure(h, h->enum_out, np * 8 + total * 22 + 64))) return rc;
    char* b0 = (char*)h->enum_out.p;
    u64* d_base = (u64*)b0;
    u64* d_lo = (u64*)(b0 + np * 8);
    u64* d_hi = d_lo + total;
    u32* d_ids = out_ids ? (u32*)(d_hi + total) : nullptr;
    uint8_t* d_idx = (uint8_t*)(d_hi + total) + total * 4;
    uint8_t* d_cnt = d_idx + total;
    HIPCHK(h, hipMemcpyAsync(d_base, base.data(), np * 8, hipMemcpyHostToDevice, h->stream));
    if (total) {
        ProfScope ps(h, S_ENUM);
        hipLaunchKernelGGL(k_enumerate, dim3((u32)std::min<u64>(np, 1u << 22)), dim3(64), 0, h->stream, h->P, h->ix, (u32)p, (u32)np, d_base, d_lo, d_hi, d_idx, d_cnt, d_ids);
        if ((rc = launch_check(h, "k_enumerate"))) return rc;
        HIPCHK(h, hipMemcpyAsync(out_lo, d_lo, total * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(out_hi, d_hi, total * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(out_minimizer_idx, d_idx, total, hipMemcpyDeviceToHost, h->stream));
        if (out_data) HIPCHK(h, hipMemcpyAsync(out_data, d_cnt, total, hipMemcpyDeviceToHost, h->stream));
        if (out_ids) HIPCHK(h, hipMemcpyAsync(out_ids, d_ids, total * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *cursor = q;
    *n_out = total;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_enumerate(brisk_hip_index* h, uint64_t* cursor, uint64_t* out_lo, uint64_t* out_hi, uint8_t* out_minimizer_idx,
                                  uint8_t* out_data, uint64_t cap, uint64_t* n_out) {
    if (cap && !out_data) return BRISK_HIP_EINVAL;
    return enumerate_impl(h, cursor, out_lo, out_hi, out_minimizer_idx, out_data, nullptr, cap, n_out);
}

BRISK_API int brisk_hip_enumerate_ids(brisk_hip_index* h, uint64_t* cursor, uint64_t* out_lo, uint64_t* out_hi, uint8_t* out_minimizer_idx,
                                      uint32_t* out_ids, uint64_t cap, uint64_t* n_out) {
    if (cap && !out_ids) return BRISK_HIP_EINVAL;
    return enumerate_impl(h, cursor, out_lo, out_hi, out_minimizer_idx, nullptr, out_ids, cap, n_out);
}

BRISK_API int brisk_hip_stats(brisk_hip_index* h, uint64_t* nb_buckets, uint64_t* nb_skmers, uint64_t* nb_kmers, uint64_t* memory_bytes,
                              uint64_t* largest_bucket) {
    if (!h) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    {
        int rcf = check_device_flags(h);
        if (rcf) return rcf;
    }
    // nb_kmers / nb_buckets / largest are reductions over the directory and the bucket bitmap
    const u64 bit_words = (h->n_buckets + 31) / 32;
    if (h->P.shift > 6) {  // partitions wider than 64 buckets: rebuild the bitmap from the entries
        HIPCHK(h, hipMemsetAsync(h->ix.bucket_bits, 0, bit_words * 4, h->stream));
        hipLaunchKernelGGL(k_bucket_bits, dim3((u32)std::min<u64>(h->n_parts, 4096)), dim3(256), 0, h->stream, h->P, h->ix, (u32)h->n_parts);
        if (int lrc = launch_check(h, "k_bucket_bits")) return lrc;
    }
    HIPCHK(h, hipMemsetAsync(h->ix.stats, 0, 24, h->stream));
    hipLaunchKernelGGL(k_stats, dim3(1024), dim3(256), 0, h->stream, h->ix.dir, h->n_parts, h->ix.bucket_bits, bit_words, h->ix.stats);
    if (int lrc = launch_check(h, "k_stats")) return lrc;
    unsigned long long st[4];
    HIPCHK(h, hipMemcpyAsync(st, h->ix.stats, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (nb_kmers) *nb_kmers = st[0];
    if (nb_buckets) *nb_buckets = st[1];
    if (largest_bucket) *largest_bucket = st[2];
    if (nb_skmers) *nb_skmers = h->nb_skmers;
    if (memory_bytes) {
        u64 m = h->arena_cap * (8ull * h->ix.key_words + 1) + h->n_parts * 16 + (h->n_buckets + 7) / 8 + (h->n_parts + 1) * 20;
        for (const DevBuf* b : {&h->bins, &h->staging, &h->parted, &h->desc, &h->chunk_buf, &h->route_buf, &h->tags_a, &h->tags_b, &h->packed_tmp, &h->bases_tmp, &h->starts_tmp, &h->sums_tmp,
                                &h->enum_out, &h->lookup_buf, &h->pend, &h->huge, &h->seq_buf, &h->packed_tmp2, &h->starts_tmp2})
            m += b->bytes;
        *memory_bytes = m;
    }
    return BRISK_HIP_OK;
}

// Brisk::reallocate (brisk/Brisk.hpp:202-224): every entry of `from` is moved into `to`, an empty index over the same k with
// its own (m, b) -- the reference re-buckets to (m + 2, b + 2).  The reference's loop calls a four-argument update_kmer that
// no file defines (the member template is never instantiated, its call site is commented out, brisk/Brisk.hpp:124-129), so
// what "the k-mer under the new m" means is taken from the path itself: the (kmer_s, minimizer_idx) that
// SuperKmerEnumerator yields for the k-mer as a sequence of k nts at the new m -- what would be there had the index been
// built at the new parameters.  On the device: entries -> reads of k nts -> the scan at the new parameters (one record
// of one k-mer per read) -> each record takes its entry's count as its multiplicity -> the insert.  Entries of `from`
// that the new minimizer maps to one identity (the same canonical k-mer stored under several identities)
// merge, their counts added mod 256.  `from` is left untouched.
BRISK_API int brisk_hip_reallocate(brisk_hip_index* from, brisk_hip_index* to) {
    if (!from || !to || from == to) return BRISK_HIP_EINVAL;
    std::lock(from->call_mu, to->call_mu);  // both or neither: reallocate(a, b) and reallocate(b, a) from two threads must not wait for each other
    std::lock_guard<std::recursive_mutex> lock_from(from->call_mu, std::adopt_lock);
    std::lock_guard<std::recursive_mutex> call_lock(to->call_mu, std::adopt_lock);
    brisk_hip_index* h = to;
    if (from->P.k != to->P.k || from->device != to->device) return fail(h, BRISK_HIP_EINVAL, "reallocate: both indexes must have the same k and device");
    if (from->entry_ids || to->entry_ids) return fail(h, BRISK_HIP_EINVAL, "reallocate: entry-id indexes are re-bucketed by the facade (DATA lives on the host)");
    if (to->P.n_owners > 1 || from->P.n_owners > 1) return fail(h, BRISK_HIP_EINVAL, "reallocate on a sharded index");
    HIPCHK(h, hipSetDevice(h->device));
    if (int frc = enter(from)) return fail(h, frc, "reallocate: " + from->err);
    if (int frc = enter(to)) return frc;
    {   // `to` must be empty (the reference re-buckets into a fresh DenseMenuYo, brisk/Brisk.hpp:205): counts would silently merge otherwise
        unsigned long long used = 0;
        HIPCHK(h, hipMemcpyAsync(&used, to->ix.cursor, 8, hipMemcpyDeviceToHost, to->stream));
        HIPCHK(h, hipStreamSynchronize(to->stream));
        if (used || to->arena_used_host || to->nb_skmers) return fail(h, BRISK_HIP_EINVAL, "reallocate: the target index is not empty");
    }
    HIPCHK(h, hipStreamSynchronize(from->stream));
    const u32 k = from->P.k;
    // the old index's partition sizes
    std::vector<u32> cnt(from->n_parts);
    hipLaunchKernelGGL(k_dir_counts, dim3(nblocks(from->n_parts, 256)), dim3(256), 0, h->stream, from->ix.dir, from->n_parts, from->d_cur32);
    if (int lrc = launch_check(h, "k_dir_counts")) return lrc;
    HIPCHK(h, hipMemcpyAsync(cnt.data(), from->d_cur32, from->n_parts * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u64 round_cap = 1ull << 24;  // entries per round
    std::vector<u64> base;
    for (u64 p = 0; p < from->n_parts;) {
        while (p < from->n_parts && cnt[p] == 0) p++;
        if (p >= from->n_parts) break;
        base.clear();
        u64 total = 0, q = p;
        while (q < from->n_parts && (total == 0 || total + cnt[q] <= round_cap)) {
            base.push_back(total);
            total += cnt[q];
            q++;
        }
        const u64 np = q - p, n_words = (total * k + 15) / 16;
        int rc;
        if ((rc = ensure(h, h->enum_out, np * 8 + total * 22 + 64))) return rc;
        if ((rc = ensure(h, h->packed_tmp, (n_words + 4) * 4))) return rc;
        if ((rc = ensure(h, h->starts_tmp, (total + 1) * 8))) return rc;
        char* b0 = (char*)h->enum_out.p;
        u64* d_base = (u64*)b0;
        u64* d_lo = (u64*)(b0 + np * 8);
        u64* d_hi = d_lo + total;
        uint8_t* d_idx = (uint8_t*)(d_hi + total) + total * 4;
        uint8_t* d_cnt = d_idx + total;
        HIPCHK(h, hipMemcpyAsync(d_base, base.data(), np * 8, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_enumerate, dim3((u32)std::min<u64>(np, 1u << 22)), dim3(64), 0, h->stream, from->P, from->ix, (u32)p, (u32)np, d_base, d_lo, d_hi, d_idx, d_cnt, (u32*)nullptr);
        if ((rc = launch_check(h, "k_enumerate"))) return rc;
        HIPCHK(h, hipMemsetAsync((char*)h->packed_tmp.p + n_words * 4, 0, 16, h->stream));
        hipLaunchKernelGGL(k_kmers_to_reads, dim3(nblocks(std::max<u64>(n_words, total + 1), 256)), dim3(256), 0, h->stream, d_lo, d_hi, total, k, (u32*)h->packed_tmp.p,
                           n_words, (u64*)h->starts_tmp.p);
        if ((rc = launch_check(h, "k_kmers_to_reads"))) return rc;
        // the scan as the query path runs it: its records carry the index of the read they came from (one record per read here)
        u64 n_rec = 0;
        bool hist_ok = true;
        if ((rc = scan_to_staging(h, (const u32*)h->packed_tmp.p, (const u64*)h->starts_tmp.p, total, true, true, &n_rec, &hist_ok))) return rc;
        if (n_rec != total) return fail(h, BRISK_HIP_EHIP, "reallocate: " + std::to_string(total) + " k-mers gave " + std::to_string(n_rec) + " records");
        hipLaunchKernelGGL(k_set_multiplicity, dim3(nblocks(n_rec, 256)), dim3(256), 0, h->stream, (u64*)h->staging.p, n_rec, h->P.stride, (const u32*)h->tags_a.p, d_cnt);
        if ((rc = launch_check(h, "k_set_multiplicity"))) return rc;
        if ((rc = insert_records_impl(h, (const u64*)h->staging.p, n_rec, hist_ok))) return rc;
        p = q;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_device_flags(h);
}

BRISK_API int brisk_hip_memory_info(brisk_hip_index* h, uint64_t out[4]) {
    if (!h || !out) return BRISK_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    out[0] = h->use_vmm ? h->vm_keys.mapped + h->vm_counts.mapped + h->vm_ids.mapped : h->arena_cap * (8ull * h->ix.key_words + 1 + (h->entry_ids ? 4 : 0));
    out[1] = h->use_vmm ? h->vm_keys.reserved + h->vm_counts.reserved + h->vm_ids.reserved : 0;
    out[2] = pool_bytes();
    out[3] = g_retired_va.load();
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_insert_slack(brisk_hip_index* h, uint64_t* entries) {
    if (!h || !entries) return BRISK_HIP_EINVAL;
    *entries = (uint64_t)h->insert_waves * ARENA_CHUNK;  // what insert_records_once adds to a batch's need (ensure_arena)
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_checksum(brisk_hip_index* h, uint64_t out[3]) {
    if (!h || !out) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    HIPCHK(h, hipMemsetAsync(h->d_small, 0, 24, h->stream));
    hipLaunchKernelGGL(k_checksum, dim3(2048), dim3(256), 0, h->stream, h->P, h->ix, (u32)h->n_parts, h->d_small);
    int rc;
    if ((rc = launch_check(h, "k_checksum"))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, 24, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out[0] = h->h_small[0];
    out[1] = h->h_small[1];
    out[2] = h->h_small[2];
    return BRISK_HIP_OK;
}

// ---- cut at the super-k-mer boundary ---------------------------------------
BRISK_API int brisk_hip_scan_bound(brisk_hip_index* h, const uint64_t* d_starts, uint64_t n_reads, uint64_t* bound) {
    if (!h || !bound || (n_reads && !d_starts)) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    *bound = 0;
    if (!n_reads) return BRISK_HIP_OK;
    u64 b = 0;
    int rc = count_kmers(h, d_starts, n_reads, &b);
    *bound = b;
    return rc;
}

BRISK_API int brisk_hip_scan_packed(brisk_hip_index* h, const uint32_t* d_packed, const uint64_t* d_starts, uint64_t n_reads, uint64_t* d_records,
                                    uint64_t cap_records, uint64_t* n_records) {
    if (!h || !n_records || (n_reads && (!d_packed || !d_starts)) || (cap_records && !d_records)) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    *n_records = 0;
    if (!n_reads) {  // an empty piece of a sharded job still exports a (zero) histogram: the exchange is collective
        h->scan_hist_valid = false;
        HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
        h->scan_hist_valid = true;
        return BRISK_HIP_OK;
    }
    u64 n = 0;
    u64 bound = 0;
    int rc = count_kmers(h, d_starts, n_reads, &bound);
    if (rc) return rc;
    // the per-partition histogram of what was scanned is kept: the owners of a sharded job need it (export_hist), also
    // when the job happens to have one owner
    const bool want_hist = true;
    bool hist_ok = false;
    h->scan_hist_valid = false;
    rc = scan_impl(h, d_packed, d_starts, n_reads, d_records, cap_records, want_hist, false, nullptr, &n, nullptr, bound, &hist_ok);
    *n_records = n;
    if (rc == BRISK_HIP_OK && want_hist) {
        if (!hist_ok && n) {  // a long sequence was re-scanned in places: rebuild from the final records
            HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
            hipLaunchKernelGGL(k_part_hist, dim3(nblocks(n, 256)), dim3(256), 0, h->stream, h->P, d_records, n, h->d_hist);
            if (int lrc = launch_check(h, "k_part_hist")) return lrc;
        }
        h->scan_hist_valid = true;
    }
    return rc;
}

BRISK_API int brisk_hip_export_hist(brisk_hip_index* h, uint64_t* d_hist_out, uint64_t* partitions_per_owner) {
    if (!h || !d_hist_out || !partitions_per_owner) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    if (!h->scan_hist_valid) return fail(h, BRISK_HIP_EINVAL, "export_hist: no histogram (brisk_hip_scan_packed on a sharded index must come right before)");
    HIPCHK(h, hipMemcpyAsync(d_hist_out, h->d_hist, h->n_parts * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (u32 o = 0; o < h->P.n_owners; o++) partitions_per_owner[o] = owner_first_partition(h, o + 1) - owner_first_partition(h, o);
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_export_hist_add(brisk_hip_index* h, uint64_t* d_hist_acc, uint64_t* partitions_per_owner) {
    if (!h || !d_hist_acc || !partitions_per_owner) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    if (!h->scan_hist_valid) return fail(h, BRISK_HIP_EINVAL, "export_hist_add: no histogram (brisk_hip_scan_packed on a sharded index must come right before)");
    hipLaunchKernelGGL(k_add_u64, dim3(std::min<u32>(nblocks(h->n_parts, 1024), 8192)), dim3(256), 0, h->stream, (const unsigned long long*)h->d_hist, h->n_parts,
                       (unsigned long long*)d_hist_acc);
    if (int lrc = launch_check(h, "k_add_u64")) return lrc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (u32 o = 0; o < h->P.n_owners; o++) partitions_per_owner[o] = owner_first_partition(h, o + 1) - owner_first_partition(h, o);
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_set_owner_cuts(brisk_hip_index* h, const uint64_t* first_partition) {
    if (!h || !first_partition) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    const u32 no = h->P.n_owners;
    if (no < 2) return fail(h, BRISK_HIP_EINVAL, "set_owner_cuts: not a sharded index");
    if (first_partition[0] != 0 || first_partition[no] != h->n_parts) return fail(h, BRISK_HIP_EINVAL, "set_owner_cuts: the ranges must cover partitions [0, 2^part_bits)");
    for (u32 o = 0; o < no; o++)
        if (first_partition[o] > first_partition[o + 1]) return fail(h, BRISK_HIP_EINVAL, "set_owner_cuts: first partitions must ascend");
    if (h->arena_used_host || h->nb_skmers) return fail(h, BRISK_HIP_EINVAL, "set_owner_cuts: the index holds entries of the old ranges (brisk_hip_clear first)");
    std::vector<u32> c32(no + 1);
    for (u32 o = 0; o <= no; o++) c32[o] = (u32)first_partition[o];
    if (!h->d_owner_cut) HIPCHK(h, hipMalloc((void**)&h->d_owner_cut, (ROUTE_MAX_OWNERS + 1) * 4));
    HIPCHK(h, hipMemcpyAsync(h->d_owner_cut, c32.data(), (no + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->owner_cut.assign(first_partition, first_partition + no + 1);
    h->scan_hist_valid = false;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_insert_records_hist(brisk_hip_index* h, const uint64_t* d_records, uint64_t n_records, const uint64_t* d_hist_slices, uint32_t n_slices) {
    if (!h || (n_records && (!d_records || !d_hist_slices || !n_slices))) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    if (h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "bulk count on an entry-id index");
    h->scan_hist_valid = false;
    if (!n_records) return BRISK_HIP_OK;
    const u64 p_lo = owner_first_partition(h, h->P.owner_rank), len = owner_first_partition(h, h->P.owner_rank + 1) - p_lo;
    HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_small + 4, 0, 8, h->stream));
    hipLaunchKernelGGL(k_sum_slices, dim3(nblocks(len, 256)), dim3(256), 0, h->stream, (const unsigned long long*)d_hist_slices, n_slices, len, h->d_hist + p_lo,
                       h->d_small + 4);
    if (int lrc = launch_check(h, "k_sum_slices")) return lrc;
    HIPCHK(h, hipMemcpyAsync(h->h_small + 4, h->d_small + 4, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->h_small[4] != n_records) return fail(h, BRISK_HIP_EINVAL, "insert_records_hist: the histogram slices count " + std::to_string(h->h_small[4]) +
                                                                        " records, " + std::to_string(n_records) + " were handed over");
    return insert_records_impl(h, d_records, n_records, true);
}

static int route_impl(brisk_hip_index* h, const uint64_t* d_records, const uint32_t* d_tags, uint64_t n_records, uint64_t* d_out, uint32_t* d_tags_out,
                      uint64_t* counts);
BRISK_API int brisk_hip_route_records(brisk_hip_index* h, const uint64_t* d_records, uint64_t n_records, uint64_t* d_out, uint64_t* counts) {
    return route_impl(h, d_records, nullptr, n_records, d_out, nullptr, counts);
}
BRISK_API int brisk_hip_route_tagged(brisk_hip_index* h, const uint64_t* d_records, const uint32_t* d_tags, uint64_t n_records, uint64_t* d_out,
                                     uint32_t* d_tags_out, uint64_t* counts) {
    if (n_records && (!d_tags || !d_tags_out)) return BRISK_HIP_EINVAL;
    return route_impl(h, d_records, d_tags, n_records, d_out, d_tags_out, counts);
}
static int route_impl(brisk_hip_index* h, const uint64_t* d_records, const uint32_t* d_tags, uint64_t n_records, uint64_t* d_out, uint32_t* d_tags_out,
                      uint64_t* counts) {
    if (!h || !counts || (n_records && (!d_records || !d_out))) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    const u32 no = h->P.n_owners;
    for (u32 i = 0; i < no; i++) counts[i] = 0;
    if (!n_records) return BRISK_HIP_OK;
    if (n_records >= (1ull << 32)) return fail(h, BRISK_HIP_EINVAL, "more than 2^32-1 records in one batch");
    if (no > h->n_parts) return fail(h, BRISK_HIP_EINVAL, "more owners than partitions");
    int rc;
    const u32 grid = std::min<u32>(ROUTE_BLOCKS, nblocks(n_records, 256));
    const u64 chunk = ((n_records + grid - 1) / grid + 255) / 256 * 256;  // records per block, whole 256-record tiles
    // owner histogram and offsets live in route_buf: d_hist may hold the scan's partition histogram (export_hist)
    if ((rc = ensure(h, h->route_buf, (size_t)(no + 1) * 12 + (size_t)grid * no * 4))) return rc;
    unsigned long long* d_ohist = (unsigned long long*)h->route_buf.p;
    u32* d_ooff = (u32*)(d_ohist + no + 1);
    u32* d_block = d_ooff + no + 1;
    HIPCHK(h, hipMemsetAsync(d_ohist, 0, ((u64)no + 1) * 8, h->stream));
    {
        ProfScope ps(h, S_HIST);
        hipLaunchKernelGGL(k_owner_hist, dim3(grid), dim3(256), 0, h->stream, h->P, d_records, n_records, chunk, d_block, d_ohist, (const u32*)h->d_owner_cut);
        if (int lrc = launch_check(h, "k_owner_hist")) return lrc;
        hipLaunchKernelGGL(k_owner_offsets, dim3(1), dim3(ROUTE_MAX_OWNERS), 0, h->stream, no, grid, d_ohist, d_block, d_ooff);
        if (int lrc = launch_check(h, "k_owner_offsets")) return lrc;
    }
    {
        ProfScope ps(h, S_SCATTER);
        hipLaunchKernelGGL(k_owner_scatter, dim3(grid), dim3(256), 0, h->stream, h->P, d_records, n_records, chunk, d_block, d_out, d_tags, d_tags_out, (const u32*)h->d_owner_cut);
        if ((rc = launch_check(h, "k_owner_scatter"))) return rc;
    }
    std::vector<u32> off(no + 1);
    HIPCHK(h, hipMemcpyAsync(off.data(), d_ooff, ((u64)no + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (u32 i = 0; i < no; i++) counts[i] = off[i + 1] - off[i];
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_insert_records(brisk_hip_index* h, const uint64_t* d_records, uint64_t n_records) {
    if (!h || (n_records && !d_records)) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    if (h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "bulk count on an entry-id index");
    return insert_records_impl(h, d_records, n_records, false);
}

BRISK_API int brisk_hip_scan_query(brisk_hip_index* h, const uint32_t* d_packed, const uint64_t* d_starts, uint64_t n_reads, uint64_t* d_records,
                                   uint32_t* d_tags, uint64_t cap_records, uint64_t* n_records) {
    if (!h || !n_records || (n_reads && (!d_packed || !d_starts)) || (cap_records && (!d_records || !d_tags))) return BRISK_HIP_EINVAL;
    if (n_reads >= (1ull << 32)) return fail(h, BRISK_HIP_EINVAL, "more than 2^32-1 reads in one query batch");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    *n_records = 0;
    if (!n_reads) return BRISK_HIP_OK;
    u64 n = 0, bound = 0;
    int rc = count_kmers(h, d_starts, n_reads, &bound);
    if (rc) return rc;
    rc = scan_impl(h, d_packed, d_starts, n_reads, d_records, cap_records, false, true, d_tags, &n, nullptr, bound);
    *n_records = n;
    return rc;
}

BRISK_API int brisk_hip_query_records(brisk_hip_index* h, const uint64_t* d_records, uint64_t n_records, uint64_t* d_sums) {
    if (!h || (n_records && (!d_records || !d_sums))) return BRISK_HIP_EINVAL;
    if (n_records >= (1ull << 32)) return fail(h, BRISK_HIP_EINVAL, "more than 2^32-1 records in one batch");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    if (!n_records) return BRISK_HIP_OK;
    int rc;
    if ((rc = ensure(h, h->tags_a, n_records * 4))) return rc;
    HIPCHK(h, hipMemsetAsync(d_sums, 0, n_records * 8, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_hist, 0, (h->n_parts + 1) * 8, h->stream));
    hipLaunchKernelGGL(k_iota, dim3(nblocks(n_records, 256)), dim3(256), 0, h->stream, (u32*)h->tags_a.p, n_records);
    hipLaunchKernelGGL(k_part_hist, dim3(nblocks(n_records, 256)), dim3(256), 0, h->stream, h->P, d_records, n_records, h->d_hist);
    if ((rc = launch_check(h, "k_part_hist"))) return rc;
    if ((rc = query_records_impl(h, d_records, (const u32*)h->tags_a.p, n_records, (unsigned long long*)d_sums))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_device_flags(h);
}

// ---- the per-call API under the C++ facade ----------------------------------------
BRISK_API int brisk_hip_scan_sequence(brisk_hip_index* h, const char* bases, uint64_t len, uint64_t cap_kmers, uint64_t* skm_ret, uint32_t* skm_n,
                                      uint64_t* km_lo, uint64_t* km_hi, uint8_t* km_idx, uint64_t* n_skm) {
    if (!h || !n_skm) return BRISK_HIP_EINVAL;
    *n_skm = 0;
    if (len < h->P.k) return BRISK_HIP_OK;
    const u64 nk = len - h->P.k + 1;
    if (!bases || !skm_ret || !skm_n || !km_lo || !km_hi || !km_idx || cap_kmers < nk) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    const uint64_t offs[2] = {0, len};
    // SuperKmerEnumerator::next yields whole vectors: no minimizer_idx classes in the routing ids of this scan (they are not used:
    // the records only carry the vectors to k_expand_records)
    BriskParams P = h->P;
    P.ext_bits -= P.cls_bits;
    P.cls_bits = 0;
    const u32 row = P.w + 1;
    return for_each_host_batch(h, bases, offs, 1, [&](u64, u64) -> int {
        int rc;
        // worst case one vector per k-mer
        if ((rc = ensure(h, h->staging, nk * P.stride * 8))) return rc;
        if ((rc = ensure(h, h->tags_a, nk * 4))) return rc;
        if ((rc = ensure(h, h->seq_buf, nk * 8 + nk * (size_t)row * 17))) return rc;
        u64* d_ret = (u64*)h->seq_buf.p;
        u64* d_lo = d_ret + nk;
        u64* d_hi = d_lo + nk * row;
        uint8_t* d_idx = (uint8_t*)(d_hi + nk * row);
        u64 n_rec = 0;
        if ((rc = scan_impl(h, (const u32*)h->packed_tmp.p, (const u64*)h->starts_tmp.p, 1, (u64*)h->staging.p, nk, false, false, (u32*)h->tags_a.p,
                            &n_rec, d_ret)))
            return rc;
        if (!n_rec) return BRISK_HIP_OK;
        hipLaunchKernelGGL(k_expand_records, dim3((u32)n_rec), dim3(64), 0, h->stream, P, (const u64*)h->staging.p, (u32)n_rec, row, d_lo, d_hi, d_idx);
        if ((rc = launch_check(h, "k_expand_records"))) return rc;
        std::vector<u64> rec(n_rec * P.stride), ret(n_rec), lo(n_rec * row), hi(n_rec * row);
        std::vector<u32> pos(n_rec);
        std::vector<uint8_t> idx(n_rec * row);
        HIPCHK(h, hipMemcpyAsync(rec.data(), h->staging.p, rec.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(ret.data(), d_ret, n_rec * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(pos.data(), h->tags_a.p, n_rec * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(lo.data(), d_lo, lo.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(hi.data(), d_hi, hi.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(idx.data(), d_idx, idx.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // vectors come back in no particular order: next() hands them out by position in the sequence
        std::vector<u32> order(n_rec);
        for (u32 i = 0; i < n_rec; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return pos[a] < pos[b]; });
        u64 at = 0;
        for (u64 o = 0; o < n_rec; o++) {
            const u32 r = order[o];
            const u32 n = (u32)(rec[(u64)r * P.stride + P.nw] >> 32) & 0xff;
            skm_ret[o] = ret[r];
            skm_n[o] = n;
            for (u32 j = 0; j < n; j++, at++) {
                km_lo[at] = lo[(u64)r * row + j];
                km_hi[at] = hi[(u64)r * row + j];
                km_idx[at] = idx[(u64)r * row + j];
            }
        }
        *n_skm = n_rec;
        return BRISK_HIP_OK;
    });
}

static int upload_queries(brisk_hip_index* h, const uint64_t* lo, const uint64_t* hi, const uint8_t* idx, u64 n, u64** d_lo, u64** d_hi,
                          uint8_t** d_idx, u32** d_ids, uint8_t** d_new) {
    int rc;
    if ((rc = ensure(h, h->lookup_buf, n * 22 + 64))) return rc;
    char* base = (char*)h->lookup_buf.p;
    *d_lo = (u64*)base;
    *d_hi = (u64*)(base + n * 8);
    *d_ids = (u32*)(base + n * 16);
    *d_idx = (uint8_t*)(base + n * 20);
    *d_new = *d_idx + n;
    HIPCHK(h, hipMemcpyAsync(*d_lo, lo, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(*d_hi, hi, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(*d_idx, idx, n, hipMemcpyHostToDevice, h->stream));
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_upsert_kmers(brisk_hip_index* h, const uint64_t* kmer_lo, const uint64_t* kmer_hi, const uint8_t* minimizer_idx, uint64_t n,
                                     uint32_t* ids, uint8_t* newly) {
    if (!h || (n && (!kmer_lo || !kmer_hi || !minimizer_idx || !ids || !newly)) || n > 255) return BRISK_HIP_EINVAL;
    if (!h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "not an entry-id index");
    if (!n) return BRISK_HIP_OK;
    for (u64 i = 0; i < n; i++)
        if (minimizer_idx[i] > h->P.w) return fail(h, BRISK_HIP_EINVAL, "minimizer_idx > k-m");
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    int rc;
    if (!h->arena_cap && (rc = ensure_arena(h, 1u << 16))) return rc;
    u64 *d_lo, *d_hi;
    uint8_t *d_idx, *d_new;
    u32* d_ids;
    u32 done = 0;
    {
        // One vector (a super-k-mer: <= k - m + 1 k-mers) per call is what Brisk::insert_superkmer brings (brisk/Brisk.hpp:123-147,
        // apps/counter.cpp:242-270): the call's cost is its host round trips, so the k-mers go in with ONE copy from pinned memory
        // and ids, flags, progress and the arena cursor come back with one (six pageable copies and two synchronisations before).
        if ((rc = ensure(h, h->lookup_buf, n * 22 + 64))) return rc;
        char* base = (char*)h->lookup_buf.p;
        const size_t tail = (n * 22 + 7) / 8 * 8;  // [n_done u32, pad | cursor u64] behind the arrays
        d_lo = (u64*)base;
        d_hi = (u64*)(base + n * 8);
        d_ids = (u32*)(base + n * 16);
        d_idx = (uint8_t*)(base + n * 20);
        d_new = d_idx + n;
        memcpy(h->h_pin, kmer_lo, n * 8);
        memcpy(h->h_pin + n * 8, kmer_hi, n * 8);
        memcpy(h->h_pin + n * 20, minimizer_idx, n);
        HIPCHK(h, hipMemcpyAsync(base, h->h_pin, n * 21, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_upsert, dim3(1), dim3(64), 0, h->stream, h->P, h->ix, d_lo, d_hi, d_idx, (u32)n, d_ids, d_new, h->d_id_counter, (u32*)(base + tail),
                           (unsigned long long*)(base + tail + 8));
        if ((rc = launch_check(h, "k_upsert"))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->h_pin + n * 16, base + n * 16, tail + 16 - n * 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const u32 step = *(const u32*)(h->h_pin + tail);
        h->arena_used_host = *(const unsigned long long*)(h->h_pin + tail + 8);
        if (step == n) {
            memcpy(ids, h->h_pin + n * 16, n * 4);
            memcpy(newly, h->h_pin + n * 21, n);
            h->nb_skmers += 1;
            h->dir_snapshot_valid = false;
            return BRISK_HIP_OK;
        }
        // the arena filled up part-way through the vector: the general path below grows it and goes on behind the k-mers that are in
        // (their ids and flags are where the general path's copy at the end finds them: same buffer, same layout)
        done = step;
        if ((rc = ensure_arena(h, h->arena_cap))) return rc;
    }
    if ((rc = upload_queries(h, kmer_lo, kmer_hi, minimizer_idx, n, &d_lo, &d_hi, &d_idx, &d_ids, &d_new))) return rc;
    while (done < n) {
        hipLaunchKernelGGL(k_upsert, dim3(1), dim3(64), 0, h->stream, h->P, h->ix, d_lo + done, d_hi + done, d_idx + done, (u32)(n - done),
                           d_ids + done, d_new + done, h->d_id_counter, (u32*)(h->d_small + 7));
        if ((rc = launch_check(h, "k_upsert"))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->h_small + 7, h->d_small + 7, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_small + 5, h->ix.cursor, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const u32 step = (u32)h->h_small[7];
        done += step;
        h->arena_used_host = h->h_small[5];
        if (done < n) {  // the arena is full: double it and go on with the rest of the vector
            if ((rc = ensure_arena(h, h->arena_cap))) return rc;
        }
    }
    HIPCHK(h, hipMemcpyAsync(ids, d_ids, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(newly, d_new, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->nb_skmers += 1;
    h->dir_snapshot_valid = false;
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_find_kmers(brisk_hip_index* h, const uint64_t* kmer_lo, const uint64_t* kmer_hi, const uint8_t* minimizer_idx, uint64_t n,
                                   uint32_t* ids) {
    if (!h || (n && (!kmer_lo || !kmer_hi || !minimizer_idx || !ids))) return BRISK_HIP_EINVAL;
    if (!h->entry_ids) return fail(h, BRISK_HIP_EINVAL, "not an entry-id index");
    if (!n) return BRISK_HIP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    if (int frc = enter(h)) return frc;
    int rc;
    u64 *d_lo, *d_hi;
    uint8_t *d_idx, *d_new;
    u32* d_ids;
    if ((rc = upload_queries(h, kmer_lo, kmer_hi, minimizer_idx, n, &d_lo, &d_hi, &d_idx, &d_ids, &d_new))) return rc;
    if (!h->arena_cap) {
        for (u64 i = 0; i < n; i++) ids[i] = 0xffffffffu;
        return BRISK_HIP_OK;
    }
    hipLaunchKernelGGL(k_lookup, dim3(nblocks(n * 64, 256)), dim3(256), 0, h->stream, h->P, h->ix, d_lo, d_hi, d_idx, n, d_new, d_new, d_ids);
    if ((rc = launch_check(h, "k_lookup(ids)"))) return rc;
    HIPCHK(h, hipMemcpyAsync(ids, d_ids, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BRISK_HIP_OK;
}

// ---- helpers ------------------------------------------------------------------
BRISK_API int brisk_hip_pack_ascii(brisk_hip_index* h, const char* d_bases, uint64_t n_bases, uint32_t* d_packed) {
    if (!h || (n_bases && (!d_bases || !d_packed))) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    const u64 n_words = (n_bases + 15) / 16;
    if (!n_words) return BRISK_HIP_OK;
    ProfScope ps(h, S_PACK);
    hipLaunchKernelGGL(k_pack_ascii, dim3(nblocks(n_words, 256)), dim3(256), 0, h->stream, (const uint8_t*)d_bases, n_bases, d_packed, n_words);
    return launch_check(h, "k_pack_ascii");
}

BRISK_API int brisk_hip_synth_reads(brisk_hip_index* h, uint64_t genome_len, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                    uint64_t seed_g, uint64_t seed_r, uint32_t* d_packed, uint64_t* d_starts) {
    if (!h || !d_packed || !d_starts || read_len == 0 || genome_len < read_len) return BRISK_HIP_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    const u64 n_words = (n_reads * (u64)read_len + 15) / 16;
    const u64 n_threads = std::max<u64>(n_words, n_reads + 1);
    ProfScope ps(h, S_SYNTH);
    hipLaunchKernelGGL(k_synth, dim3(nblocks(n_threads, 256)), dim3(256), 0, h->stream, genome_len, first_read, n_reads, read_len, seed_g, seed_r,
                       d_packed, n_words, d_starts);
    return launch_check(h, "k_synth");
}

BRISK_API int brisk_hip_debug_order_keys(brisk_hip_index* h, const uint64_t* mmers, uint64_t n, int exact, uint64_t* keys) {
    if (!h || (n && (!mmers || !keys))) return BRISK_HIP_EINVAL;
    if (!n) return BRISK_HIP_OK;
    HIPCHK(h, hipSetDevice(h->device));
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    int rc;
    if ((rc = ensure(h, h->lookup_buf, n * 16))) return rc;
    u64* d_x = (u64*)h->lookup_buf.p;
    u64* d_k = d_x + n;
    HIPCHK(h, hipMemcpyAsync(d_x, mmers, n * 8, hipMemcpyHostToDevice, h->stream));
    const size_t lds = (size_t)h->scfg.n_tab * 8;
    hipLaunchKernelGGL(k_debug_keys, dim3(nblocks(n, 256)), dim3(256), lds, h->stream, h->P, h->scfg, h->d_tabs, d_x, n, exact, d_k);
    if ((rc = launch_check(h, "k_debug_keys"))) return rc;
    HIPCHK(h, hipMemcpyAsync(keys, d_k, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BRISK_HIP_OK;
}

BRISK_API int brisk_hip_debug_host_pack(const char* bases, uint64_t n_bases, uint32_t* packed, int scalar) {  // the upload threads' packer, for tests (no device)
    if (n_bases && (!bases || !packed)) return BRISK_HIP_EINVAL;
    if (scalar) host_pack_scalar(bases, n_bases, packed);
    else host_pack(bases, n_bases, packed);
    return BRISK_HIP_OK;
}
#ifdef BRISK_PHASE_PROF
BRISK_API int brisk_hip_debug_scan_counts(uint64_t out[8], int reset) {  // debug builds only (tools/phase_profile.py)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_cnt), 8 * 8) != hipSuccess) return BRISK_HIP_EHIP;
    if (reset) {
        uint64_t z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_cnt), z, 8 * 8) != hipSuccess) return BRISK_HIP_EHIP;
    }
    return BRISK_HIP_OK;
}
BRISK_API int brisk_hip_debug_phases(uint64_t out[32], int reset) {  // debug builds only (tools/phase_profile.py)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), 16 * 8) != hipSuccess) return BRISK_HIP_EHIP;
    if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_cnt), 16 * 8) != hipSuccess) return BRISK_HIP_EHIP;
    if (reset) {
        uint64_t z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, 16 * 8) != hipSuccess) return BRISK_HIP_EHIP;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), z, 16 * 8) != hipSuccess) return BRISK_HIP_EHIP;
    }
    return BRISK_HIP_OK;
}
#endif

// ---- measurement ---------------------------------------------------------------
BRISK_API int brisk_hip_profile_enable(brisk_hip_index* h, int on) {
    if (!h) return BRISK_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    int rc = drain_profile(h);
    h->profiling = on != 0;
    return rc;
}
BRISK_API int brisk_hip_profile_reset(brisk_hip_index* h) {
    if (!h) return BRISK_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    int rc = drain_profile(h);
    for (int i = 0; i < S_NSLOTS; i++) { h->prof_ms[i] = 0; h->prof_launches[i] = 0; }
    return rc;
}
BRISK_API int brisk_hip_profile_read(brisk_hip_index* h, uint32_t* n_slots, const char** names, uint64_t* launches, double* ms) {
    if (!h || !n_slots) return BRISK_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> call_lock(h->call_mu);
    int rc = drain_profile(h);
    *n_slots = S_NSLOTS;
    for (int i = 0; i < S_NSLOTS; i++) {
        if (names) names[i] = kSlotNames[i];
        if (launches) launches[i] = h->prof_launches[i];
        if (ms) ms[i] = h->prof_ms[i];
    }
    return rc;
}

}  // extern "C"
