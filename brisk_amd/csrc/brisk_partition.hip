// brisk_partition.hip -- records to partitions and owners: prefix sums, touched-partition list and descriptors,
// owner routing, partition histogram and scatter.  Included by brisk_kernels.hip (one translation unit).
// ===========================================================================
// exclusive prefix sum over the low 32 bits of the 64-bit histogram
#define SCAN_ITEMS 16
// `sub`: the prefix runs over max(count, sub) - sub -- the records a partition has beyond its bin (binned layout); 0: all of them
__device__ __forceinline__ u32 beyond(unsigned long long h, u32 sub) {
    const u32 c = (u32)h;
    return c > sub ? c - sub : 0u;
}
__global__ void __launch_bounds__(256) k_psum_block(const unsigned long long* __restrict__ hist, u64 n, u32* __restrict__ block_sums, u32 sub) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * 256 * SCAN_ITEMS;
    u32 acc = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + (u64)i * 256 + threadIdx.x;
        if (j < n) acc += beyond(hist[j], sub);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// single block: in-place exclusive scan of block_sums[nb]
__global__ void __launch_bounds__(1024) k_psum_top(u32* __restrict__ block_sums, u32 nb) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nb; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nb ? block_sums[i] : 0;
        u32 x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
        __syncthreads();
        u32 woff = 0;
        for (u32 j = 0; j < (threadIdx.x >> 6); j++) woff += s_wave[j];
        const u32 carry = s_carry;
        if (i < nb) block_sums[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + x;
        __syncthreads();
    }
}
// per block: write exclusive offsets; also seeds the scatter cursors
__global__ void __launch_bounds__(256) k_psum_apply(const unsigned long long* __restrict__ hist, u64 n, const u32* __restrict__ block_sums,
                                                    u32* __restrict__ off, u32* __restrict__ cursor, u32 sub) {
    __shared__ u32 s_wave[4];
    const u64 base = (u64)blockIdx.x * 256 * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + i;
        v[i] = j < n ? beyond(hist[j], sub) : 0;
        tsum += v[i];
    }
    u32 x = tsum;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    u32 woff = 0;
    for (u32 j = 0; j < (threadIdx.x >> 6); j++) woff += s_wave[j];
    u32 run = block_sums[blockIdx.x] + woff + x - tsum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + i;
        if (j < n) {
            off[j] = run;
            cursor[j] = run;
        }
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) off[n] = run;  // total
}

// acc[i] += v[i] (both counters of a histogram word at once: neither carries into the other before 2^32 records)
__global__ void __launch_bounds__(256) k_add_u64(const unsigned long long* __restrict__ v, u64 n, unsigned long long* __restrict__ acc) {
    for (u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (u64)gridDim.x * 512) {
        if (i + 1 < n) {
            const ulonglong2 a = *(const ulonglong2*)(v + i);
            ulonglong2 b = *(ulonglong2*)(acc + i);
            b.x += a.x;
            b.y += a.y;
            *(ulonglong2*)(acc + i) = b;
        } else {
            acc[i] += v[i];
        }
    }
}

// list of partitions with records, ascending inside a block.  One list-cursor atomic per block of 8192
// partitions: every same-address atomic costs ~15 ns device-wide, whoever issues it.
#define TOUCHED_ITEMS 8
// (hist, n: the index's own partitions, the first of them partition `first`)
__global__ void __launch_bounds__(1024) k_touched(const unsigned long long* __restrict__ hist, u64 n, u32 first, u32* __restrict__ list, u32* __restrict__ n_list) {
    __shared__ u32 s_wsum[16];
    __shared__ u32 s_base;
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const u64 p0 = ((u64)blockIdx.x * 1024 + threadIdx.x) * TOUCHED_ITEMS;
    u32 mask = 0;
#pragma unroll
    for (u32 j = 0; j < TOUCHED_ITEMS; j++)
        if (p0 + j < n && (u32)hist[p0 + j] != 0) mask |= 1u << j;
    const u32 cnt = (u32)__popc(mask);
    u32 incl = cnt;  // inclusive scan over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += y;
    }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0;
        for (u32 i = 0; i < 16; i++) {
            const u32 c = s_wsum[i];
            s_wsum[i] = tot;
            tot += c;
        }
        s_base = tot ? atomicAdd(n_list, tot) : 0u;
    }
    __syncthreads();
    u32 at = s_base + s_wsum[wid] + incl - cnt;
#pragma unroll
    for (u32 j = 0; j < TOUCHED_ITEMS; j++)
        if (mask >> j & 1) list[at++] = first + (u32)(p0 + j);
}

// Per touched partition: a 32-byte work descriptor for k_insert (so that its
// persistent workgroups fetch ONE predictable line per partition instead of
// chasing touched[] -> part_off[] -> dir_*[]), and the arena space the batch may
// need if every instance were new.
struct DirEnt {                  // one 16-byte directory line per partition
    unsigned long long off;      // first entry of the partition's slice
    u32 cnt, cap;                // entries in use / slice capacity
};
struct PartDesc {
    u32 part, r_begin, n_rec, n_inst, n_exist, cap;
    unsigned long long off;
};
__device__ __forceinline__ u32 grow_cap(u32 n) { return n + (n >> 2) + 8; }
#define PART_HUGE 0x80000000u   // PartDesc::n_exist of a query batch: k_query_huge takes this partition
// bin_cap > 0 (binned layout): n_rec comes from the histogram, r_begin is the partition's offset among the overflow
// records (part_off then holds the prefix over the records beyond the bins; null when no partition overflowed)
// huge (null: none): huge[0] counts, huge[1..] lists, the descriptors of partitions with more than huge_at k-mer instances (k_insert_huge) or,
// huge_at == 0, with more than huge_exist_at entries (k_query_huge)
__global__ void __launch_bounds__(256) k_need(const unsigned long long* __restrict__ hist, const u32* __restrict__ part_off,
                                              const u32* __restrict__ list, u32 n_list, const DirEnt* __restrict__ dir,
                                              PartDesc* __restrict__ desc, unsigned long long* out, u32 bin_cap, u32 huge_at = 0, u32* __restrict__ huge = nullptr,
                                              u32 huge_cap = 0, u32 huge_exist_at = 0, unsigned long long huge_work = 0) {
    __shared__ unsigned long long s_sum[4];
    unsigned long long need = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {  // grid-stride: few blocks, few atomics
        const u32 p = list[i];
        PartDesc d;
        d.part = p;
        if (bin_cap) {
            d.r_begin = part_off ? part_off[p] : 0u;
            d.n_rec = (u32)hist[p];
        } else {
            d.r_begin = part_off[p];
            d.n_rec = part_off[p + 1] - d.r_begin;
        }
        d.n_inst = (u32)(hist[p] >> 32);
        const DirEnt de = dir[p];
        d.n_exist = de.cnt;
        d.cap = de.cap;
        d.off = de.off;
        const u32 tot = d.n_exist + d.n_inst;
        if (huge) {
            if (huge_at) {  // insert: many instances (the host diverts none if the list overflows)
                if (d.n_inst > huge_at) {
                    const u32 at = atomicAdd(&huge[0], 1u);
                    if (at < huge_cap) huge[1 + at] = i;
                }
            } else if (d.n_exist > huge_exist_at && (unsigned long long)d.n_exist * d.n_inst > huge_work) {
                // query: entries x instances is what a wave would have to do; a listed partition carries PART_HUGE in n_exist
                const u32 at = atomicAdd(&huge[0], 1u);
                if (at < huge_cap) {
                    huge[1 + at] = i;
                    d.n_exist |= PART_HUGE;
                }
            }
        }
        desc[i] = d;
        if (tot > d.cap) need += grow_cap(tot);
    }
    for (int o = 32; o > 0; o >>= 1) need += __shfl_down(need, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = need;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// ===========================================================================
// k_scatter: bucket radix -- move each record to its partition's slice.
// owner mode (n_owners > 1 and by_owner): the bins are owners instead of partitions.
// Owner routing has few bins (n_owners <= ROUTE_MAX_OWNERS), so a global atomic per record -- or even per
// wave -- would serialise on a handful of addresses.  It is a two-pass radix step without them: every
// block owns a contiguous range of records, counts them per owner in LDS (k_owner_hist), a small kernel
// turns the [block][owner] counts into exclusive offsets (k_owner_offsets), and the second pass ranks
// records with LDS cursors seeded from those offsets (k_owner_scatter).
#define ROUTE_BLOCKS 2048u
#define ROUTE_MAX_OWNERS 256u
// Owner of a record: the owners hold contiguous partition ranges.  cuts == null: equal ranges, owner = partition * N >> part_bits.
// Else cuts[o] = first partition of owner o (cuts[0] = 0, n_owners + 1 entries, ascending): histogram-balanced ranges
// (brisk_hip_set_owner_cuts; SURVEY.md 8(e): a partition is a range of hash values and minimizers are the SMALLEST hashes of
// their windows, so equal ranges do not carry equal loads).
__device__ __forceinline__ u32 owner_of_partition(const BriskParams& P, u32 part, const u32* __restrict__ cuts) {
    if (!cuts) return (u32)(((u64)part * P.n_owners) >> P.part_bits);
    u32 lo = 0, hi = P.n_owners;  // the owner o with cuts[o] <= part < cuts[o + 1]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (cuts[mid] <= part) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ u32 owner_of_record(const BriskParams& P, u64 hdr, const u32* __restrict__ cuts) { return owner_of_partition(P, hdr_bucket(hdr) >> P.shift, cuts); }
__global__ void __launch_bounds__(256) k_owner_hist(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u64 chunk, u32* __restrict__ block_cnt,
                                                    unsigned long long* __restrict__ hist, const u32* __restrict__ cuts) {
    __shared__ u32 s_cnt[ROUTE_MAX_OWNERS];
    __shared__ u32 s_inst[ROUTE_MAX_OWNERS];
    for (u32 o = threadIdx.x; o < P.n_owners; o += 256) s_cnt[o] = s_inst[o] = 0;
    __syncthreads();
    const u64 begin = (u64)blockIdx.x * chunk, end = begin + chunk < n_rec ? begin + chunk : n_rec;
    const u32 lane = threadIdx.x & 63;
    for (u64 base = begin; base < end; base += 256) {
        const u64 i = base + threadIdx.x;
        const bool ok = i < end;
        u64 hdr = 0;
        if (ok) hdr = rec[i * P.stride + P.nw];
        const u32 owner = ok ? owner_of_record(P, hdr, cuts) : 0xffffffffu;
        unsigned long long todo = __ballot(ok);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const u32 o = (u32)__builtin_amdgcn_readlane((int)owner, lead);
            const unsigned long long same = __ballot(owner == o);
            u32 inst = owner == o ? hdr_n(hdr) : 0;  // k-mer instances of this owner in the wave
            for (int d = 32; d > 0; d >>= 1) inst += __shfl_xor(inst, d, 64);
            if ((int)lane == lead) {
                atomicAdd(&s_cnt[o], (u32)__popcll(same));
                atomicAdd(&s_inst[o], inst);
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    for (u32 o = threadIdx.x; o < P.n_owners; o += 256) {
        block_cnt[(u64)blockIdx.x * P.n_owners + o] = s_cnt[o];
        if (s_cnt[o]) atomicAdd(&hist[o], (unsigned long long)s_cnt[o] | ((unsigned long long)s_inst[o] << 32));
    }
}
// counts -> exclusive offsets, in place; off[o] = first slot of owner o, off[n_owners] = total
__global__ void __launch_bounds__(ROUTE_MAX_OWNERS) k_owner_offsets(u32 n_owners, u32 n_blocks, const unsigned long long* __restrict__ hist,
                                                                   u32* __restrict__ block_cnt, u32* __restrict__ off) {
    const u32 o = threadIdx.x;
    if (o > n_owners) return;
    u32 start = 0;
    for (u32 j = 0; j < o && j < n_owners; j++) start += (u32)hist[j];
    off[o] = start;
    if (o == n_owners) return;
    for (u32 b = 0; b < n_blocks; b++) {
        const u32 c = block_cnt[(u64)b * n_owners + o];
        block_cnt[(u64)b * n_owners + o] = start;
        start += c;
    }
}
__global__ void __launch_bounds__(256) k_owner_scatter(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u64 chunk,
                                                       const u32* __restrict__ block_off, u64* __restrict__ out,
                                                       const u32* __restrict__ tag_in, u32* __restrict__ tag_out, const u32* __restrict__ cuts) {
    __shared__ u32 s_cur[ROUTE_MAX_OWNERS];
    for (u32 o = threadIdx.x; o < P.n_owners; o += 256) s_cur[o] = block_off[(u64)blockIdx.x * P.n_owners + o];
    __syncthreads();
    const u64 begin = (u64)blockIdx.x * chunk, end = begin + chunk < n_rec ? begin + chunk : n_rec;
    const u32 lane = threadIdx.x & 63;
    for (u64 base = begin; base < end; base += 256) {
        const u64 i = base + threadIdx.x;
        const bool ok = i < end;
        const u64* src = rec + i * P.stride;
        u64 hdr = 0;
        if (ok) hdr = src[P.nw];
        const u32 owner = ok ? owner_of_record(P, hdr, cuts) : 0xffffffffu;
        unsigned long long todo = __ballot(ok);
        u32 slot = 0;
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const u32 o = (u32)__builtin_amdgcn_readlane((int)owner, lead);
            const unsigned long long same = __ballot(owner == o);
            u32 b0 = 0;
            if ((int)lane == lead) b0 = atomicAdd(&s_cur[o], (u32)__popcll(same));
            b0 = (u32)__builtin_amdgcn_readlane((int)b0, lead);
            if (owner == o) slot = b0 + (u32)__popcll(same & lanes_below(lane));
            todo &= ~same;
        }
        if (ok) {
            u64* dst = out + (u64)slot * P.stride;
            if (P.stride == 4) {
                const uint4* s4 = reinterpret_cast<const uint4*>(src);
                uint4* d4 = reinterpret_cast<uint4*>(dst);
                const uint4 a = s4[0], b = s4[1];
                d4[0] = a;
                d4[1] = b;
            } else {
                for (u32 j = 0; j < P.stride; j++) dst[j] = src[j];
            }
            if (tag_in) tag_out[slot] = tag_in[i];
        }
    }
}
// an owner's histogram = the sum of the slices the scanning ranks sent for its partition range
__global__ void __launch_bounds__(256) k_sum_slices(const unsigned long long* __restrict__ slices, u32 n_slices, u64 len, unsigned long long* __restrict__ hist_at_range,
                                                    unsigned long long* __restrict__ n_rec_total) {
    __shared__ unsigned long long s_sum[4];
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    if (i < len) {
        for (u32 sidx = 0; sidx < n_slices; sidx++) acc += slices[(u64)sidx * len + i];
        hist_at_range[i] = acc;
    }
    unsigned long long recs = acc & 0xffffffffull;
    for (int o = 32; o > 0; o >>= 1) recs += __shfl_down(recs, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = recs;
    __syncthreads();
    if (threadIdx.x == 0 && (s_sum[0] | s_sum[1] | s_sum[2] | s_sum[3])) atomicAdd(n_rec_total, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}
// the records in the scan's overflow regions -> *out
__global__ void __launch_bounds__(1024) k_sum_regions(const u32* __restrict__ cnt, u32 region_cap, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s[16];
    unsigned long long acc = 0;
    for (u32 i = threadIdx.x; i < OVF_REGIONS; i += 1024) acc += min(cnt[i], region_cap);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; i++) t += s[i];
        *out = t;
    }
}
__global__ void __launch_bounds__(256) k_rebase(u64* __restrict__ v, u64 n, u64 base) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] -= base;
}
__global__ void __launch_bounds__(256) k_iota(u32* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u32)i;
}
__global__ void __launch_bounds__(256) k_part_hist(BriskParams P, const u64* __restrict__ rec, u64 n_rec, unsigned long long* __restrict__ hist) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u64 hdr = rec[i * P.stride + P.nw];
    atomicAdd(&hist[hdr_bucket(hdr) >> P.shift], 1ull | ((unsigned long long)hdr_n(hdr) << 32));
}
// region_cnt != null: `rec` is the scan's overflow area -- OVF_REGIONS regions of region_cap slots, region g filled up to region_cnt[g]
// (ScanOut) -- and n_rec the records in it; the launch covers every slot.  Null: n_rec records back to back.
__global__ void __launch_bounds__(256) k_scatter(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u32* __restrict__ cursor,
                                                 u64* __restrict__ out, int by_owner, const u32* __restrict__ tag_in, u32* __restrict__ tag_out,
                                                 u32* __restrict__ err, const u32* __restrict__ region_cnt = nullptr, u32 region_cap = 0) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (region_cnt) {
        const u64 reg = i / region_cap;
        if (reg >= OVF_REGIONS || i - reg * region_cap >= min(region_cnt[reg], region_cap)) return;
    } else if (i >= n_rec) return;
    const u64* src = rec + i * P.stride;
    const u64 hdr = src[P.nw];
    u32 bin = hdr_bucket(hdr) >> P.shift;
    if (by_owner) bin = owner_of_partition(P, bin, nullptr);
    const u32 slot = atomicAdd(&cursor[bin], 1u);
    if (slot >= n_rec) {  // histogram and records disagree: never write out of range
        atomicOr(err, 1u);
        return;
    }
    u64* dst = out + (u64)slot * P.stride;
    if (P.stride == 4) {  // 32-byte records (k63/m21/b14): two 16-byte moves
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        const uint4 a = s4[0], b = s4[1];
        d4[0] = a;
        d4[1] = b;
    } else {
        for (u32 j = 0; j < P.stride; j++) dst[j] = src[j];
    }
    if (tag_in) tag_out[slot] = tag_in[i];
}
