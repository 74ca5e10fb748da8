// brisk_insert.hip -- the index: partition directory, arena, k_insert (one wave per partition), bucket bits, stats.
// Included by brisk_kernels.hip (one translation unit).
// ===========================================================================
// k_insert: persistent workgroups, each walking a strided share of the touched
// partitions.  Per partition (per chunk of at most WI_MAX_INST k-mer instances):
//   0. every k-mer instance of the chunk's records is expanded to its 128-bit
//      entry key in LDS (one wave per record, one lane per k-mer);
//   1. the instances are de-duplicated in an LDS table whose slots hold the index
//      of the first instance and a multiplicity;
//   2. the partition's existing entries stream through the table: a hit adds the
//      multiplicity to the entry's count (uint8_t, wraps; counter.cpp:264-268);
//   3. unmatched table entries are appended as new entries (count = multiplicity).
// Storage per partition: keys[] (u128) and counts[] (u8) in a bump-allocated arena.
// A partition that outgrows its slice moves to a fresh one taken from the
// workgroup's private arena chunk, so the global cursor sees one atomic per
// ARENA_CHUNK entries.  nb_kmers / nb_buckets are reductions done at stats() time:
// the kernel has no same-address global atomics on its data path.
#ifndef ARENA_CHUNK
#define ARENA_CHUNK 16384u   // entries a persistent wave takes from the global cursor at a time
#endif
// k_insert is bound by each wave's own serial instruction stream (LDS round trips, short dependent
// chains), so throughput follows the number of resident waves: chunks of 256 instances keep LDS at
// 10 KB and registers at 128 per wave => 4 waves per SIMD (512-instance chunks: 16 KB, 201 registers,
// 2 waves per SIMD, 61 ms instead of 50 ms on the 50M-read job; 128-instance chunks spill and split
// too many partitions: 84 ms).
#ifndef INSERT_SLOTS
#define INSERT_SLOTS 8192u   // private allocator slots: an upper bound of the persistent waves (256 CUs x 4 SIMDs x 8 waves)
#endif
#ifndef WI_WAVES_PER_EU
#define WI_WAVES_PER_EU 4
#endif
// Phase timers of k_insert (debug builds only: -DBRISK_PHASE_PROF): s_memtime at phase boundaries, scalar arithmetic,
// one atomic per phase and wave at the end.  Wall cycles of a wave, so they include what it waits for.
#ifdef BRISK_PHASE_PROF
__device__ unsigned long long g_phase[16];
#define PHASE_DECL unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_last = __builtin_amdgcn_s_memtime(); u32 dbg_cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PHASE(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_last; ph_last = t_; }  /* phase i ends here */
__device__ unsigned long long g_cnt[16];
#define PHASE_FLUSH { PHASE(10) if (threadIdx.x == 0) { _Pragma("unroll") for (int q_ = 0; q_ < 12; q_++) atomicAdd(&g_phase[q_], ph_acc[q_]); \
                                                         _Pragma("unroll") for (int q_ = 0; q_ < 12; q_++) atomicAdd(&g_cnt[q_], (unsigned long long)dbg_cnt[q_]); } }
#define CNT(i, v) dbg_cnt[i] += (v);
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#define CNT(i, v)
#endif
struct IndexDev {
    u64* keys;                   // 2 u64 per entry
    uint8_t* counts;
    DirEnt* dir;
    unsigned long long* cursor;  // arena entries handed out
    u32* bucket_bits;            // one bit per bucket id
    unsigned long long* stats;   // [3] garbage entries (abandoned slices)
    unsigned long long* slot_cur;  // per persistent workgroup: private chunk [cur, end)
    unsigned long long* slot_end;
    u32* ids;                    // entry-id mode only: stable dense id of every entry (insertion order)
    unsigned long long arena_cap;  // entries the arena can hold
    u32* err;                    // sticky violation bits: 1 scatter slot out of range, 2 arena exhausted, 4 chunk overflow
    u32 bits_check;              // few buckets (2b < 20): read a bucket-bitmap word before or-ing into it
    u32 huge_at;                 // partitions of more k-mer instances than this are left to k_insert_huge (0: none are)
    u32 key_words;               // u64 words per stored key: 1 where [routing id low bits | compacted k-mer | idx'] fits 64 bits
                                 // (shift + 2(k-b) + 6 <= 64: k31 b14, k31 b11, ...: 9 bytes per entry instead of 17), else 2
};
// an entry's key in the arena (KW: the word count where the caller knows it at compile time, 0: from ix)
template <u32 KW = 0>
__device__ __forceinline__ u128x load_key(const IndexDev& ix, unsigned long long at) {
    if ((KW ? KW : ix.key_words) == 1) return u128x{ix.keys[at], 0};
    return u128x{ix.keys[2 * at], ix.keys[2 * at + 1]};
}
template <u32 KW = 0>
__device__ __forceinline__ void store_key(const IndexDev& ix, unsigned long long at, u64 lo, u64 hi) {
    if ((KW ? KW : ix.key_words) == 1) {
        ix.keys[at] = lo;
    } else {
        ix.keys[2 * at] = lo;
        ix.keys[2 * at + 1] = hi;
    }
}
template <u32 KW = 0>
__device__ __forceinline__ void move_key(const IndexDev& ix, unsigned long long dst, unsigned long long src) {
    if ((KW ? KW : ix.key_words) == 1) {
        ix.keys[dst] = ix.keys[src];
    } else {
        ix.keys[2 * dst] = ix.keys[2 * src];
        ix.keys[2 * dst + 1] = ix.keys[2 * src + 1];
    }
}

// k_insert: ONE WAVE per partition, no workgroup barriers: every wave is an
// independent stream of partitions, so a CU keeps ~10 of them in flight and their
// LDS / HBM latencies overlap.  Sized for partitions of a few hundred k-mer
// instances (part_bits = 24 at b = 14: 16 buckets per partition).
#ifndef WI_MAX_INST
#define WI_MAX_INST 256     // k-mer instances per chunk
#endif
#define WI_TABLE (2 * WI_MAX_INST)   // LDS table slots (load <= 0.5)
static_assert(WI_MAX_INST % 256 == 0 && WI_MAX_INST <= 1024, "chunk size: whole 32-bit words of record marks per lane, 10-bit instance index");
#define WI_MAX_REC 64       // records per chunk: one per lane
#define WI_CNT_SHIFT 10     // table word = [MATCHED | multiplicity (21 b) | instance (10 b)]
#define WI_IDX_MASK 0x3ffu
#define wave_sync() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

// record words of the calling lane's record (lane r < n loads record first+r), 5 words at most
struct RecRegs {
    u64 w0, w1, w2, w3, w4;
};
__device__ __forceinline__ RecRegs load_rec_regs(const BriskParams& P, const u64* __restrict__ rec, u32 first, u32 n, u32 lane) {
    RecRegs r{0, 0, 0, 0, 0};
    if (lane < n) {
        const u64* c = rec + (u64)(first + lane) * P.stride;
        r.w0 = c[0];
        r.w1 = c[1];
        if (P.stride > 2) r.w2 = c[2];
        if (P.stride > 3) r.w3 = c[3];
        if (P.stride > 4) r.w4 = c[4];
    }
    return r;
}
// Where a partition's records are.  Classic (bin_cap == 0): `rec` holds all records in partition order, a partition's at
// [r_begin, r_begin + n_rec).  Binned (the scan wrote every record straight into its partition's bin):
// record i of partition `part` is rec[part * bin_cap + i] for i < bin_cap, and ovf[r_begin + i - bin_cap] beyond (the few
// partitions that overflow their bin; r_begin is then the partition's offset among the overflow records).
struct RecSrc {
    u64* rec;
    const u64* ovf;
    u32 bin_cap;
};
__device__ __forceinline__ RecRegs load_part_recs(const BriskParams& P, const RecSrc& src, u32 part, u32 r_begin, u32 first, u32 n, u32 lane) {
    if (!src.bin_cap) return load_rec_regs(P, src.rec, first, n, lane);
    RecRegs r{0, 0, 0, 0, 0};
    if (lane < n) {
        const u32 i = first - r_begin + lane;
        const u64* c = i < src.bin_cap ? src.rec + ((u64)part * src.bin_cap + i) * P.stride : src.ovf + ((u64)r_begin + i - src.bin_cap) * P.stride;
        r.w0 = c[0];
        r.w1 = c[1];
        if (P.stride > 2) r.w2 = c[2];
        if (P.stride > 3) r.w3 = c[3];
        if (P.stride > 4) r.w4 = c[4];
    }
    return r;
}
// Inclusive scans over the 64 lanes on the DPP network: row_shr 1,2,4,8 inside each row of 16, then row_bcast15
// and row_bcast31 carry the row totals over.  Lanes without a source keep the identity 0.  Full EXEC mask only.
#define WAVE_SCAN_STEP(x, OP, CTRL, ROW_MASK)                                                         \
    {                                                                                                 \
        const u32 y_ = (u32)__builtin_amdgcn_update_dpp(0, (int)(x), CTRL, ROW_MASK, 0xf, false); \
        x = OP(x, y_);                                                                                \
    }
__device__ __forceinline__ u32 op_add_u32(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 op_max_u32(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 wave_incl_scan(u32 x, u32 /*lane*/) {
    WAVE_SCAN_STEP(x, op_add_u32, 0x111, 0xf)  // row_shr:1
    WAVE_SCAN_STEP(x, op_add_u32, 0x112, 0xf)  // row_shr:2
    WAVE_SCAN_STEP(x, op_add_u32, 0x114, 0xf)  // row_shr:4
    WAVE_SCAN_STEP(x, op_add_u32, 0x118, 0xf)  // row_shr:8
    WAVE_SCAN_STEP(x, op_add_u32, 0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    WAVE_SCAN_STEP(x, op_add_u32, 0x143, 0xc)  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ u32 wave_incl_max_scan(u32 x) {
    WAVE_SCAN_STEP(x, op_max_u32, 0x111, 0xf)
    WAVE_SCAN_STEP(x, op_max_u32, 0x112, 0xf)
    WAVE_SCAN_STEP(x, op_max_u32, 0x114, 0xf)
    WAVE_SCAN_STEP(x, op_max_u32, 0x118, 0xf)
    WAVE_SCAN_STEP(x, op_max_u32, 0x142, 0xa)
    WAVE_SCAN_STEP(x, op_max_u32, 0x143, 0xc)
    return x;
}
// OR over all 64 lanes, the same value in every lane (through lane 63 and a scalar register).  Full EXEC mask only.
__device__ __forceinline__ u32 op_or_u32(u32 a, u32 b) { return a | b; }
__device__ __forceinline__ u32 wave_or_all(u32 x) {
    WAVE_SCAN_STEP(x, op_or_u32, 0x111, 0xf)
    WAVE_SCAN_STEP(x, op_or_u32, 0x112, 0xf)
    WAVE_SCAN_STEP(x, op_or_u32, 0x114, 0xf)
    WAVE_SCAN_STEP(x, op_or_u32, 0x118, 0xf)
    WAVE_SCAN_STEP(x, op_or_u32, 0x142, 0xa)
    WAVE_SCAN_STEP(x, op_or_u32, 0x143, 0xc)
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}
// value of the previous lane (0 for lane 0)
__device__ __forceinline__ u32 wave_prev_lane(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false); }  // wave_shr:1

#define WI_TS (WI_TABLE / 64)      // table words per lane

// k-mer j of record words in LDS, branch-free (every load is unconditional so that the
// unrolled instances of a lane keep their LDS reads in flight together)
template <u32 NW>  // NW > 0: compile-time record width; 0: P.nw
__device__ __forceinline__ u128x record_kmer_lds(const BriskParams& P, const u64* c, u32 n, u32 j) {
    const u32 s = 2 * (n - 1 - j);
    const u32 ws = s >> 6, bs = s & 63;
    const u32 last = (NW ? NW : P.nw) - 1;
    const u64 t0 = c[ws < last ? ws : last], t1 = c[ws + 1 < last ? ws + 1 : last], t2 = c[ws + 2 < last ? ws + 2 : last];
    const u64 a0 = ws <= last ? t0 : 0, a1 = ws + 1 <= last ? t1 : 0, a2 = ws + 2 <= last ? t2 : 0;
    u128x r;
    r.lo = bs ? ((a0 >> bs) | (a1 << (64 - bs))) : a0;
    r.hi = bs ? ((a1 >> bs) | (a2 << (64 - bs))) : a1;
    return and128(r, mask128(2 * P.kb));
}

#define WI_BATCH 64u   // partitions a wave takes per work-counter atomic (same-address atomics serialise device-wide)

// NI-instances-per-lane body of the expand + de-duplicate phases (NI = 4 when the
// chunk has <= 256 instances, else 8: all NI instances of a lane are in flight together)
template <u32 NI, u32 NW>
__device__ __forceinline__ void expand_and_dedupe(const BriskParams& P, u32 lane, u32 ninst, u32 tsize, const u64* s_rec, const u32* s_pref,
                                                  const uint8_t* s_irec, const u32* s_rmult, u64* s_key, u32* s_tab) {
    u64 klo[NI], khi[NI];
    u32 hh[NI], mult[NI];
    u32 rix[NI];
#pragma unroll
    for (u32 it = 0; it < NI; it++) {
        const u32 i = it * 64 + lane;
        rix[it] = s_irec[i < ninst ? i : 0];
    }
#pragma unroll
    for (u32 it = 0; it < NI; it++) {
        const u32 i = it * 64 + lane;
        const u32 r = rix[it];
        const u64* c = s_rec + r * (NW ? NW + 1 : P.stride);
        const u64 hdr = c[NW ? NW : P.nw];
        const u32 j = i < ninst ? i - s_pref[r] : 0;
        const u128x key = make_key(P, hdr_bucket(hdr), record_kmer_lds<NW>(P, c, hdr_n(hdr), j), hdr_idx0(hdr) + j);
        klo[it] = key.lo;
        khi[it] = key.hi;
        hh[it] = hash_key32(key) & (tsize - 1);
        mult[it] = (s_rmult[r] & 0xffu) << WI_CNT_SHIFT;  // counts wrap at 256: so may the multiplicities
        if (i < ninst) {
            s_key[2 * i] = key.lo;
            s_key[2 * i + 1] = key.hi;
        }
    }
    wave_sync();
    // de-duplicate: all of a lane's instances probe in lockstep rounds
    u32 pending = 0;
#pragma unroll
    for (u32 it = 0; it < NI; it++)
        if (it * 64 + lane < ninst) pending |= 1u << it;
    while (__any(pending != 0)) {
        u32 old[NI];
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            old[it] = EMPTY_SLOT;
            if (pending >> it & 1) old[it] = atomicCAS(&s_tab[hh[it]], EMPTY_SLOT, (it * 64 + lane) | mult[it]);
        }
        u64 olo[NI], ohi[NI];
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            const u32 oi = old[it] == EMPTY_SLOT ? 0 : (old[it] & WI_IDX_MASK);
            olo[it] = s_key[2 * oi];
            ohi[it] = s_key[2 * oi + 1];
        }
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            if (pending >> it & 1) {
                if (old[it] == EMPTY_SLOT) {
                    pending &= ~(1u << it);
                } else if (olo[it] == klo[it] && ohi[it] == khi[it]) {
                    atomicAdd(&s_tab[hh[it]], mult[it]);
                    pending &= ~(1u << it);
                } else {
                    hh[it] = (hh[it] + 1) & (tsize - 1);
                }
            }
        }
    }
}

// ---- compile-time record geometry (the common parameter sets): records are laid down in LDS as 32-bit words of
// (C << 6) followed by one packed info word, and k-mer j's 128-bit entry key is five adjacent words funnel-shifted
// (v_alignbit) by s = 2(n-1-j): bits [0,6) of the window then take idx', bits above 2kb+6 the routing id's low bits.
// 36 vector instructions per instance instead of 73 with the u64 shifts of record_kmer_lds + make_key + shl128.
template <u32 NW>
struct RecGeom {
    static constexpr u32 CW = 2 * NW + 1;     // words of C << 6
    static constexpr u32 RS = 2 * NW + 3;     // record stride in words (odd: lanes on different records hit different banks)
    static constexpr u32 INFO = 2 * NW + 1;   // [pref:10 | n:8 | idx0:8 | routing id low bits:6]
};
template <u32 NW>
__device__ __forceinline__ void store_rec_words(u32* dst, const RecRegs& rr, u32 info) {
    const u64 c0 = rr.w0, c1 = NW > 1 ? rr.w1 : 0, c2 = NW > 2 ? rr.w2 : 0, c3 = NW > 3 ? rr.w3 : 0;
    const u64 s0 = c0 << 6, s1 = (c1 << 6) | (c0 >> 58), s2 = (c2 << 6) | (c1 >> 58), s3 = (c3 << 6) | (c2 >> 58);
    const u32 top = (u32)((NW == 1 ? c0 : NW == 2 ? c1 : NW == 3 ? c2 : c3) >> 58);
    dst[0] = (u32)s0;
    dst[1] = (u32)(s0 >> 32);
    if (NW > 1) { dst[2] = (u32)s1; dst[3] = (u32)(s1 >> 32); }
    if (NW > 2) { dst[4] = (u32)s2; dst[5] = (u32)(s2 >> 32); }
    if (NW > 3) { dst[6] = (u32)s3; dst[7] = (u32)(s3 >> 32); }
    dst[2 * NW] = top;
    dst[RecGeom<NW>::INFO] = info;
    dst[RecGeom<NW>::INFO + 1] = 0;  // read (and shifted out) by the last k-mer's window
}
// What a lane keeps of its instances for the append (register arrays, compile-time indices only): their keys, the table
// slot each ended up in, and which of them CREATED their slot (the first copy of a key).
template <u32 NIMAX>
struct LaneInst {
    u64 klo[NIMAX], khi[NIMAX];
    u32 hh[NIMAX];
    u32 won;
};
template <u32 NI, u32 NW, u32 KB, u32 SHIFT, u32 NIMAX>
__device__ __forceinline__ void expand_and_dedupe_words(u32 lane, u32 ninst, u32 tsize, const u32* s_rw, const uint8_t* s_irec, const u32* s_rmult,
                                                        u64* s_key, u32* s_tab, u32* dbg_rounds, LaneInst<NIMAX>& li) {
    constexpr u32 RS = RecGeom<NW>::RS, INFO = RecGeom<NW>::INFO, KBITS = 2 * KB + 6;
    static_assert(KBITS + SHIFT <= 128 && SHIFT <= 6, "entry key: [routing id low bits | compacted k-mer | idx']");
    static_assert(NI <= NIMAX, "instances per lane");
    u64 (&klo)[NIMAX] = li.klo;
    u64 (&khi)[NIMAX] = li.khi;
    u32 (&hh)[NIMAX] = li.hh;
    u32 mult[NI];
    u32 rix[NI];
    u32 won = 0;
#pragma unroll
    for (u32 it = 0; it < NI; it++) {
        const u32 i = it * 64 + lane;
        rix[it] = s_irec[i < ninst ? i : 0];
    }
#pragma unroll
    for (u32 it = 0; it < NI; it++) {
        const u32 i = it * 64 + lane;
        const u32* base = s_rw + rix[it] * RS;
        const u32 info = base[INFO];
        const u32 j = i < ninst ? i - (info & 0x3ffu) : 0;
        const u32 n = (info >> 10) & 0xffu;
        const u32 s = 2 * (n - 1 - j);
        const u32* wp = base + (s >> 5);
        const u32 sh = s & 31;
        const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4];
        u32 k0 = __builtin_amdgcn_alignbit(w1, w0, sh), k1 = __builtin_amdgcn_alignbit(w2, w1, sh);
        u32 k2 = __builtin_amdgcn_alignbit(w3, w2, sh), k3 = __builtin_amdgcn_alignbit(w4, w3, sh);
        k0 = (k0 & ~0x3fu) | (((info >> 18) & 0xffu) + j);  // idx' = idx0' + j (SuperKmerLight.hpp:98)
        // keep the 2kb + 6 key bits, put the routing id's low bits on top
        constexpr u32 m0 = KBITS >= 32 ? ~0u : (1u << KBITS) - 1, m1 = KBITS >= 64 ? ~0u : KBITS <= 32 ? 0u : (1u << (KBITS - 32)) - 1;
        constexpr u32 m2 = KBITS >= 96 ? ~0u : KBITS <= 64 ? 0u : (1u << (KBITS - 64)) - 1, m3 = KBITS >= 128 ? ~0u : KBITS <= 96 ? 0u : (1u << (KBITS - 96)) - 1;
        k0 &= m0; k1 &= m1; k2 &= m2; k3 &= m3;
        u64 lo = ((u64)k1 << 32) | k0, hi = ((u64)k3 << 32) | k2;
        if (SHIFT) {
            const u64 rl = (info >> 26) & ((1u << SHIFT) - 1);
            if (KBITS >= 64) hi |= rl << (KBITS - 64);
            else {
                lo |= rl << KBITS;
                if (KBITS + SHIFT > 64) hi |= rl >> (64 - KBITS);
            }
        }
        klo[it] = lo;
        khi[it] = hi;
        hh[it] = hash_key32(mk128(lo, hi)) & (tsize - 1);
        mult[it] = (s_rmult[rix[it]] & 0xffu) << WI_CNT_SHIFT;  // counts wrap at 256: so may the multiplicities
        if (i < ninst) {
            s_key[2 * i] = lo;
            s_key[2 * i + 1] = hi;
        }
    }
    wave_sync();  // every lane has read its records: the table may take their place
    for (u32 w = 0; w * 64 < tsize; w++) s_tab[w * 64 + lane] = EMPTY_SLOT;
    wave_sync();
    u32 pending = 0;
#pragma unroll
    for (u32 it = 0; it < NI; it++)
        if (it * 64 + lane < ninst) pending |= 1u << it;
    while (__any(pending != 0)) {
        *dbg_rounds += NI;
        u32 old[NI];
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            old[it] = EMPTY_SLOT;
            if (pending >> it & 1) old[it] = atomicCAS(&s_tab[hh[it]], EMPTY_SLOT, (it * 64 + lane) | mult[it]);
        }
        u64 olo[NI], ohi[NI];
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            const u32 oi = old[it] == EMPTY_SLOT ? 0 : (old[it] & WI_IDX_MASK);
            olo[it] = s_key[2 * oi];
            ohi[it] = s_key[2 * oi + 1];
        }
#pragma unroll
        for (u32 it = 0; it < NI; it++) {
            if (pending >> it & 1) {
                if (old[it] == EMPTY_SLOT) {
                    pending &= ~(1u << it);
                    won |= 1u << it;
                } else if (olo[it] == klo[it] && ohi[it] == khi[it]) {
                    atomicAdd(&s_tab[hh[it]], mult[it]);
                    pending &= ~(1u << it);
                } else {
                    hh[it] = (hh[it] + 1) & (tsize - 1);
                }
            }
        }
    }
    li.won = won;
}

// Record-level de-duplication of the <= 64 records the lanes hold (also in s_rec): the first copy of every distinct record
// survives, s_rmult[its lane] = the multiplicities of all its copies added up; returns whether this lane's record is a
// later copy.  Header bits 48..55 carry a record's multiplicity (mod 256: counts wrap there anyway) once a partition's
// records have been collapsed; they are not part of its identity.
#define HDR_ID_MASK 0x0000ffffffffffffull
#define HDR_HAS_MULT (1ull << 56)   // header bits 48..55 hold the record's multiplicity (mod 256, as the counts are)
__device__ __forceinline__ bool dedupe_records(u32 stride, const RecRegs& rr, u32 my_mult, u32 nrec, u32 lane, const u64* s_rec, u32* s_rtab, u32* s_rmult) {
    s_rtab[lane] = EMPTY_SLOT;
    s_rtab[lane + 64] = EMPTY_SLOT;
    s_rmult[lane] = my_mult;
    wave_sync();
    bool dup = false;
    if (lane < nrec) {
        const u64 k1 = stride == 2 ? HDR_ID_MASK : ~0ull, k2 = stride == 3 ? HDR_ID_MASK : ~0ull, k3 = stride == 4 ? HDR_ID_MASK : ~0ull,
                  k4 = stride == 5 ? HDR_ID_MASK : ~0ull;
        const u64 w1 = rr.w1 & k1, w2 = rr.w2 & k2, w3 = rr.w3 & k3, w4 = rr.w4 & k4;
        // a weak hash is enough for <= 64 records in 128 slots: rotate-xor fold, one 32-bit multiply
        const u64 z = rr.w0 ^ ((w1 << 17) | (w1 >> 47)) ^ ((w2 << 31) | (w2 >> 33)) ^ ((w3 << 47) | (w3 >> 17)) ^ w4;
        u32 h = ((((u32)z ^ (u32)(z >> 32)) * 0x9E3779B1u) >> 20) & (2 * WI_MAX_REC - 1);
        for (;;) {
            const u32 o = atomicCAS(&s_rtab[h], EMPTY_SLOT, lane);
            if (o == EMPTY_SLOT) break;
            const u64* oc = s_rec + o * stride;
            bool same = oc[0] == rr.w0 && (oc[1] & k1) == w1;
            if (stride > 2) same = same && (oc[2] & k2) == w2;
            if (stride > 3) same = same && (oc[3] & k3) == w3;
            if (stride > 4) same = same && (oc[4] & k4) == w4;
            if (same) {
                atomicAdd(&s_rmult[o], my_mult);
                dup = true;
                break;
            }
            h = (h + 1) & (2 * WI_MAX_REC - 1);
        }
    }
    return dup;
}

// MAXI: k-mer instances per chunk.  256 (10 KB of LDS, 128 registers: 4 waves per SIMD) for the usual partitions of a
// few hundred instances; 512 (2 waves per SIMD) when partitions are big -- few distinct minimizers, as with m <= 11 --
// and the passes over a partition's entries saved by half as many chunks outweigh the occupancy.
// NW > 0: the record width (u64 words of the compacted super-k-mer), kb = k - b and the routing-id bits kept in the
// key are compile-time constants (the common parameter sets; the host picks the instantiation): shifts, masks and
// strides fold, nothing of BriskParams stays in scalar registers (the generic body spills 48 of them into vector
// lanes and pays a v_readlane per use), and k-mers are cut out of 32-bit words (expand_and_dedupe_words).  NW == 0:
// everything from P at run time.
template <u32 MAXI, u32 NW, u32 KB, u32 SHIFT>
__device__ __forceinline__ void insert_body(const BriskParams& PP, const RecSrc& src, const PartDesc* __restrict__ desc, u32 n_touched, const IndexDev& ix,
                                            u32* __restrict__ work_counter) {
    u64* const rec = src.rec;  // the in-place collapse of the big-partition kernel (classic layout only)
    BriskParams P = PP;
    if (NW) {  // the fields the body reads, as constants
        P.nw = NW;
        P.stride = NW + 1;
        P.kb = KB;
        P.shift = SHIFT;
    }
    constexpr u32 TABLE = 2 * MAXI, TS = TABLE / 64, NI = MAXI / 64;
    constexpr u32 KW = NW ? (2 * KB + 6 + SHIFT <= 64 ? 1u : 2u) : 0u;  // words of a stored key (0: ix.key_words)
    static_assert(MAXI % 256 == 0 && MAXI <= 1024, "chunk size: whole 32-bit words of record marks per lane, 10-bit instance index");
    // LDS of one wave, one buffer cut into regions.  k_insert's throughput follows the number of resident waves almost
    // linearly (4096 -> 30.8 ms, 3072 -> 38.3, 2048 -> 54.2, 1024 -> 103.6 per 50 M reads: every wave is a serial chain of
    // LDS and memory round trips), so what is not live at the same time shares its bytes.  Compile-time geometry (NW > 0):
    // the probe table lies over the records' region -- the records are last read when the keys are built, the table is
    // first written right after -- and nothing of the generic body's prefix array exists: 7,680 B per wave instead of
    // 10,000, i.e. 21 waves per CU by LDS instead of 16.
    //   s_key   [2 * MAXI] u64   the instances' keys                          expand .. stream
    //   s_rec   [REC_U64]  u64   records as u64 words (record-level pass), then as shifted 32-bit words (s_rw; + 4 words:
    //                            the last record's window reads two words past its slot), generic: then the new entries' list
    //   s_tab   [TABLE]    u32   probe table                                   NW > 0: over s_rec
    //   s_rtab, s_rmult, s_irec, (generic: s_pref)
    constexpr u32 REC_U64 = WI_MAX_REC * 5 > MAXI / 2 ? WI_MAX_REC * 5 : MAXI / 2;
    static_assert(NW == 0 || (WI_MAX_REC * RecGeom<NW ? NW : 1>::RS + 4) * 4 <= REC_U64 * 8, "shifted record words must fit the record buffer");
    constexpr bool TAB_OVER_REC = NW && TABLE * 4 <= REC_U64 * 8;  // (the 512-instance body's table is larger than its record buffer)
    constexpr u32 OFF_REC = 16 * MAXI, OFF_TAB = TAB_OVER_REC ? OFF_REC : OFF_REC + 8 * REC_U64, OFF_RTAB = TAB_OVER_REC ? OFF_REC + 8 * REC_U64 : OFF_TAB + 4 * TABLE;
    constexpr u32 OFF_RMULT = OFF_RTAB + 4 * 2 * WI_MAX_REC, OFF_IREC = OFF_RMULT + 4 * WI_MAX_REC, OFF_PREF = OFF_IREC + MAXI;
    constexpr u32 LDS_BYTES = NW ? OFF_PREF : OFF_PREF + 4 * (WI_MAX_REC + 1) + 4;
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[LDS_BYTES];
    u64* s_key = (u64*)s_mem;
    u64* s_rec = (u64*)(s_mem + OFF_REC);
    u32* s_rw = (u32*)s_rec;
    u32* s_tab = (u32*)(s_mem + OFF_TAB);
    u32* s_list = (u32*)s_rec;  // generic body: [MAXI] the new entries' table words, built after the records have been expanded
    u32* s_rtab = (u32*)(s_mem + OFF_RTAB);
    u32* s_rmult = (u32*)(s_mem + OFF_RMULT);
    uint8_t* s_irec = s_mem + OFF_IREC;
    u32* s_pref = (u32*)(s_mem + (NW ? 0 : OFF_PREF));  // generic body only

    const u32 lane = threadIdx.x;
    PHASE_DECL
    unsigned long long acur = ix.slot_cur[blockIdx.x], aend = ix.slot_end[blockIdx.x], garbage = 0;
    const u32 kbits = 2 * P.kb + 6;

    for (;;) {
        // ---- take the next batch of partitions (one atomic per WI_BATCH partitions)
        u32 t0 = 0;
        if (lane == 0) t0 = atomicAdd(work_counter, WI_BATCH);
        t0 = (u32)__builtin_amdgcn_readfirstlane((int)t0);
        if (t0 >= n_touched) break;
        const u32 t_end = min(t0 + WI_BATCH, n_touched);
        // A partition's descriptor comes by scalar loads (s_load_dwordx8: t and the array's address are wave-uniform, the array is
        // read-only).  Round 2 kept the batch's 64 descriptors in registers, one per lane, and broadcast one with eight v_readlane:
        // those were the eight registers the body spilled at 96 (10 spilled -> 3; 26.2 -> 25.9 ms per 50 M reads).
        PartDesc d = desc[t0];
        RecRegs rr = load_part_recs(P, src, d.part, d.r_begin, d.r_begin, min(d.n_rec, (u32)WI_MAX_REC), lane);

        for (u32 t = t0; t < t_end; t++) {
            const u32 tn = t + 1;
            PartDesc dn{};
            if (tn < t_end) dn = desc[tn];
            // The next partition's records are requested when this partition is done, into rr.  Round 2 requested them at the top
            // of the current partition, into a second set of registers: the memory counter retires loads in issue order and the
            // compiler's wait in front of the first use of rr is vmcnt(0), so every partition began by sitting out the round trip
            // of the prefetch it had just issued, and kept eight more registers live for it (31.1 ms per 50 M reads against 30.0
            // this way; requesting them in front of the partition's last stores, or where the last chunk has expanded its
            // records, measured 32.6 / 32.3: DESIGN.md section 4).
#define NEXT_PARTITION                                                                                                  \
    {                                                                                                                   \
        d = dn;                                                                                                         \
        if (tn < t_end) rr = load_part_recs(P, src, d.part, d.r_begin, d.r_begin, min(d.n_rec, (u32)WI_MAX_REC), lane); \
    }

            // rr must have landed on EVERY path before it is requested again at the end of the partition (a load's destination
            // is not written while the load is in flight), a partition left to k_insert_huge or one whose lanes all skip the
            // store below included: without this unconditional use the compiler waits at the end of every partition instead --
            // for all the stores the partition has just issued.
            asm volatile("" ::"v"(rr.w0), "v"(rr.w1), "v"(rr.w2), "v"(rr.w3), "v"(rr.w4));
            do {  // (one pass: a partition that is not this kernel's leaves through `break`)
            if (ix.huge_at && d.n_inst > ix.huge_at) break;  // a block of 16 waves takes this one (k_insert_huge)
            CNT(0, 1)
            const u32 part = d.part;
            u32 r_end = d.r_begin + d.n_rec;
            u32 n_exist = d.n_exist;
            u32 inst_left = d.n_inst;  // instances not yet processed: bounds the final size
            unsigned long long off = d.off;
            u32 cap = d.cap;
            u32 bm0 = 0, bm1 = 0;

            // A partition of many records (one hot bucket) first collapses its records window by window, in place:
            // the chunks below then see each distinct record of a window once, with its multiplicity in the header,
            // and far fewer chunks -- each of which streams the partition's entries -- are needed.
            // (only in the big-partition kernel: the usual one is 2-3 % slower with this path compiled in)
            bool collapsed = false;
            if (MAXI > WI_MAX_INST && !src.bin_cap && d.n_rec > 2 * WI_MAX_REC) {
                u32 wr = d.r_begin;
                for (u32 rd = d.r_begin; rd < r_end; rd += WI_MAX_REC) {
                    const u32 avail = min(r_end - rd, (u32)WI_MAX_REC);
                    if (rd != d.r_begin) rr = load_rec_regs(P, rec, rd, avail, lane);
                    wave_sync();
                    if (lane < avail) {
                        u64* dst = s_rec + lane * P.stride;
                        dst[0] = rr.w0;
                        dst[1] = rr.w1;
                        if (P.stride > 2) dst[2] = rr.w2;
                        if (P.stride > 3) dst[3] = rr.w3;
                        if (P.stride > 4) dst[4] = rr.w4;
                    }
                    // a record may carry a multiplicity already (brisk_hip_reallocate: one k-mer with its entry's count): it is what
                    // the record adds to its survivor, and it is replaced, not OR-ed over, in the survivor's header
                    const u64 hdr_c = P.stride == 2 ? rr.w1 : P.stride == 3 ? rr.w2 : P.stride == 4 ? rr.w3 : rr.w4;
                    const u32 mult_c = (hdr_c & HDR_HAS_MULT) ? (u32)(hdr_c >> 48) & 0xffu : 1u;
                    const bool dup = dedupe_records(P.stride, rr, mult_c, avail, lane, s_rec, s_rtab, s_rmult);
                    wave_sync();
                    const bool keep = lane < avail && !dup;
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {  // survivors move to the front of the partition's records (never past what is still to be read)
                        u64* dst = rec + (u64)(wr + (u32)__popcll(bal & lanes_below(lane))) * P.stride;
                        const u64 keep_id = HDR_ID_MASK, mult = ((u64)(s_rmult[lane] & 0xffu) << 48) | HDR_HAS_MULT;
                        dst[0] = rr.w0;
                        dst[1] = P.stride == 2 ? (rr.w1 & keep_id) | mult : rr.w1;
                        if (P.stride > 2) dst[2] = P.stride == 3 ? (rr.w2 & keep_id) | mult : rr.w2;
                        if (P.stride > 3) dst[3] = P.stride == 4 ? (rr.w3 & keep_id) | mult : rr.w3;
                        if (P.stride > 4) dst[4] = (rr.w4 & keep_id) | mult;
                    }
                    wr += (u32)__popcll(bal);
                }
                collapsed = true;
                r_end = wr;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the chunks read what was just written: same wave, same CU
                rr = load_rec_regs(P, rec, d.r_begin, min(r_end - d.r_begin, (u32)WI_MAX_REC), lane);
            }

            for (u32 rc = d.r_begin; rc < r_end;) {
                // ---- pick the chunk: up to WI_MAX_REC records / MAXI instances
                PHASE(0)
                CNT(1, 1)
                const u32 avail = min(r_end - rc, (u32)WI_MAX_REC);
                if (rc != d.r_begin) rr = load_part_recs(P, src, part, d.r_begin, rc, avail, lane);
                wave_sync();
                if (lane < avail) {
                    u64* dst = s_rec + lane * P.stride;
                    dst[0] = rr.w0;
                    dst[1] = rr.w1;
                    if (P.stride > 2) dst[2] = rr.w2;
                    if (P.stride > 3) dst[3] = rr.w3;
                    if (P.stride > 4) dst[4] = rr.w4;
                }
                const u64 my_hdr = P.stride == 2 ? rr.w1 : P.stride == 3 ? rr.w2 : P.stride == 4 ? rr.w3 : rr.w4;
                const u32 raw_n = lane < avail ? hdr_n(my_hdr) : 0;
                // a record stands for one vector, unless its header says how many (collapsed records above; the entries of
                // an index being re-bucketed, brisk_hip_reallocate: one k-mer with its count)
                const u32 my_mult = (my_hdr & HDR_HAS_MULT) ? (u32)(my_hdr >> 48) & 0xffu : 1u;
                const u32 x0 = wave_incl_scan(raw_n, lane);
                // First try every available record: identical records (the same super-k-mer seen in
                // several reads) collapse into one with a multiplicity, so far more raw instances fit.
                // If the collapsed chunk is still too big, shrink to the prefix whose collapsed count fits (counted
                // again on its own it can come out a little higher, once the first copy of a record lies beyond
                // it: hence the loop), at the latest to the raw-count prefix, which always fits.
                const u32 rawfit = (u32)__popcll(__ballot(lane < avail && x0 <= MAXI));  // >= 1; a prefix: x0 is monotone
                u32 nrec = avail, my_n = 0, x = 0, ninst = 0;
                for (int attempt = 0;; attempt++) {
                    CNT(2, 1)
                    const bool dup = dedupe_records(P.stride, rr, my_mult, nrec, lane, s_rec, s_rtab, s_rmult);
                    my_n = (lane < nrec && !dup) ? raw_n : 0;
                    x = wave_incl_scan(my_n, lane);
                    ninst = __shfl(x, 63, 64);
                    if (ninst <= MAXI) break;
                    const u32 fit = (u32)__popcll(__ballot(lane < nrec && x <= MAXI));  // x is monotone too
                    nrec = (attempt >= 2 || fit <= rawfit) ? rawfit : min(fit, nrec - 1);
                    wave_sync();
                }
                PHASE(1)
                const u32 raw_inst = __shfl(x0, nrec - 1, 64);
                // (a quarter full at most in instances, less in distinct keys: probe rounds are wave-wide -- every round costs all 64 lanes
                // whoever is still probing -- so a sparser table is worth its clearing: 25.8 -> 25.0 ms per 50 M reads against half full.
                // Skipping, by a scalar branch, the instance slots nobody is pending in lost: 28.7 ms -- seven more spilled registers)
                u32 tsize = 128;
                while (tsize < 4 * ninst && tsize < TABLE) tsize <<= 1;
                if (!NW) {  // (compile-time geometry: the prefix travels in the records' info words, and the table -- which lies
                            // over the records there -- is cleared once the keys have been built from them)
                    s_pref[lane + 1] = x;
                    if (lane == 0) s_pref[0] = 0;
#pragma unroll
                    for (u32 w = 0; w < TS; w++)
                        if (w * 64 < tsize) s_tab[w * 64 + lane] = EMPTY_SLOT;
                }
                {
                    // instance -> record: each record marks its first instance, a running maximum spreads the marks
                    // (records lie in lane order).  Every lane owns MAXI/64 consecutive instances here.
                    u32* irec32 = (u32*)s_irec;
#pragma unroll
                    for (u32 q = 0; q < MAXI / 256; q++) irec32[q * 64 + lane] = 0;
                    wave_sync();
                    if (my_n) s_irec[x - my_n] = (uint8_t)(lane + 1);
                    wave_sync();
                    u32 wv[MAXI / 256], run = 0;
#pragma unroll
                    for (u32 q = 0; q < MAXI / 256; q++) {
                        wv[q] = irec32[lane * (MAXI / 256) + q];
                        run = op_max_u32(run, op_max_u32(op_max_u32(wv[q] & 0xff, (wv[q] >> 8) & 0xff), op_max_u32((wv[q] >> 16) & 0xff, wv[q] >> 24)));
                    }
                    u32 carry = wave_prev_lane(wave_incl_max_scan(run));  // the last mark before this lane's instances
#pragma unroll
                    for (u32 q = 0; q < MAXI / 256; q++) {
                        const u32 b0 = op_max_u32(carry, wv[q] & 0xff), b1 = op_max_u32(b0, (wv[q] >> 8) & 0xff);
                        const u32 b2 = op_max_u32(b1, (wv[q] >> 16) & 0xff), b3 = op_max_u32(b2, wv[q] >> 24);
                        carry = b3;
                        // marks are lane + 1; instances past the last record (none are read) may hold 0 - 1
                        irec32[lane * (MAXI / 256) + q] = ((b0 - 1) & 0xff) | (((b1 - 1) & 0xff) << 8) | (((b2 - 1) & 0xff) << 16) | ((b3 - 1) << 24);
                    }
                }
                wave_sync();
                // ---- 0/1. expand to entry keys and de-duplicate
                PHASE(2)
                u32 dbg_r = 0;
                LaneInst<NI> li;
                li.won = 0;
                CNT(3, (ninst + 63) / 64)
                CNT(4, ninst)
                CNT(5, nrec)
                if (NW) {
                    // the records once more, as 32-bit words of C << 6 with their packed info (the u64 copy was for the
                    // record-level pass above; its last reader is behind the wave_sync before the instance map)
                    if (lane < nrec) {
                        const u32 info = (x - my_n) | (raw_n << 10) | (hdr_idx0(my_hdr) << 18) | ((hdr_bucket(my_hdr) & ((1u << SHIFT) - 1)) << 26);
                        store_rec_words<NW ? NW : 1>(s_rw + lane * RecGeom<NW ? NW : 1>::RS, rr, info);
                    }
                    wave_sync();
                    if (ninst <= 64) expand_and_dedupe_words<1, NW ? NW : 1, KB, SHIFT, NI>(lane, ninst, tsize, s_rw, s_irec, s_rmult, s_key, s_tab, &dbg_r, li);
                    else if (ninst <= 128) expand_and_dedupe_words<2, NW ? NW : 1, KB, SHIFT, NI>(lane, ninst, tsize, s_rw, s_irec, s_rmult, s_key, s_tab, &dbg_r, li);
                    else if (ninst <= 192) expand_and_dedupe_words<3, NW ? NW : 1, KB, SHIFT, NI>(lane, ninst, tsize, s_rw, s_irec, s_rmult, s_key, s_tab, &dbg_r, li);
                    else if (NI <= 4 || ninst <= 256) expand_and_dedupe_words<4, NW ? NW : 1, KB, SHIFT, NI>(lane, ninst, tsize, s_rw, s_irec, s_rmult, s_key, s_tab, &dbg_r, li);
                    else expand_and_dedupe_words<NI, NW ? NW : 1, KB, SHIFT, NI>(lane, ninst, tsize, s_rw, s_irec, s_rmult, s_key, s_tab, &dbg_r, li);
                } else {
                    if (ninst <= 128) expand_and_dedupe<2, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else if (NI <= 4 || ninst <= 256) expand_and_dedupe<4, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else expand_and_dedupe<NI, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                }
                wave_sync();
                CNT(8, dbg_r)
                PHASE(3)

                // ---- 2. existing entries probe the table.  After the first chunk they include what this wave
                // appended itself: same wave, same CU, so those stores only have to be complete (workgroup
                // scope; __threadfence() would write back and invalidate the XCD's whole L2), and waiting
                // for them here rather than at the end of the last chunk hides them behind phases 0 and 1.
                if (rc != d.r_begin) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                // Four strides of entries at a time: their key and count loads are in flight together (a big partition
                // of a hot bucket streams thousands of entries per chunk; one dependent load per stride was most
                // of this kernel's time at k31/b11).
                for (u32 e0 = 0; e0 < n_exist; e0 += 4 * 64) {
                    u64 klo[4], khi[4];
                    uint8_t cnt[4];
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        const u32 e = e0 + q * 64 + lane;
                        const unsigned long long at = off + (e < n_exist ? e : 0);
                        const u128x kq = load_key<KW>(ix, at);
                        klo[q] = kq.lo;
                        khi[q] = kq.hi;
                        cnt[q] = ix.counts[at];
                    }
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        const u32 e = e0 + q * 64 + lane;
                        if (e >= n_exist) continue;
                        u32 h = hash_key32(mk128(klo[q], khi[q])) & (tsize - 1);
                        for (;;) {
                            const u32 v = s_tab[h];
                            if (v == EMPTY_SLOT) break;
                            const u32 i = v & WI_IDX_MASK;
                            if (s_key[2 * i] == klo[q] && s_key[2 * i + 1] == khi[q]) {
                                ix.counts[off + e] = (uint8_t)(cnt[q] + ((v & ~MATCHED_BIT) >> WI_CNT_SHIFT));
                                s_tab[h] = v | MATCHED_BIT;
                                break;
                            }
                            h = (h + 1) & (tsize - 1);
                        }
                    }
                }
                wave_sync();

                PHASE(5)
                // ---- 3. append the unmatched ones.  Generic body: the table is swept, the new entries' table words are
                // compacted in LDS and written out by full waves.  Compile-time geometry: every lane still holds its
                // instances' keys and slots and knows which of them created their slot, so the new entries are ranked with
                // one ballot per instance slot and written straight from registers -- no sweep of the table (8 rounds for
                // 512 slots), no list, no keys read back from LDS: 160 -> ~75 vector instructions per partition.
                u32 n_new = 0;
                u32 new_mask = 0, new_rank[NI], new_word[NI];
                if (NW) {
#pragma unroll
                    for (u32 it = 0; it < NI; it++) {
                        new_rank[it] = 0;
                        new_word[it] = 0;
                        if (it * 64 < ninst) {  // wave-uniform
                            const bool made = (li.won >> it) & 1;
                            if (made) new_word[it] = s_tab[li.hh[it]];
                            const bool is_new = made && !(new_word[it] & MATCHED_BIT);
                            const unsigned long long bal = __ballot(is_new);
                            new_rank[it] = n_new + (u32)__popcll(bal & lanes_below(lane));
                            if (is_new) new_mask |= 1u << it;
                            n_new += (u32)__popcll(bal);
                        }
                    }
                } else {
#pragma unroll
                    for (u32 w = 0; w < TS; w++) {
                        if (w * 64 < tsize) {
                            const u32 v = s_tab[w * 64 + lane];
                            const bool is_new = v != EMPTY_SLOT && !(v & MATCHED_BIT);
                            const unsigned long long bal = __ballot(is_new);
                            if (is_new) s_list[n_new + (u32)__popcll(bal & lanes_below(lane))] = v;
                            n_new += (u32)__popcll(bal);
                        }
                    }
                }
                wave_sync();
#ifndef INSERT_NO_EARLY_WAIT
                // The prefetched records are waited for HERE, before this chunk's stores are issued: vmcnt retires in issue
                // order, so a wait placed behind the stores (the next partition's first use) would sit out the stores' round
                // trip as well.  They were requested a partition's worth of work ago.
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#endif

                PHASE(6)
                CNT(6, (n_new + 63) / 64)
                CNT(7, n_new)
                inst_left -= raw_inst;
                if (n_exist + n_new > cap) {
                    // move to a fresh slice, sized so that this partition moves at most once per batch
                    const unsigned long long want = grow_cap(n_exist + n_new + inst_left);
                    unsigned long long noff;
                    if (want > ARENA_CHUNK / 8) {
                        // a large slice goes straight to the global cursor: the private chunk never strands more
                        // than a small request (< ARENA_CHUNK/8) at a refill, which the host's reserve covers
                        unsigned long long got = 0;
                        if (lane == 0) got = atomicAdd(ix.cursor, want);
                        noff = __shfl(got, 0, 64);
                    } else {
                        if (acur + want > aend) {  // private chunk exhausted: abandon its tail, take a new one
                            garbage += aend - acur;
                            unsigned long long got = 0;
                            if (lane == 0) got = atomicAdd(ix.cursor, (unsigned long long)ARENA_CHUNK);
                            acur = __shfl(got, 0, 64);
                            aend = acur + ARENA_CHUNK;
                        }
                        noff = acur;
                        acur += want;
                    }
                    if (noff + want > ix.arena_cap || ninst > MAXI) {  // must not happen (host reserves the bound): drop, flag
                        if (lane == 0) atomicOr(ix.err, ninst > MAXI ? 4u : 2u);
                        break;
                    }
                    garbage += cap;
                    cap = (u32)want;
                    for (u32 e = lane; e < n_exist; e += 64) {
                        move_key<KW>(ix, noff + e, off + e);
                        ix.counts[noff + e] = ix.counts[off + e];
                    }
                    off = noff;
                }
                PHASE(7)
                if (NW) {
#pragma unroll
                    for (u32 it = 0; it < NI; it++) {
                        if ((new_mask >> it) & 1) {
                            const unsigned long long at = off + n_exist + new_rank[it];
                            store_key<KW>(ix, at, li.klo[it], li.khi[it]);
                            ix.counts[at] = (uint8_t)(new_word[it] >> WI_CNT_SHIFT);
                            const u32 bl = P.shift ? ((u32)shr128(mk128(li.klo[it], li.khi[it]), kbits).lo & ((1u << P.shift) - 1)) : 0;
                            const u32 bb = P.shift > 6 ? (bl >> (P.shift - 6)) : bl;  // 64 bins at most
                            if (bb < 32) bm0 |= 1u << bb; else bm1 |= 1u << (bb - 32);
                        }
                    }
                } else {
                    for (u32 q = lane; q < n_new; q += 64) {
                        const u32 v = s_list[q];
                        const u32 i = v & WI_IDX_MASK;
                        const u64 klo2 = s_key[2 * i], khi2 = s_key[2 * i + 1];
                        const unsigned long long at = off + n_exist + q;
                        store_key<KW>(ix, at, klo2, khi2);
                        ix.counts[at] = (uint8_t)(v >> WI_CNT_SHIFT);
                        // bucket id inside the partition: the key's top `shift` bits (<= 6 of them used here)
                        const u32 bl = P.shift ? ((u32)shr128(mk128(klo2, khi2), kbits).lo & ((1u << P.shift) - 1)) : 0;
                        const u32 bb = P.shift > 6 ? (bl >> (P.shift - 6)) : bl;  // 64 bins at most
                        if (bb < 32) bm0 |= 1u << bb; else bm1 |= 1u << (bb - 32);
                    }
                }
                n_exist += n_new;
                rc += nrec;
                PHASE(8)

            }
            if (P.shift <= 6) {  // OR the lanes' bucket bits together on the DPP network (an LDS atomicOr on one word is turned
                                 // by the compiler into a scalar loop over the active lanes: 64 rounds per partition)
                bm0 = wave_or_all(bm0);
                if (P.shift == 6) bm1 = wave_or_all(bm1);
            }
            if (lane == 0) ix.dir[part] = DirEnt{off, n_exist, cap};
            // bucket occupancy bits: exact when a partition holds <= 64 buckets (shift <= 6);
            // partitions of more buckets are handled by k_bucket_bits below
            if (P.shift <= 6 && lane < 2) {
                const u32 mask = lane == 0 ? bm0 : bm1;
                const u32 nb = 1u << P.shift;  // buckets per partition
                const u64 first = ((u64)part << P.shift) >> P.ext_bits;  // ext_bits > 0 => shift == 0: the one bucket this partition is a slice of
                // With few buckets (small b) every partition of a bucket would hit the same word, and same-address atomics
                // serialise device-wide: there a bit that is already set is not set again (bits_check).  With many buckets
                // the words are all different, and the read before the atomic would be a dependent load behind this
                // partition's stores: the wave would sit out their whole round trip (it was a third of k_insert's time).
                if (mask) {
                    if (nb >= 32) {
                        if (lane * 32 < nb && (!ix.bits_check || (ix.bucket_bits[(first >> 5) + lane] & mask) != mask)) atomicOr(&ix.bucket_bits[(first >> 5) + lane], mask);
                    } else if (lane == 0) {
                        const u32 bits = mask << (first & 31);
                        if (!ix.bits_check || (ix.bucket_bits[first >> 5] & bits) != bits) atomicOr(&ix.bucket_bits[first >> 5], bits);
                    }
                }
            }
            } while (0);
            NEXT_PARTITION
#undef NEXT_PARTITION
            PHASE(9)
        }
    }
    PHASE_FLUSH
    if (lane == 0) {
        ix.slot_cur[blockIdx.x] = acur;
        ix.slot_end[blockIdx.x] = aend;
        if (garbage) atomicAdd(&ix.stats[3], garbage);
    }
}


#ifndef WI_WAVES_PER_EU_FAST
// Five waves per SIMD (<= 96 registers, 20 waves per CU: what the LDS holds at 7.7 KB a wave).  The body needs 115 since the
// next partition's records share the current ones' registers (127 before: 18 spilled at 96 and five waves gained nothing);
// at 96 it spills 10 to scratch and the fifth wave is worth it: 29.8 -> 26.3 ms per 50 M reads (5120 resident waves).
#define WI_WAVES_PER_EU_FAST 5
#endif
template <u32 NW, u32 KB, u32 SHIFT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WI_WAVES_PER_EU_FAST, 8))) k_insert_fast(BriskParams P, RecSrc src, const PartDesc* __restrict__ desc,
                                                                                                        u32 n_touched, IndexDev ix, u32* __restrict__ work_counter) {
    static_assert(NW > 0, "compile-time record geometry");
    insert_body<WI_MAX_INST, NW, KB, SHIFT>(P, src, desc, n_touched, ix, work_counter);
}
template <u32 NW, u32 KB, u32 SHIFT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WI_WAVES_PER_EU, 8))) k_insert(BriskParams P, RecSrc src, const PartDesc* __restrict__ desc,
                                               u32 n_touched, IndexDev ix, u32* __restrict__ work_counter) {
    insert_body<WI_MAX_INST, NW, KB, SHIFT>(P, src, desc, n_touched, ix, work_counter);
}
template <u32 NW, u32 KB, u32 SHIFT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) k_insert_big(BriskParams P, RecSrc src, const PartDesc* __restrict__ desc,
                                                                                         u32 n_touched, IndexDev ix, u32* __restrict__ work_counter) {
    insert_body<2 * WI_MAX_INST, NW, KB, SHIFT>(P, src, desc, n_touched, ix, work_counter);
}

// bucket occupancy for partitions wider than 64 buckets (small part_bits): one pass over all entries
__global__ void __launch_bounds__(256) k_bucket_bits(BriskParams P, IndexDev ix, u32 n_parts) {
    const u32 kbits = 2 * P.kb + 6;
    for (u32 part = blockIdx.x; part < n_parts; part += gridDim.x) {
        const u32 cnt = ix.dir[part].cnt;
        const unsigned long long off = ix.dir[part].off;
        for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
            const u128x key = load_key(ix, off + e);
            const u32 bucket = (part << P.shift) | ((u32)shr128(key, kbits).lo & ((1u << P.shift) - 1));
            const u32 bit = 1u << (bucket & 31);
            if (!(ix.bucket_bits[bucket >> 5] & bit)) atomicOr(&ix.bucket_bits[bucket >> 5], bit);
        }
    }
}

// stats(): nb_kmers = sum dir_cnt, largest = max dir_cnt, nb_buckets = popcount(bucket_bits)
__global__ void __launch_bounds__(256) k_stats(const DirEnt* __restrict__ dir, u64 n_parts, const u32* __restrict__ bits, u64 n_words,
                                               unsigned long long* out /* [0] kmers [1] buckets [2] largest */) {
    __shared__ unsigned long long s_a[4], s_b[4], s_c[4];
    unsigned long long a = 0, b = 0, c = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_parts; i += stride) {
        const u32 v = dir[i].cnt;
        a += v;
        c = v > c ? v : c;
    }
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) b += __popc(bits[i]);
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
        const unsigned long long c2 = __shfl_down(c, o, 64);
        c = c2 > c ? c2 : c;
    }
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = a;
        s_b[threadIdx.x >> 6] = b;
        s_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], s_a[0] + s_a[1] + s_a[2] + s_a[3]);
        atomicAdd(&out[1], s_b[0] + s_b[1] + s_b[2] + s_b[3]);
        unsigned long long m = s_c[0];
        for (int i = 1; i < 4; i++) m = s_c[i] > m ? s_c[i] : m;
        atomicMax(&out[2], m);
    }
}


// ===========================================================================
// k_insert_huge: one WORKGROUP of 16 waves per partition, for the partitions one wave must not be left alone with: a hot
// minimizer (satellite repeats, poly-A) puts 10^5..10^6 distinct k-mers into ONE partition, which the wave-per-partition
// kernels work off in chunks of 256 / 512 instances, every chunk streaming the partition's entries through its table --
// entries x instances / 256 probes by 64 lanes while the rest of the device idles.  Here a chunk is 2048 instances in a
// table shared by 1024 lanes: 8 times fewer passes over the entries, each 16 times as wide.  Same semantics as insert_body
// (find-all then insert-missing, DenseMenuYo.hpp:248-310; counts wrap at 256); run-time geometry; classic and binned record
// layouts.  Partitions are chosen by k_need (PartDesc::n_inst > IndexDev::huge_at); the wave kernels skip them.
#define HG_THREADS 1024u
#define HG_INST 2048u          // k-mer instances per chunk
#define HG_TAB 4096u           // table slots
#define HG_LIVE 0x80000000u    // s_cnt: the key owns a table slot (keys allocated by the losers of a CAS race stay dead)
#define HG_EXIST 0x40000000u   // s_cnt: an existing entry has this key
#define HG_MULT 0x3fffffffu
__device__ __forceinline__ const u64* huge_rec(const BriskParams& P, const RecSrc& src, const PartDesc& d, u32 i) {  // record i of the partition (RecSrc)
    if (!src.bin_cap) return src.rec + (u64)(d.r_begin + i) * P.stride;
    return i < src.bin_cap ? src.rec + ((u64)d.part * src.bin_cap + i) * P.stride : src.ovf + ((u64)d.r_begin + i - src.bin_cap) * P.stride;
}
// inclusive prefix sum over the block's 1024 lanes (wave scan + the waves' totals through LDS); *total: the block's sum
__device__ __forceinline__ u32 block_incl_scan(u32 v, u32* s_wsum, u32* total) {
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = __shfl_up(x, o, 64);
        if ((int)lane >= o) x += y;
    }
    __syncthreads();  // s_wsum may still be read from the previous call
    if (lane == 63) s_wsum[wid] = x;
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 i = 0; i < HG_THREADS / 64; i++) {
        const u32 c = s_wsum[i];
        if (i < wid) before += c;
        all += c;
    }
    *total = all;
    return x + before;
}
__global__ void __launch_bounds__(HG_THREADS) k_insert_huge(BriskParams P, RecSrc src, const PartDesc* __restrict__ desc, const u32* __restrict__ huge_list,
                                                            const u32* __restrict__ n_huge, IndexDev ix) {
    __shared__ u64 s_key[2 * HG_INST];
    __shared__ u32 s_tab[HG_TAB];
    __shared__ u32 s_cnt[HG_INST];
    __shared__ u32 s_pref[HG_THREADS + 1];
    __shared__ u32 s_wsum[HG_THREADS / 64];
    __shared__ u32 s_nkeys, s_bm[2], s_fail;
    __shared__ unsigned long long s_noff;
    const u32 tid = threadIdx.x;
    const u32 kbits = 2 * P.kb + 6;
    for (u32 hi = blockIdx.x; hi < *n_huge; hi += gridDim.x) {
        const PartDesc d = desc[huge_list[hi]];
        u32 n_exist = d.n_exist, cap = d.cap, inst_left = d.n_inst;
        unsigned long long off = d.off, garbage = 0;
        if (tid < 2) s_bm[tid] = 0;
        if (tid == 0) s_fail = 0;
        bool failed = false;
        for (u32 rc = 0; rc < d.n_rec && !failed;) {
            // ---- the chunk: the longest run of records (one per lane) with at most HG_INST instances
            const u32 avail = min(d.n_rec - rc, HG_THREADS);
            u32 my_n = 0;
            if (tid < avail) my_n = hdr_n(huge_rec(P, src, d, rc + tid)[P.nw]);
            u32 all;
            const u32 x = block_incl_scan(my_n, s_wsum, &all);
            s_pref[tid + 1] = x;
            if (tid == 0) {
                s_pref[0] = 0;
                s_nkeys = 0;
            }
            for (u32 i = tid; i < HG_TAB; i += HG_THREADS) s_tab[i] = EMPTY_SLOT;
            u32 nrec;
            (void)block_incl_scan(tid < avail && x <= HG_INST ? 1u : 0u, s_wsum, &nrec);  // a prefix: >= 1 (a record has <= 255 instances)
            __syncthreads();
            const u32 ninst = s_pref[nrec];
            // ---- expand and de-duplicate: every distinct key gets a slot in s_key / s_cnt and an entry in the table
            for (u32 i = tid; i < ninst; i += HG_THREADS) {
                u32 lo = 0, hi2 = nrec;  // the record r with s_pref[r] <= i < s_pref[r + 1]
                while (hi2 - lo > 1) {
                    const u32 mid = (lo + hi2) >> 1;
                    if (s_pref[mid] <= i) lo = mid; else hi2 = mid;
                }
                const u64* c = huge_rec(P, src, d, rc + lo);
                u64 w[5];
                for (u32 q = 0; q < 5; q++) w[q] = q <= P.nw ? c[q] : 0;
                const u64 hdr = w[P.nw];
                const u32 j = i - s_pref[lo];
                const u128x key = make_key(P, hdr_bucket(hdr), record_kmer(P, w, hdr_n(hdr), j), hdr_idx0(hdr) + j);
                const u32 mult = (hdr & HDR_HAS_MULT) ? (u32)(hdr >> 48) & 0xffu : 1u;
                u32 h = hash_key32(key) & (HG_TAB - 1), mine = EMPTY_SLOT;
                for (;;) {
                    u32 v = s_tab[h];
                    if (v == EMPTY_SLOT) {
                        if (mine == EMPTY_SLOT) {
                            mine = atomicAdd(&s_nkeys, 1u);
                            s_key[2 * mine] = key.lo;
                            s_key[2 * mine + 1] = key.hi;
                            s_cnt[mine] = 0;
                            __threadfence_block();  // the key is in place before the table points at it
                        }
                        v = atomicCAS(&s_tab[h], EMPTY_SLOT, mine);
                        if (v == EMPTY_SLOT) {
                            atomicAdd(&s_cnt[mine], mult | HG_LIVE);
                            break;
                        }
                    }
                    __threadfence_block();
                    if (s_key[2 * v] == key.lo && s_key[2 * v + 1] == key.hi) {
                        atomicAdd(&s_cnt[v], mult);
                        break;
                    }
                    h = (h + 1) & (HG_TAB - 1);
                }
            }
            __syncthreads();
            const u32 nkeys = s_nkeys;
            // ---- the partition's entries: the ones that are in the chunk take its multiplicity (mod 256)
            for (u32 e = tid; e < n_exist; e += HG_THREADS) {
                const u128x ke = load_key(ix, off + e);
                const u64 klo = ke.lo, khi = ke.hi;
                u32 h = hash_key32(mk128(klo, khi)) & (HG_TAB - 1);
                for (;;) {
                    const u32 v = s_tab[h];
                    if (v == EMPTY_SLOT) break;
                    if (s_key[2 * v] == klo && s_key[2 * v + 1] == khi) {
                        const u32 m = atomicOr(&s_cnt[v], HG_EXIST) & HG_MULT;
                        ix.counts[off + e] = (uint8_t)(ix.counts[off + e] + m);
                        break;
                    }
                    h = (h + 1) & (HG_TAB - 1);
                }
            }
            __syncthreads();
            // ---- the new ones: counted, given room (the partition moves at most once per batch), appended
            u32 mine_new = 0;
            for (u32 q = tid; q < nkeys; q += HG_THREADS) {
                const u32 cv = s_cnt[q];
                mine_new += (cv & HG_LIVE) && !(cv & HG_EXIST) ? 1u : 0u;
            }
            u32 n_new;
            (void)block_incl_scan(mine_new, s_wsum, &n_new);
            inst_left -= ninst;
            if (n_exist + n_new > cap) {
                const unsigned long long want = grow_cap(n_exist + n_new + inst_left);
                if (tid == 0) {
                    const unsigned long long got = atomicAdd(ix.cursor, want);
                    s_noff = got;
                    if (got + want > ix.arena_cap) {  // must not happen (the host reserves the bound): drop, flag
                        atomicOr(ix.err, 2u);
                        s_fail = 1;
                    }
                }
                __syncthreads();
                if (s_fail) {
                    failed = true;
                    break;
                }
                const unsigned long long noff = s_noff;
                for (u32 e = tid; e < n_exist; e += HG_THREADS) {
                    move_key(ix, noff + e, off + e);
                    ix.counts[noff + e] = ix.counts[off + e];
                }
                garbage += cap;
                cap = (u32)want;
                off = noff;
            }
            u32 done = 0;  // new entries of the rounds before this one
            for (u32 q0 = 0; q0 < nkeys; q0 += HG_THREADS) {
                const u32 q = q0 + tid;
                const u32 cv = q < nkeys ? s_cnt[q] : 0;
                const bool is_new = (cv & HG_LIVE) && !(cv & HG_EXIST);
                u32 round;
                const u32 pos = block_incl_scan(is_new ? 1u : 0u, s_wsum, &round);
                if (is_new) {
                    const unsigned long long at = off + n_exist + done + pos - 1;
                    const u64 klo = s_key[2 * q], khi = s_key[2 * q + 1];
                    store_key(ix, at, klo, khi);
                    ix.counts[at] = (uint8_t)(cv & HG_MULT);
                    // bucket id inside the partition: the key's top `shift` bits (<= 6 of them used here), as insert_body has it
                    const u32 bl = P.shift ? ((u32)shr128(mk128(klo, khi), kbits).lo & ((1u << P.shift) - 1)) : 0;
                    const u32 bb = P.shift > 6 ? (bl >> (P.shift - 6)) : bl;
                    atomicOr(&s_bm[bb >> 5], 1u << (bb & 31));
                }
                done += round;
            }
            n_exist += n_new;
            rc += nrec;
            __syncthreads();  // the table and the key store are reused by the next chunk
        }
        __syncthreads();
        if (tid == 0 && !failed) {
            ix.dir[d.part] = DirEnt{off, n_exist, cap};
            if (garbage) atomicAdd(&ix.stats[3], garbage);
        }
        if (P.shift <= 6 && tid < 2 && !failed) {  // bucket occupancy bits, as in insert_body's epilogue
            const u32 mask = s_bm[tid];
            const u32 nb = 1u << P.shift;
            const u64 first = ((u64)d.part << P.shift) >> P.ext_bits;
            if (mask) {
                if (nb >= 32) {
                    if (tid * 32 < nb) atomicOr(&ix.bucket_bits[(first >> 5) + tid], mask);
                } else if (tid == 0) {
                    atomicOr(&ix.bucket_bits[first >> 5], mask << (first & 31));
                }
            }
        }
        __syncthreads();
    }
}
