// brisk_readout.hip -- reading the index: k_query, checksum, enumeration, look-ups, and the per-call upsert of the facade.
// Included by brisk_kernels.hip (one translation unit).
// ===========================================================================
// k_query: k_insert's structure (one wave per partition, persistent waves, descriptors), but the
// table keeps every k-mer instance in its own slot (equal keys sit behind each other in the probe
// chain), the partition's entries stream through it, and every hit adds the entry's count to the
// instance's record; a record's total goes to its read with one atomic
// (get_superkmer, Brisk.hpp:102-118; summed per read as counter.cpp:296-301 does).
__global__ void __launch_bounds__(64) k_query(BriskParams P, const u64* __restrict__ rec, const u32* __restrict__ tags,
                                              const PartDesc* __restrict__ desc, u32 n_touched, IndexDev ix,
                                              unsigned long long* __restrict__ per_read_sum, u32* __restrict__ work_counter) {
    __shared__ u64 s_key[2 * WI_MAX_INST];
    __shared__ u64 s_rec[WI_MAX_REC * 5];
    __shared__ u32 s_tab[WI_TABLE];
    __shared__ u32 s_pref[WI_MAX_REC + 1];
    __shared__ u32 s_rsum[WI_MAX_REC];
    __shared__ uint8_t s_irec[WI_MAX_INST];

    const u32 lane = threadIdx.x;
    for (;;) {
        u32 t0 = 0;
        if (lane == 0) t0 = atomicAdd(work_counter, WI_BATCH);
        t0 = __shfl(t0, 0, 64);
        if (t0 >= n_touched) break;
        const u32 t_end = min(t0 + WI_BATCH, n_touched);
        for (u32 t = t0; t < t_end; t++) {
            const PartDesc d = desc[t];
            if (d.n_exist == 0) continue;  // nothing to find in an empty partition
            if (d.n_exist & PART_HUGE) continue;  // k_query_huge takes it
            const u32 r_end = d.r_begin + d.n_rec;
            for (u32 rc = d.r_begin; rc < r_end;) {
                const u32 avail = min(r_end - rc, (u32)WI_MAX_REC);
                const RecRegs rr = load_rec_regs(P, rec, rc, avail, lane);
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
                const u32 x0 = wave_incl_scan(raw_n, lane);
                const u32 nrec = (u32)__popcll(__ballot(lane < avail && x0 <= WI_MAX_INST));  // >= 1; a prefix
                const u32 my_n = lane < nrec ? raw_n : 0;
                const u32 ninst = __shfl(x0, nrec - 1, 64);
                s_pref[lane + 1] = x0;
                if (lane == 0) s_pref[0] = 0;
                s_rsum[lane] = 0;
#pragma unroll
                for (u32 w = 0; w < WI_TS; w++) s_tab[w * 64 + lane] = EMPTY_SLOT;
                {
                    const u32 start = x0 - raw_n;
                    for (u32 j = 0; j < my_n; j++) s_irec[start + j] = (uint8_t)lane;
                }
                wave_sync();
                // expand and give every instance a slot of its own
                for (u32 i = lane; i < ninst; i += 64) {
                    const u32 r = s_irec[i];
                    const u64* c = s_rec + r * P.stride;
                    const u64 hdr = c[P.nw];
                    const u32 j = i - s_pref[r];
                    const u128x key = make_key(P, hdr_bucket(hdr), record_kmer_lds<0>(P, c, hdr_n(hdr), j), hdr_idx0(hdr) + j);
                    s_key[2 * i] = key.lo;
                    s_key[2 * i + 1] = key.hi;
                    u32 h = hash_key32(key) & (WI_TABLE - 1);
                    while (atomicCAS(&s_tab[h], EMPTY_SLOT, i) != EMPTY_SLOT) h = (h + 1) & (WI_TABLE - 1);
                }
                wave_sync();
                // the partition's entries look their key up; every instance holding it gets the count
                for (u32 e = lane; e < d.n_exist; e += 64) {
                    const u128x key = load_key(ix, d.off + e);
                    const u32 cnt = ix.counts[d.off + e];
                    u32 h = hash_key32(key) & (WI_TABLE - 1);
                    for (;;) {
                        const u32 v = s_tab[h];
                        if (v == EMPTY_SLOT) break;
                        if (s_key[2 * v] == key.lo && s_key[2 * v + 1] == key.hi) atomicAdd(&s_rsum[s_irec[v]], cnt);
                        h = (h + 1) & (WI_TABLE - 1);
                    }
                }
                wave_sync();
                if (lane < nrec) {
                    const u32 sum = s_rsum[lane];
                    if (sum) atomicAdd(&per_read_sum[tags[rc + lane]], (unsigned long long)sum);
                }
                rc += nrec;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_query_fast: the query with the roles turned round and the record geometry as constants (the parameter sets k_insert_fast
// has).  The table holds a chunk of the partition's ENTRIES -- distinct keys, so a probe chain ends at the first match and
// building it needs no comparison -- and the k-mer instances of the partition's records, cut out of 32-bit record words as in
// expand_and_dedupe_words, probe it straight from registers: no per-instance state in LDS, no chains of equal keys (k_query
// parks every instance in a slot of its own, 3.4 equal keys behind each other at 15x coverage, and walks all of them for
// every entry).  Same results: per record the sum of the counts of its k-mers that are present, one atomic per record into
// its read's sum.
// ENT: entries per table chunk.  The kernel needs 46 registers, so what decides how many waves are resident -- and with them,
// as in k_insert, the time -- is its LDS: 256 entries, 64 records and 768 instances per chunk take 9.7 KB (4 waves per SIMD),
// 128 / 32 / 384 take 4.9 KB (8 per SIMD).  The host picks 128 while the index averages at most 100 entries per partition.
template <u32 NW, u32 KB, u32 SHIFT>
__device__ __forceinline__ void inst_key_words(const u32* s_rw, u32 r, u32 i, u64* lo_out, u64* hi_out) {
    constexpr u32 RS = RecGeom<NW>::RS, INFO = RecGeom<NW>::INFO, KBITS = 2 * KB + 6;
    const u32* base = s_rw + r * RS;
    const u32 info = base[INFO];
    const u32 j = i - (info & 0x3ffu);
    const u32 n = (info >> 10) & 0xffu;
    const u32 s = 2 * (n - 1 - j);
    const u32* wp = base + (s >> 5);
    const u32 sh = s & 31;
    const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4];
    u32 k0 = __builtin_amdgcn_alignbit(w1, w0, sh), k1 = __builtin_amdgcn_alignbit(w2, w1, sh);
    u32 k2 = __builtin_amdgcn_alignbit(w3, w2, sh), k3 = __builtin_amdgcn_alignbit(w4, w3, sh);
    k0 = (k0 & ~0x3fu) | (((info >> 18) & 0xffu) + j);  // idx' = idx0' + j (SuperKmerLight.hpp:98)
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
    *lo_out = lo;
    *hi_out = hi;
}
template <u32 NW, u32 KB, u32 SHIFT, u32 ENT>
__global__ void __launch_bounds__(64) k_query_fast(BriskParams PP, RecSrc src, const u32* __restrict__ tags_binned, const u32* __restrict__ tags,
                                                   const PartDesc* __restrict__ desc, u32 n_touched, IndexDev ix, unsigned long long* __restrict__ per_read_sum,
                                                   u32* __restrict__ work_counter) {
    // records and their reads' indices: classic layout (src.bin_cap == 0) src.rec / tags in partition order; binned: record i of a
    // partition in its bin (tags_binned alongside) for i < bin_cap, beyond it among the overflow records src.ovf / tags (RecSrc)
    constexpr u32 RS = RecGeom<NW>::RS;
    // (QF_MAX_INST: whole 32-bit words of instance marks per lane; the info word's prefix field has 10 bits.  The table is a
    // quarter full at most with 128-entry chunks: a probe round costs the whole wave ~25 vector instructions whichever lane needs it)
    constexpr u32 QF_ENT = ENT, QF_TABLE = 512, QF_REC = ENT / 4, QF_MAX_INST = ENT == 128 ? 256 : 768, IW = QF_MAX_INST / 256;
    BriskParams P = PP;
    P.nw = NW;
    P.stride = NW + 1;
    __shared__ u64 s_ekey[2 * QF_ENT];
    __shared__ u32 s_tab[QF_TABLE];
    __shared__ u32 s_rw[QF_REC * RS + 4];
    __shared__ u32 s_rsum[QF_REC];
    __shared__ __attribute__((aligned(4))) uint8_t s_irec[QF_MAX_INST];
    u32* irec32 = (u32*)s_irec;
    const u32 lane = threadIdx.x;
    for (;;) {
        u32 t0 = 0;
        if (lane == 0) t0 = atomicAdd(work_counter, WI_BATCH);
        t0 = (u32)__builtin_amdgcn_readfirstlane((int)t0);
        if (t0 >= n_touched) break;
        const u32 t_n = min((u32)WI_BATCH, n_touched - t0);
        for (u32 ti = 0; ti < t_n; ti++) {
            const PartDesc d = desc[t0 + ti];  // wave-uniform index, read-only array: scalar loads (as in insert_body)
            if (d.n_exist == 0) continue;  // nothing to find in an empty partition
            if (d.n_exist & PART_HUGE) continue;  // k_query_huge takes it
            const u32 r_end = d.r_begin + d.n_rec;
            for (u32 ec = 0; ec < d.n_exist; ec += QF_ENT) {
                const u32 ne = min(d.n_exist - ec, QF_ENT);
                wave_sync();  // the previous chunk's probes are done
#pragma unroll
                for (u32 w = 0; w < QF_TABLE / 64; w++) s_tab[w * 64 + lane] = EMPTY_SLOT;
                wave_sync();
                for (u32 e = lane; e < ne; e += 64) {
                    const u128x kv = load_key<(2 * KB + 6 + SHIFT <= 64 ? 1u : 2u)>(ix, d.off + ec + e);
                    s_ekey[2 * e] = kv.lo;
                    s_ekey[2 * e + 1] = kv.hi;
                    const u32 word = e | ((u32)ix.counts[d.off + ec + e] << 16);  // table word: entry of the chunk | its count
                    u32 h = hash_key32(kv) & (QF_TABLE - 1);
                    while (atomicCAS(&s_tab[h], EMPTY_SLOT, word) != EMPTY_SLOT) h = (h + 1) & (QF_TABLE - 1);
                }
                for (u32 rc = d.r_begin; rc < r_end;) {
                    const u32 avail = min(r_end - rc, QF_REC);
                    const RecRegs rr = load_part_recs(P, src, d.part, d.r_begin, rc, avail, lane);
                    const u64 my_hdr = NW == 1 ? rr.w1 : NW == 2 ? rr.w2 : NW == 3 ? rr.w3 : rr.w4;
                    const u32 raw_n = lane < avail ? hdr_n(my_hdr) : 0;
                    const u32 x0 = wave_incl_scan(raw_n, lane);
                    const u32 nrec = (u32)__popcll(__ballot(lane < avail && x0 <= QF_MAX_INST));  // >= 1; a prefix
                    const u32 ninst = (u32)__builtin_amdgcn_readlane((int)x0, (int)nrec - 1);
                    wave_sync();  // the previous record chunk's sums have been read
                    if (lane < QF_REC) s_rsum[lane] = 0;
#pragma unroll
                    for (u32 q = 0; q < IW; q++) irec32[q * 64 + lane] = 0;
                    wave_sync();
                    if (lane < nrec) {
                        const u32 start = x0 - raw_n;
                        const u32 info = start | (raw_n << 10) | (hdr_idx0(my_hdr) << 18) | ((hdr_bucket(my_hdr) & ((1u << SHIFT) - 1)) << 26);
                        store_rec_words<NW>(s_rw + lane * RS, rr, info);
                        if (raw_n) s_irec[start] = (uint8_t)(lane + 1);  // instance -> record: a mark on every record's first instance ...
                    }
                    wave_sync();
                    {  // ... and a running maximum spreads the marks (records lie in lane order); a lane owns IW * 4 consecutive instances here
                        u32 wv[IW], run = 0;
#pragma unroll
                        for (u32 q = 0; q < IW; q++) {
                            wv[q] = irec32[lane * IW + q];
                            run = op_max_u32(run, op_max_u32(op_max_u32(wv[q] & 0xff, (wv[q] >> 8) & 0xff), op_max_u32((wv[q] >> 16) & 0xff, wv[q] >> 24)));
                        }
                        u32 carry = wave_prev_lane(wave_incl_max_scan(run));
#pragma unroll
                        for (u32 q = 0; q < IW; q++) {
                            const u32 b0 = op_max_u32(carry, wv[q] & 0xff), b1 = op_max_u32(b0, (wv[q] >> 8) & 0xff);
                            const u32 b2 = op_max_u32(b1, (wv[q] >> 16) & 0xff), b3 = op_max_u32(b2, wv[q] >> 24);
                            carry = b3;
                            irec32[lane * IW + q] = ((b0 - 1) & 0xff) | (((b1 - 1) & 0xff) << 8) | (((b2 - 1) & 0xff) << 16) | ((b3 - 1) << 24);
                        }
                    }
                    wave_sync();
                    for (u32 i0 = 0; i0 < ninst;) {
                        if (ninst - i0 > 64) {  // two instances per lane in flight
                            const u32 ia = i0 + lane, ib = i0 + 64 + lane;
                            const bool vb = ib < ninst;
                            const u32 ra = s_irec[ia], rb = s_irec[vb ? ib : ia];
                            u64 alo, ahi, blo, bhi;
                            inst_key_words<NW, KB, SHIFT>(s_rw, ra, ia, &alo, &ahi);
                            inst_key_words<NW, KB, SHIFT>(s_rw, rb, vb ? ib : ia, &blo, &bhi);
                            u32 ha = hash_key32(mk128(alo, ahi)) & (QF_TABLE - 1), hb = hash_key32(mk128(blo, bhi)) & (QF_TABLE - 1);
                            bool pa = true, pb = vb;
                            u32 fa = 0, fb = 0;
                            while (__any(pa || pb)) {
                                const u32 ea = pa ? s_tab[ha] : EMPTY_SLOT, eb = pb ? s_tab[hb] : EMPTY_SLOT;
                                const u32 xa = ea == EMPTY_SLOT ? 0 : (ea & 0xffffu), xb = eb == EMPTY_SLOT ? 0 : (eb & 0xffffu);
                                const u64 qa0 = s_ekey[2 * xa], qa1 = s_ekey[2 * xa + 1], qb0 = s_ekey[2 * xb], qb1 = s_ekey[2 * xb + 1];
                                if (pa) {
                                    if (ea == EMPTY_SLOT) pa = false;
                                    else if (qa0 == alo && qa1 == ahi) {
                                        fa = ea >> 16;
                                        pa = false;
                                    } else ha = (ha + 1) & (QF_TABLE - 1);
                                }
                                if (pb) {
                                    if (eb == EMPTY_SLOT) pb = false;
                                    else if (qb0 == blo && qb1 == bhi) {
                                        fb = eb >> 16;
                                        pb = false;
                                    } else hb = (hb + 1) & (QF_TABLE - 1);
                                }
                            }
                            if (fa) atomicAdd(&s_rsum[ra], fa);
                            if (fb) atomicAdd(&s_rsum[rb], fb);
                            i0 += 128;
                        } else {  // the last <= 64
                            const u32 ia = i0 + lane;
                            const bool va = ia < ninst;
                            const u32 ra = s_irec[va ? ia : i0];
                            u64 alo, ahi;
                            inst_key_words<NW, KB, SHIFT>(s_rw, ra, va ? ia : i0, &alo, &ahi);
                            u32 ha = hash_key32(mk128(alo, ahi)) & (QF_TABLE - 1);
                            bool pa = va;
                            u32 fa = 0;
                            while (__any(pa)) {
                                const u32 ea = pa ? s_tab[ha] : EMPTY_SLOT;
                                const u32 xa = ea == EMPTY_SLOT ? 0 : (ea & 0xffffu);
                                const u64 qa0 = s_ekey[2 * xa], qa1 = s_ekey[2 * xa + 1];
                                if (pa) {
                                    if (ea == EMPTY_SLOT) pa = false;
                                    else if (qa0 == alo && qa1 == ahi) {
                                        fa = ea >> 16;
                                        pa = false;
                                    } else ha = (ha + 1) & (QF_TABLE - 1);
                                }
                            }
                            if (fa) atomicAdd(&s_rsum[ra], fa);
                            i0 += 64;
                        }
                    }
                    wave_sync();
                    if (lane < nrec) {
                        const u32 sum = s_rsum[lane];
                        if (sum) {
                            u32 tag;
                            if (!src.bin_cap) tag = tags[rc + lane];
                            else {
                                const u32 i = rc - d.r_begin + lane;
                                tag = i < src.bin_cap ? tags_binned[(u64)d.part * src.bin_cap + i] : tags[d.r_begin + i - src.bin_cap];
                            }
                            atomicAdd(&per_read_sum[tag], (unsigned long long)sum);
                        }
                    }
                    rc += nrec;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_query_huge: a workgroup of 16 waves per partition of very many entries (a hot minimizer, see k_insert_huge): the wave
// kernels walk all of a partition's records once per table chunk of 128 / 256 entries -- entries / 256 x instances probes by
// one wave.  Here the table holds 2048 entries and 1024 lanes probe it.  Run-time geometry; classic and binned records.
#define HQ_LIST_CAP 65536u   // listed partitions per batch (the rest stay with the wave kernels)
#define HQ_ENT 2048u
#define HQ_TAB 4096u
__global__ void __launch_bounds__(HG_THREADS) k_query_huge(BriskParams P, RecSrc src, const u32* __restrict__ tags_binned, const u32* __restrict__ tags,
                                                           const PartDesc* __restrict__ desc, const u32* __restrict__ huge_list, const u32* __restrict__ n_huge, IndexDev ix,
                                                           unsigned long long* __restrict__ per_read_sum) {
    __shared__ u64 s_ekey[2 * HQ_ENT];
    __shared__ u32 s_tab[HQ_TAB];
    __shared__ u32 s_pref[HG_THREADS + 1];
    __shared__ u32 s_rsum[HG_THREADS];
    __shared__ u32 s_wsum[HG_THREADS / 64];
    const u32 tid = threadIdx.x;
    for (u32 hi = blockIdx.x; hi < min(*n_huge, (u32)HQ_LIST_CAP); hi += gridDim.x) {
        PartDesc d = desc[huge_list[hi]];
        d.n_exist &= ~PART_HUGE;
        for (u32 ec = 0; ec < d.n_exist; ec += HQ_ENT) {
            const u32 ne = min(d.n_exist - ec, HQ_ENT);
            __syncthreads();  // the previous chunk's probes are done
            for (u32 i = tid; i < HQ_TAB; i += HG_THREADS) s_tab[i] = EMPTY_SLOT;
            __syncthreads();
            for (u32 e = tid; e < ne; e += HG_THREADS) {
                const u128x ke = load_key(ix, d.off + ec + e);
                const u64 klo = ke.lo, khi = ke.hi;
                s_ekey[2 * e] = klo;
                s_ekey[2 * e + 1] = khi;
                const u32 word = e | ((u32)ix.counts[d.off + ec + e] << 16);  // entry of the chunk | its count
                u32 h = hash_key32(mk128(klo, khi)) & (HQ_TAB - 1);
                while (atomicCAS(&s_tab[h], EMPTY_SLOT, word) != EMPTY_SLOT) h = (h + 1) & (HQ_TAB - 1);
            }
            for (u32 rc = 0; rc < d.n_rec; rc += HG_THREADS) {  // the partition's records, one per lane; their instances spread over the lanes
                const u32 avail = min(d.n_rec - rc, HG_THREADS);
                const u32 my_n = tid < avail ? hdr_n(huge_rec(P, src, d, rc + tid)[P.nw]) : 0;
                u32 ninst;
                const u32 x = block_incl_scan(my_n, s_wsum, &ninst);
                s_pref[tid + 1] = x;
                if (tid == 0) s_pref[0] = 0;
                s_rsum[tid] = 0;
                __syncthreads();  // also: the table is complete
                for (u32 i = tid; i < ninst; i += HG_THREADS) {
                    u32 lo = 0, hi2 = avail;  // the record r with s_pref[r] <= i < s_pref[r + 1]
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
                    u32 h = hash_key32(key) & (HQ_TAB - 1);
                    for (;;) {
                        const u32 v = s_tab[h];
                        if (v == EMPTY_SLOT) break;
                        const u32 e = v & 0xffffu;
                        if (s_ekey[2 * e] == key.lo && s_ekey[2 * e + 1] == key.hi) {
                            if (v >> 16) atomicAdd(&s_rsum[lo], v >> 16);
                            break;
                        }
                        h = (h + 1) & (HQ_TAB - 1);
                    }
                }
                __syncthreads();
                if (tid < avail && s_rsum[tid]) {
                    const u32 i = rc + tid;
                    u32 tag;
                    if (!src.bin_cap) tag = tags[d.r_begin + i];
                    else tag = i < src.bin_cap ? tags_binned[(u64)d.part * src.bin_cap + i] : tags[d.r_begin + i - src.bin_cap];
                    atomicAdd(&per_read_sum[tag], (unsigned long long)s_rsum[tid]);
                }
            }
        }
    }
}

// ===========================================================================
// k_enumerate: entries of partitions [p_begin, p_end) in order; out_base[p - p_begin]
// is the exclusive prefix of dir_cnt over that range (Brisk::next yields unhashed k-mers).
// order-independent digest of every entry (brisk_hip_checksum)
__device__ __forceinline__ u64 fmix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_checksum(BriskParams P, IndexDev ix, u32 n_parts, unsigned long long* out) {
    unsigned long long na = 0, sa = 0, da = 0;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6, lane = threadIdx.x & 63;
    for (u32 part = wave; part < n_parts; part += n_waves) {
        const DirEnt de = ix.dir[part];
        for (u32 e = lane; e < de.cnt; e += 64) {
            const u128x key = load_key(ix, de.off + e);
            u32 idx;
            u128x hk = entry_hashed_kmer(P, part, key, &idx);
            const u64 mm = mix2m_inv(shr128(hk, 2 * idx).lo & P.m_mask, P.m_mask);
            hk = or128(andn128(hk, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(mm, 0), 2 * idx));
            const u32 cnt = ix.counts[de.off + e];
            na += 1;
            sa += cnt;
            da += fmix(hk.lo ^ fmix(hk.hi ^ fmix(((u64)idx << 8) | cnt)));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        na += __shfl_xor(na, o, 64);
        sa += __shfl_xor(sa, o, 64);
        da += __shfl_xor(da, o, 64);
    }
    if (lane == 0 && na) {
        atomicAdd(&out[0], na);
        atomicAdd(&out[1], sa);
        atomicAdd(&out[2], da);
    }
}

__global__ void __launch_bounds__(256) k_dir_counts(const DirEnt* __restrict__ dir, u64 n, u32* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dir[i].cnt;
}

__global__ void __launch_bounds__(64) k_enumerate(BriskParams P, IndexDev ix, u32 p_begin, u32 n_parts, const u64* __restrict__ out_base,
                                                  u64* __restrict__ out_lo, u64* __restrict__ out_hi, uint8_t* __restrict__ out_idx,
                                                  uint8_t* __restrict__ out_cnt, u32* __restrict__ out_id) {
    for (u32 pi = blockIdx.x; pi < n_parts; pi += gridDim.x) {  // (a chunk of a sparse index spans up to 2^27 partitions: more than a grid of 64-lane blocks may have)
        const u32 part = p_begin + pi;
        const u32 cnt = ix.dir[part].cnt;
        if (!cnt) continue;
        const unsigned long long off = ix.dir[part].off;
        const u64 ob = out_base[pi];
        for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
            const u128x key = load_key(ix, off + e);
            u32 idx;
            u128x hk = entry_hashed_kmer(P, part, key, &idx);
            // unhash_kmer_minimizer (Kmers.cpp:178-187)
            const u64 hm = shr128(hk, 2 * idx).lo & P.m_mask;
            const u64 mm = mix2m_inv(hm, P.m_mask);
            hk = or128(andn128(hk, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(mm, 0), 2 * idx));
            out_lo[ob + e] = hk.lo;
            out_hi[ob + e] = hk.hi;
            out_idx[ob + e] = (uint8_t)idx;
            out_cnt[ob + e] = ix.counts[off + e];
            if (out_id) out_id[ob + e] = ix.ids[off + e];
        }
    }
}

// k_lookup: one wave per query (Brisk::get: hash the minimizer, find the bucket, compare compacted k-mers)
__global__ void __launch_bounds__(256) k_lookup(BriskParams P, IndexDev ix, const u64* __restrict__ q_lo, const u64* __restrict__ q_hi,
                                                const uint8_t* __restrict__ q_idx, u64 n, uint8_t* __restrict__ out_data,
                                                uint8_t* __restrict__ out_found, u32* __restrict__ out_id) {
    const u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63;
    if (qi >= n) return;
    const u32 idx = q_idx[qi];
    u128x km = mk128(q_lo[qi], q_hi[qi]);
    bool found = false;
    u32 data = 0, id = 0;
    if (idx <= P.w) {
        const u64 mm = shr128(km, 2 * idx).lo & P.m_mask;
        const u64 h = mix2m(mm, P.m_mask);
        const u32 bucket = routing_id(P, h, idx);
        km = or128(andn128(km, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(h, 0), 2 * idx));
        const u32 cut = idx + P.suff_reduc;
        const u128x lowm = mask128(2 * cut);
        const u128x comp = or128(andn128(shr128(km, 2 * P.b), lowm), and128(km, lowm));
        const u128x key = make_key(P, bucket, and128(comp, mask128(2 * P.kb)), cut);
        const u32 part = bucket >> P.shift;
        const u32 cnt = ix.dir[part].cnt;
        const unsigned long long off = ix.dir[part].off;
        for (u32 e = lane; e < cnt && !found; e += 64) {
            if (eq128(load_key(ix, off + e), key)) {
                found = true;
                data = ix.counts[off + e];
                if (out_id) id = ix.ids[off + e];
            }
        }
    }
    const unsigned long long bal = __ballot(found);
    if (bal) {
        const int src = __ffsll((long long)bal) - 1;
        data = __shfl(data, src, 64);
        id = __shfl(id, src, 64);
    }
    if (lane == 0) {
        out_found[qi] = bal ? 1 : 0;
        out_data[qi] = (uint8_t)data;
        if (out_id) out_id[qi] = bal ? id : 0xffffffffu;
    }
}

// ---- the per-call API of the facade (Brisk::insert_superkmer, Brisk.hpp:123-147) ----
// entry key, partition and bucket of an UNHASHED (kmer_s, minimizer_idx): hash the minimizer
// (Kmers.cpp:191-200), pick the bucket (Brisk.hpp:135-137), drop its nts (Kmers.cpp:138-145)
__device__ __forceinline__ u128x key_of_kmer(const BriskParams& P, u128x km, u32 idx, u32* part, u32* bucket_out) {
    const u64 mm = shr128(km, 2 * idx).lo & P.m_mask;
    const u64 h = mix2m(mm, P.m_mask);
    const u32 bucket = routing_id(P, h, idx);
    km = or128(andn128(km, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(h, 0), 2 * idx));
    const u32 cut = idx + P.suff_reduc;
    const u128x lowm = mask128(2 * cut);
    const u128x comp = or128(andn128(shr128(km, 2 * P.b), lowm), and128(km, lowm));
    *part = bucket >> P.shift;
    *bucket_out = bucket >> P.ext_bits;  // the bucket id proper (for the occupancy bitmap)
    return make_key(P, bucket, and128(comp, mask128(2 * P.kb)), cut);
}

// find-all then insert-missing for the k-mers of ONE vector, in order (DenseMenuYo.hpp:248-310).
// One wave; every k-mer scans its partition with 64 lanes.  Entry-id mode: a new entry takes
// the next dense id; DATA lives with the caller, indexed by id.  Stops (and reports how many
// k-mers it handled) when the arena cannot hold a move; the host grows it and calls again.
__global__ void __launch_bounds__(64) k_upsert(BriskParams P, IndexDev ix, const u64* __restrict__ q_lo, const u64* __restrict__ q_hi,
                                               const uint8_t* __restrict__ q_idx, u32 n, u32* __restrict__ out_id, uint8_t* __restrict__ out_new,
                                               unsigned long long* __restrict__ id_counter, u32* __restrict__ n_done,
                                               unsigned long long* __restrict__ cursor_out = nullptr) {
    const u32 lane = threadIdx.x;
    u32 done = 0;
    for (u32 qi = 0; qi < n; qi++) {
        const u32 idx = q_idx[qi];
        u32 part, bucket;
        const u128x key = key_of_kmer(P, mk128(q_lo[qi], q_hi[qi]), idx <= P.w ? idx : 0, &part, &bucket);
        DirEnt de = ix.dir[part];
        bool found = false;
        u32 id = 0;
        for (u32 e = lane; e < de.cnt && !found; e += 64) {
            if (eq128(load_key(ix, de.off + e), key)) {
                found = true;
                id = ix.ids[de.off + e];
            }
        }
        const unsigned long long bal = __ballot(found);
        if (bal) {
            id = __shfl(id, __ffsll((long long)bal) - 1, 64);
            if (lane == 0) {
                out_id[qi] = id;
                out_new[qi] = 0;
            }
        } else {
            if (de.cnt == de.cap) {  // move the partition to a larger slice
                const u32 want = grow_cap(de.cnt + 1);
                unsigned long long noff = 0;
                if (lane == 0) noff = atomicAdd(ix.cursor, (unsigned long long)want);
                noff = __shfl(noff, 0, 64);
                if (noff + want > ix.arena_cap) {  // host must grow the arena; nothing was changed for this k-mer
                    if (lane == 0) atomicAdd(ix.cursor, (unsigned long long)(0ull - want));
                    break;
                }
                for (u32 e = lane; e < de.cnt; e += 64) {
                    move_key(ix, noff + e, de.off + e);
                    ix.counts[noff + e] = ix.counts[de.off + e];
                    ix.ids[noff + e] = ix.ids[de.off + e];
                }
                if (lane == 0) atomicAdd(&ix.stats[3], (unsigned long long)de.cap);
                de.off = noff;
                de.cap = want;
            }
            if (lane == 0) {
                const u32 nid = (u32)atomicAdd(id_counter, 1ull);
                const unsigned long long at = de.off + de.cnt;
                store_key(ix, at, key.lo, key.hi);
                ix.counts[at] = 0;
                ix.ids[at] = nid;
                ix.dir[part] = DirEnt{de.off, de.cnt + 1, de.cap};
                if (!(ix.bucket_bits[bucket >> 5] >> (bucket & 31) & 1u)) atomicOr(&ix.bucket_bits[bucket >> 5], 1u << (bucket & 31));
                out_id[qi] = nid;
                out_new[qi] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the next k-mer of the vector reads this partition again (same wave)
        }
        done = qi + 1;
    }
    if (lane == 0) {
        *n_done = done;
        if (cursor_out) *cursor_out = *ix.cursor;  // (this wave's own atomics on it are done: same lane, program order)
    }
}

// records -> the k-mers of each vector, unhashed (what SuperKmerEnumerator::next hands out):
// one wave per record, one lane per k-mer; out row r holds up to `row` k-mers
__global__ void __launch_bounds__(64) k_expand_records(BriskParams P, const u64* __restrict__ rec, u32 n_rec, u32 row, u64* __restrict__ out_lo,
                                                       u64* __restrict__ out_hi, uint8_t* __restrict__ out_idx) {
    const u32 r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_rec) return;
    const u64* c = rec + (u64)r * P.stride;
    const u64 hdr = c[P.nw];
    const u32 n = hdr_n(hdr);
    if (lane >= n) return;
    const u32 bucket = hdr_bucket(hdr);
    const u128x key = make_key(P, bucket, record_kmer(P, c, n, lane), hdr_idx0(hdr) + lane);
    u32 idx;
    u128x hk = entry_hashed_kmer(P, bucket >> P.shift, key, &idx);
    const u64 hm = shr128(hk, 2 * idx).lo & P.m_mask;
    const u64 mm = mix2m_inv(hm, P.m_mask);
    hk = or128(andn128(hk, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(mm, 0), 2 * idx));
    out_lo[(u64)r * row + lane] = hk.lo;
    out_hi[(u64)r * row + lane] = hk.hi;
    out_idx[(u64)r * row + lane] = (uint8_t)idx;
}

// ---- re-bucketing (Brisk::reallocate, brisk/Brisk.hpp:202-224): the entries of an index become reads of exactly k nts ----
// one thread per word (16 nts) of the packed stream of n k-mers laid end to end: k-mer e = nts [e*k, (e+1)*k)
__global__ void __launch_bounds__(256) k_kmers_to_reads(const u64* __restrict__ lo, const u64* __restrict__ hi, u64 n, u32 k, u32* __restrict__ packed, u64 n_words,
                                                        u64* __restrict__ starts) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w <= n) starts[w] = w * (u64)k;
    if (w >= n_words) return;
    const u64 total = n * (u64)k;
    u32 v = 0;
    for (int i = 0; i < 16; i++) {
        const u64 q = w * 16 + i;
        u32 c = 0;
        if (q < total) {
            const u64 e = q / k;
            const u32 j = (u32)(q - e * k);  // nt j of the k-mer, from its left end: bits [2(k-1-j), 2(k-j))
            c = (u32)shr128(mk128(lo[e], hi[e]), 2 * (k - 1 - j)).lo & 3u;
        }
        v = (v << 2) | c;
    }
    packed[w] = v;
}
// every record of a re-bucketing scan holds one k-mer: it carries the count of the entry it came from as its multiplicity
__global__ void __launch_bounds__(256) k_set_multiplicity(u64* __restrict__ rec, u64 n_rec, u32 stride, const u32* __restrict__ tags, const uint8_t* __restrict__ cnt) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    rec[i * stride + stride - 1] |= HDR_HAS_MULT | ((u64)cnt[tags[i]] << 48);
}
