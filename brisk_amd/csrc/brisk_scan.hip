// brisk_scan.hip -- reads in: packing, synthetic reads, the super-k-mer scan (k_scan, k_scan2) and the chunk machinery
// for long sequences.  Included by brisk_kernels.hip (one translation unit).
// ===========================================================================
// ASCII -> 2-bit packed (nuc2int, Kmers.cpp:442-444), 16 bases per thread
__global__ void __launch_bounds__(256) k_pack_ascii(const uint8_t* __restrict__ bases, u64 n_bases, u32* __restrict__ packed, u64 n_words) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const u64 base = w * 16;
    u32 v = 0;
    if (base + 16 <= n_bases && ((uintptr_t)(bases + base) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(bases + base);
        const u32 ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) v = (v << 2) | ((ws[i] >> (8 * j + 1)) & 3u);
        }
    } else {
        for (int i = 0; i < 16; i++) {
            const u64 p = base + i;
            const u32 c = p < n_bases ? ((bases[p] >> 1) & 3u) : 0u;
            v = (v << 2) | c;
        }
    }
    packed[w] = v;
}

// ===========================================================================
// synthetic reads: splitmix64 n-th output; written packed.
__device__ __forceinline__ u64 sm_mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ u64 sm_u(u64 s, u64 i) { return sm_mix(s + (i + 1) * 0x9E3779B97F4A7C15ull); }
// genome letter index 0..3 = "ACGT" -> 2-bit code A0 C1 T2 G3
__device__ __forceinline__ u32 acgt_to_code(u32 i) { return i == 2 ? 3u : i == 3 ? 2u : i; }

// one thread per output word (16 nts) of the packed stream of fixed-length reads
__global__ void __launch_bounds__(256) k_synth(u64 genome_len, u64 first_read, u64 n_reads, u32 L, u64 seed_g, u64 seed_r,
                                               u32* __restrict__ packed, u64 n_words, u64* __restrict__ starts) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w <= n_reads) starts[w] = w * (u64)L;
    if (w >= n_words) return;
    const u64 total = n_reads * (u64)L;
    u32 v = 0;
    u64 cur_read = ~0ull, p = 0;
    u32 strand = 0;
    for (int i = 0; i < 16; i++) {
        const u64 q = w * 16 + i;
        u32 c = 0;
        if (q < total) {
            const u64 r = q / L;
            const u32 off = (u32)(q - r * L);
            if (r != cur_read) {
                cur_read = r;
                const u64 rid = first_read + r;
                p = sm_u(seed_r, 2 * rid) % (genome_len - L + 1);
                strand = (u32)(sm_u(seed_r, 2 * rid + 1) >> 63);
            }
            if (!strand)
                c = acgt_to_code((u32)(sm_u(seed_g, p + off) >> 62));
            else
                c = acgt_to_code((u32)(sm_u(seed_g, p + L - 1 - off) >> 62)) ^ 2u;
        }
        v = (v << 2) | c;
    }
    packed[w] = v;
}

// out[0] = sum over reads of max(0, len-k+1): the number of k-mer instances (an upper bound on records);
// out[1] = the share of it in reads of more than 1024 k-mers; its top bit: starts[] does not ascend (the host refuses the batch) (a record every ~(w+2)/2 k-mers there, while a
// short read makes a few records whatever its length).  One atomic pair per block.
__global__ void __launch_bounds__(256) k_count_kmers(const u64* __restrict__ starts, u64 n_reads, u32 k, unsigned long long* out) {
    __shared__ unsigned long long s_sum[4], s_long[4];
    unsigned long long acc = 0, lng = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        const u64 len = starts[r + 1] - starts[r];
        if (starts[r + 1] < starts[r]) {  // not a read table: a length that wraps would send the scan far outside the caller's buffer
            atomicOr(out + 1, 1ull << 63);
            continue;
        }
        if (len >= k) {
            acc += len - k + 1;
            if (len - k + 1 > 1024) lng += len - k + 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        lng += __shfl_down(lng, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = acc;
        s_long[threadIdx.x >> 6] = lng;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        const unsigned long long l = s_long[0] + s_long[1] + s_long[2] + s_long[3];
        if (l) atomicAdd(out + 1, l);
    }
}

// ===========================================================================
// k_scan v1: one lane per read runs the enumerator state machine of
// Kmers.cpp:509-603 exactly; each closed super-k-mer becomes one record.
struct ScanOut {
    u64* rec;                    // cap * stride words
    u64 cap;
    unsigned long long* n_rec;   // record cursor
    unsigned long long* hist;    // per partition: low 32 records, high 32 k-mer instances (may be null)
    u32* overflow;
    u32* tag;                    // query mode: read index per record; sequence mode: position of the first k-mer (may be null)
    u64* ret;                    // sequence mode: the minimizer value next() returns with the vector (may be null)
    // binned output (may be null; needs hist): record number i of partition p goes to bins[(p * bin_cap + i) * stride], the
    // rank i being what the partition's histogram counter held before this record; records beyond a bin go to ovf
    u64* bins;
    u32 bin_cap;
    u64* ovf;
    // the records beyond their bins: OVF_REGIONS regions of ovf_region_cap slots, each filled through its own counter.  One counter
    // for all of them was a same-address atomic WITH a return value per record: at k31 m15 b14 -- minimizers are the smallest hashes
    // of their windows and a partition is the TOP 24 bits of the hash there, so the partitions' loads are skewed and 3.7 % of the
    // records lie beyond a bin of twice the mean -- those 9.7 M serialised atomics were 21 of the scan's 43 ms per 20 M reads.
    u32 ovf_region_cap;
    u32* ovf_cnt;
};
#define OVF_REGIONS 4096u   // a record's region: the low bits of its partition

struct MiniState {
    u64 mini;
    u32 pos;
    bool rev;
};

// get_minimizer (Kmers.cpp:367-408) of the K-mer at stream nts [q, q+K): the
// re-scan runs over the LOW 64 BITS of the k-mer only (line 371, F2).
__device__ MiniState rescan_minimizer(const u32* __restrict__ packed, u64 q, u32 K, u32 m, u64 M, const double* coef) {
    const u32 nlow = K < 32 ? K : 32;
    const u64 low = load_nts(packed, q + K - nlow, nlow);
    u64 cur = low;
    u64 fwd = cur & M;
    u64 rc = rc64(fwd, m);
    MiniState s;
    s.mini = fwd < rc ? fwd : rc;
    s.rev = s.mini != fwd;
    s.pos = 0;
    u64 best = order_key(s.mini, m, M, coef);
    int canon = -1;  // canonized(seq,K), evaluated on first use
    for (u32 i = 1; i <= K - m; i++) {
        cur >>= 2;
        fwd = cur & M;
        rc = rc64(fwd, m);
        const u64 c = fwd < rc ? fwd : rc;
        const u64 h = order_key(c, m, M, coef);
        if (h < best) {
            s.pos = i;
            s.mini = c;
            s.rev = c != fwd;
            best = h;
        } else if (h == best) {
            const u32 d = K - m - i;
            if (d < s.pos) {
                s.pos = d;
                s.mini = c;
                s.rev = c != fwd;
            } else if (d == s.pos) {
                if (canon < 0) {
                    const u64 hi = K > 32 ? load_nts(packed, q, K - 32) : 0;
                    canon = canonized_as_executed(mk128(low, hi), K) ? 1 : 0;
                }
                if (!canon) {
                    s.pos = d;
                    s.mini = c;
                    s.rev = false;
                }
            }
        }
    }
    return s;
}

// records a super-k-mer becomes: one, or -- where minimizer_idx classes extend the routing id (BriskParams::cls_bits) -- one
// per class its k-mers fall into (their minimizer_idx rises by one per element: idx_end - (n - 1) .. idx_end)
template <bool CLS>
__device__ __forceinline__ u32 superkmer_pieces(const BriskParams& P, u32 n, u32 idx_end) {
    return (CLS ? P.cls_bits : 0u) ? cls_of(P, idx_end) - cls_of(P, idx_end - (n - 1)) + 1 : 1u;
}

// ---- a super-k-mer's nucleotides as one wide value: 256 bits in general (2k-m <= 125 nts), 128 bits where 2k-m <= 64 (k <= 32 with
// the usual m: half the shifts and masks of the record builder, which is most of the scan at short k -- 13 records per 150-bp read
// at k31 m15 against 3.2 at k63 m21)
__device__ __forceinline__ W4 x_shr(W4 a, u32 s) { return w4_shr(a, s); }
__device__ __forceinline__ W4 x_shl(W4 a, u32 s) { return w4_shl(a, s); }
__device__ __forceinline__ W4 x_and(W4 a, W4 b) { return w4_and(a, b); }
__device__ __forceinline__ W4 x_andn(W4 a, W4 b) { return w4_andn(a, b); }
__device__ __forceinline__ W4 x_or(W4 a, W4 b) { return w4_or(a, b); }
__device__ __forceinline__ W4 x_xor(W4 a, W4 b) { return W4{a.w0 ^ b.w0, a.w1 ^ b.w1, a.w2 ^ b.w2, a.w3 ^ b.w3}; }
__device__ __forceinline__ W4 x_rc(W4 a, u32 len) { return w4_rc(a, len); }
__device__ __forceinline__ u64 x_w0(W4 a) { return a.w0; }
__device__ __forceinline__ u64 x_word(W4 a, u32 i) { return i == 0 ? a.w0 : i == 1 ? a.w1 : i == 2 ? a.w2 : a.w3; }
__device__ __forceinline__ u128x x_shr(u128x a, u32 s) { return shr128(a, s); }
__device__ __forceinline__ u128x x_shl(u128x a, u32 s) { return shl128(a, s); }
__device__ __forceinline__ u128x x_and(u128x a, u128x b) { return and128(a, b); }
__device__ __forceinline__ u128x x_andn(u128x a, u128x b) { return andn128(a, b); }
__device__ __forceinline__ u128x x_or(u128x a, u128x b) { return or128(a, b); }
__device__ __forceinline__ u128x x_xor(u128x a, u128x b) { return u128x{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ u128x x_rc(u128x a, u32 len) {  // true reverse complement of a len-nt value, len in [1,64]
    const u64 c = 0xaaaaaaaaaaaaaaaaull;
    return shr128(u128x{rev_nts64(a.hi ^ c), rev_nts64(a.lo ^ c)}, 128 - 2 * len);
}
__device__ __forceinline__ u64 x_w0(u128x a) { return a.lo; }
__device__ __forceinline__ u64 x_word(u128x a, u32 i) { return i == 0 ? a.lo : i == 1 ? a.hi : 0ull; }
template <class T> __device__ __forceinline__ T x_mask(u32 bits);
template <> __device__ __forceinline__ W4 x_mask<W4>(u32 bits) { return w4_mask(bits); }
template <> __device__ __forceinline__ u128x x_mask<u128x>(u32 bits) { return mask128(bits); }
template <class T> __device__ __forceinline__ T x_from64(u64 v);
template <> __device__ __forceinline__ W4 x_from64<W4>(u64 v) { return W4{v, 0, 0, 0}; }
template <> __device__ __forceinline__ u128x x_from64<u128x>(u64 v) { return u128x{v, 0}; }
template <class T> __device__ __forceinline__ T x_load(const u32* __restrict__ packed, u64 q, u32 len);
template <> __device__ __forceinline__ W4 x_load<W4>(const u32* __restrict__ packed, u64 q, u32 len) { return load_span(packed, q, len); }
template <> __device__ __forceinline__ u128x x_load<u128x>(const u32* __restrict__ packed, u64 q, u32 len) {  // len in [1,64]
    const u32 c = len >= 32 ? 32 : len;
    u128x r{load_nts(packed, q + len - c, c), 0};
    if (len > 32) r.hi = load_nts(packed, q, len - 32);
    return r;
}

// Build and append the record(s) of one super-k-mer: k-mers at read positions
// [p0, p0+n), vector reversed if `rev` (Kmers.cpp:554-556,597-599); idx_end is
// the minimizer_idx of the LAST element of the returned vector.  `slot`: the first of superkmer_pieces() consecutive
// slots (classic output); binned output takes its slots from the partitions' histogram counters.
template <bool CLS, class T>  // CLS false: the caller knows there are no minimizer_idx classes (m >= 12); T: W4 or u128x
__device__ __forceinline__ void emit_record_wide(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                                                 u32 idx_end, const ScanOut& out, u32 tag, u64 ret, unsigned long long slot) {
    const u32 L = P.k + n - 1;
    T S = x_load<T>(packed, q0 + p0, L);
    if (rev) S = x_rc(S, L);
    // minimizer of every k-mer of the vector = the m-mer at suffix offset idx_end
    // of the last one (hash_kmer_minimizer_inplace re-extracts it, Kmers.cpp:191-200)
    const u64 mm = x_w0(x_shr(S, 2 * idx_end)) & P.m_mask;
    const u64 h = mix2m(mm, P.m_mask);
    const u32 rbase = routing_base(P, h);  // Brisk.hpp:135-137, plus the extra routing bits of the hash
    // replace the minimizer by its hash (replace_slice, Kmers.cpp:149-159): both have 2m bits, so XOR-ing their difference
    // into place does it with one shift
    S = x_xor(S, x_shl(x_from64<T>(mm ^ h), 2 * idx_end));
    // drop the b bucket nts at suffix offset idx_end + suff_reduc (get_compacted, Kmers.cpp:138-145)
    const u32 cut = idx_end + P.suff_reduc;
    const T lowm = x_mask<T>(2 * cut);
    const T C = x_or(x_andn(x_shr(S, 2 * P.b), lowm), x_and(S, lowm));
    const u32 idx_first = idx_end - (n - 1);

    // elements [j0, j1) of the vector share a routing id: all of them without classes
    for (u32 j0 = 0; j0 < n;) {
        u32 j1 = n, bucket = rbase;
        if ((CLS ? P.cls_bits : 0u)) {
            const u32 c = cls_of(P, idx_first + j0);
            if (c + 1 < (1u << (CLS ? P.cls_bits : 0u))) j1 = min(n, (c + 1) * P.cls_width - idx_first);
            bucket = (rbase << (CLS ? P.cls_bits : 0u)) | c;
        }
        const u32 np = j1 - j0;
        // the piece's compacted string: its last k-mer ends n - j1 nts before the vector's
        T Cp = C;
        if (np != n) Cp = x_and(x_shr(C, 2 * (n - j1)), x_mask<T>(2 * (P.kb + np - 1)));
        u64* r;
        if (out.bins) {
            // One pass over the records instead of two: the histogram atomic every record pays anyway returns the record's
            // rank in its partition, and the record goes straight to that slot of the partition's bin.  These random 32-byte
            // stores ride along with a kernel that is bound by its vector instructions; k_scatter (209 M of the same stores and
            // nothing else: 10-12 ms per 50 M reads) and the staging copy it read are gone.
            const u32 part = bucket >> P.shift;
#ifdef SCAN_ATTR_NOATOMIC  // attribution builds: wrong results, timing only
            const u32 rank = (part * 7u + (threadIdx.x & 63)) % out.bin_cap;
#else
            // (Non-temporal record stores, to keep the histogram in the memory-side cache: 31-33 ms against 21.0 at k31 m15, 27.1 against
            // 25.6 at k63.)
            // (Issuing all of a flush's atomics first -- a record's partition needs its minimizer only -- and building the records afterwards
            // was measured: 26.1 against 26.0 ms per 50 M reads at k63, 43.4 against 42.9 per 20 M at k31 m15.  What the records waited
            // for was not this atomic but the one behind the overflow area: ScanOut::ovf_cnt.)
            const u32 rank = (u32)atomicAdd(&out.hist[part], 1ull | ((unsigned long long)np << 32));
#endif
            if (rank < out.bin_cap) {
                r = out.bins + ((u64)part * out.bin_cap + rank) * P.stride;
                if (out.tag) out.tag[(u64)part * out.bin_cap + rank] = tag;  // query mode: the records' reads, laid out like the records
            } else {
                const u32 reg = part & (OVF_REGIONS - 1);
                const u32 at = atomicAdd(&out.ovf_cnt[reg], 1u);
                if (at >= out.ovf_region_cap) {
                    *out.overflow = 1;
                    return;
                }
                const u64 o = (u64)reg * out.ovf_region_cap + at;
                r = out.ovf + o * P.stride;
                if (out.tag) out.tag[((u64)out.bin_cap << P.part_bits) + o] = tag;
            }
        } else {
            r = out.rec + slot * P.stride;
        }
#ifdef SCAN_ATTR_NOSTORE  // attribution builds: the record is computed and (practically) never stored
        if ((x_word(Cp, 0) ^ x_word(Cp, 1)) == 0x123456789abcdefull)
#endif
        {
            r[0] = x_word(Cp, 0);
            if (P.nw > 1) r[1] = x_word(Cp, 1);
            if (P.nw > 2) r[2] = x_word(Cp, 2);
            if (P.nw > 3) r[3] = x_word(Cp, 3);
            r[P.nw] = rec_header(bucket, np, idx_first + j0 + P.suff_reduc);
        }
        if (!out.bins) {
            if (out.tag) out.tag[slot] = tag;
            if (out.ret) out.ret[slot] = ret;
            if (out.hist) atomicAdd(&out.hist[bucket >> P.shift], 1ull | ((unsigned long long)np << 32));
        }
        slot++;
        j0 = j1;
    }
}
#ifndef EMIT_SMALL
#define EMIT_SMALL 1   // 0: every span through the 256-bit path (A/B)
#endif
template <bool CLS>
__device__ void emit_record_at(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                               u32 idx_end, const ScanOut& out, u32 tag, u64 ret, unsigned long long slot) {
    if (!out.bins && slot + superkmer_pieces<CLS>(P, n, idx_end) > out.cap) {
        *out.overflow = 1;
        return;
    }
    // (a vector has at most k - m + 1 k-mers: its span at most 2k - m nts.  Wave-uniform; folds where k and m are constants)
    if (EMIT_SMALL && 2 * P.k - P.m <= 64) emit_record_wide<CLS, u128x>(P, packed, q0, p0, n, rev, idx_end, out, tag, ret, slot);
    else emit_record_wide<CLS, W4>(P, packed, q0, p0, n, rev, idx_end, out, tag, ret, slot);
}
__device__ void emit_record(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                            u32 idx_end, const ScanOut& out, u32 tag, u64 ret = 0) {
    // (sequence mode, out.ret: the vectors go back to the caller as SuperKmerEnumerator::next yields them -- whole;
    // BriskParams::cls_bits is 0 on the handles that serve it, brisk_hip_scan_sequence sees to that)
    emit_record_at<true>(P, packed, q0, p0, n, rev, idx_end, out, tag, ret, atomicAdd(out.n_rec, (unsigned long long)superkmer_pieces<true>(P, n, idx_end)));
}

// query_mode: stop after the first super-k-mer whose returned minimizer is 0,
// the first one excepted (counter.cpp:296-307)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan(BriskParams P, const u32* __restrict__ packed, const u64* __restrict__ starts,
                                                     u64 n_reads, const double* __restrict__ g_coef, ScanOut out, int query_mode) {
    __shared__ double s_coef[128];
    for (u32 i = threadIdx.x; i < 4 * P.m; i += blockDim.x) s_coef[i] = g_coef[i];
    __syncthreads();
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const u64 q0 = starts[r];
    const u64 len = starts[r + 1] - q0;
    const u32 k = P.k, m = P.m, w = P.w;
    if (len < k) return;  // counter.cpp:233-235
    const u64 M = P.m_mask;

    // candidate m-mer state: bases [k-m-1, k-1), forward keeps m-1 of them (Kmers.cpp:531)
    u64 cf = 0, cr = 0;
    for (u32 i = 0; i < m; i++) {
        const u32 c = nt_at(packed, q0 + k - m - 1 + i);
        cf = ((cf << 2) + c) & (M >> 2);
        cr = (cr >> 2) + ((u64)(c ^ 2u) << (2 * m - 2));
    }
    MiniState st = rescan_minimizer(packed, q0, k - 1, m, M, s_coef);  // Kmers.cpp:533
    u64 mini_hash = order_key(st.mini, m, M, s_coef);
    u32 mini_pos = st.pos;
    bool reversed = st.rev;
    u64 mini = st.mini;

    const u32 nk = (u32)(len - k + 1);
    u32 n = 0, p0 = 0, first_idx = 0, last_idx = 0, n_emitted = 0;
    for (u32 p = 0; p < nk; p++) {
        const u32 c = nt_at(packed, q0 + k - 1 + p);
        cf = ((cf << 2) + c) & M;
        cr = (cr >> 2) + ((u64)(c ^ 2u) << (2 * m - 2));
        mini_pos++;
        const u64 cand = cf < cr ? cf : cr;
        const u64 h = order_key(cand, m, M, s_coef);
        bool closed = false;
        const bool old_rev = reversed;
        u64 ret = 0;
        if (mini_pos > w) {  // the minimizer left the k-mer (Kmers.cpp:551-562)
            closed = true;
            ret = mini;
            st = rescan_minimizer(packed, q0 + p, k, m, M, s_coef);
            mini = st.mini;
            mini_pos = st.pos;
            reversed = st.rev;
            mini_hash = order_key(mini, m, M, s_coef);
        } else if (h < mini_hash) {  // strictly smaller candidate (Kmers.cpp:564-577)
            closed = true;
            ret = mini;
            mini_hash = h;
            mini_pos = 0;
            mini = cand;
            reversed = cand == cr;
        }
        const u32 idx = reversed ? w - mini_pos : mini_pos;  // Kmers.cpp:578-584
        if (closed && p > 0) {  // a close at p == 0 is ignored (Kmers.cpp:585-592)
            if (query_mode && n_emitted > 0 && ret == 0) return;
            emit_record(P, packed, q0, p0, n, old_rev, old_rev ? first_idx : last_idx, out, out.ret ? p0 : (u32)r, ret);
            n_emitted++;
            n = 0;
        }
        if (n == 0) {
            p0 = p;
            first_idx = idx;
        }
        last_idx = idx;
        n++;
    }
    if (n > 0) {  // Kmers.cpp:596-601
        if (query_mode && n_emitted > 0 && mini == 0) return;
        emit_record(P, packed, q0, p0, n, reversed, reversed ? first_idx : last_idx, out, out.ret ? p0 : (u32)r, mini);
    }
}

__device__ __forceinline__ unsigned long long lanes_below(u32 lane) { return (1ull << lane) - 1; }

// ===========================================================================
// k_scan2: the production scan.  Same results as k_scan (kept above as the plain
// restatement used for A/B), restructured for the wave:
//   * one lane per read steps the candidate m-mer; its order key is a table-driven
//     decycling class (4-nt chunk sums in LDS, exact fold only inside a 1e-9 guard
//     band around +-eps) plus the integer mixer;
//   * a re-scan (get_minimizer, Kmers.cpp:367-408) is done by a half-wave, one window per lane
//     straight from the k-mer's low 64 bits (zero-padded "fake" windows included, F2), two
//     k-mers per round, using the closed form of the tie rules (first and last position of
//     the minimum key);
//   * closed super-k-mers are queued in LDS and turned into records by full waves.
// ---- layout of the decycling chunk tables (shared by the host builder in brisk_capi.hip and the kernels) ----------------
// An m-mer is cut into L = ceil(m / CLS_W) chunks of at most CLS_W nts, as even as they come, low nts first: CLS_W = 5 gives
// m = 21 -> [5,4,4,4,4], 15 -> [5,5,5], 11 -> [4,4,3], 31 -> [5,5,5,4,4,4,4].  One look-up per chunk (tools/valu_rates.hip:
// a random ds_read_b64 with its address arithmetic costs ~26 cycles per wave against ~4.3 for a vector instruction), so wider
// chunks mean fewer look-ups -- CLS_W = 6: m = 21 -> [6,5,5,5], four instead of five -- but also tables of 57 KB instead of
// 16, and the LDS decides how many waves stay resident.  Measured at m = 21, ms per 50 M reads: CLS_W 6, two 8-wave blocks per
// CU (4 waves per SIMD, all that fits): 30.3; CLS_W 4 / 5, three blocks (6 waves per SIMD, 80 registers, no scratch):
// 26.1 / 25.6; four blocks (64 registers, 16 of them spilled): 33.9; one 16-wave block per CU 43.9, 10-wave blocks 49
// (profiles/r02_scan_attribution.txt).  Occupancy is worth more than a look-up, scratch costs more than occupancy.
#ifndef CLS_W
#define CLS_W 5
#endif
__host__ __device__ constexpr u32 cls_nch(u32 m) { return (m + CLS_W - 1) / CLS_W; }
__host__ __device__ constexpr u32 cls_width(u32 m, u32 c) {  // nts in chunk c
    return m / cls_nch(m) + (c < m % cls_nch(m) ? 1u : 0u);
}
__host__ __device__ constexpr u32 cls_off(u32 m, u32 c) {  // first nt of chunk c
    u32 o = 0;
    for (u32 j = 0; j < c; j++) o += cls_width(m, j);
    return o;
}
__host__ __device__ constexpr u32 cls_base(u32 m, u32 c) {  // first table entry of chunk c; c = cls_nch(m): entries in all
    u32 b = 0;
    for (u32 j = 0; j < c; j++) b += 1u << (2 * cls_width(m, j));
    return b;
}
#define CLS_MAX_CHUNKS 8

struct ScanCfg {
    u32 nlow;     // nts of a k-mer that get_minimizer sees: min(32, k)   (F2)
    u32 nlow1;    // same for the (k-1)-mer
    u32 nch;      // chunks of an m-mer: cls_nch(m)
    u32 qcap;     // emit queue entries per wave
    u32 n_tab;    // doubles staged to LDS: coef[128] + the chunk tables
    u32 chunk[CLS_MAX_CHUNKS];  // run-time layout for the generic kernels: first bit (6 b) | bits (4 b) << 6 | first entry << 10
};

// Decycling class from packed fixed-point chunk tables.  tabs[base_c + v] (one u64 per value v of chunk c) is the 64-bit
// integer a + b * 2^32: a = the chunk's share of R(x), b = its share of R(rot(x)), both signed, in units of 2^-24.  The sum
// of an m-mer's words is A + B * 2^32 with the whole sums A, B (|A|, |B| < 2^30): one look-up and one 64-bit add per chunk
// give both.  Every a and b is rounded to a unit (<= 0.5 unit of error each, <= 4 units per sum for up to 8 chunks; the
// reference's own FP64 fold is within 1e-5 unit of the exact value), eps = 1e-6 is 16.78 units: a sum >= 21 is certainly
// > eps, a sum <= 12 certainly < eps, anything in [13, 20] (either sign) is decided by the exact FP64 fold in the
// reference's order (decy_class).  Host side: brisk_hip_create in brisk_capi.hip.
#define CLS_HI 21
#define CLS_LO 12
// NCH > 0: compile-time chunk count (unrolled look-ups); MM > 0: compile-time m, the whole layout folds into the instructions
template <int NCH, int MM>
__device__ __forceinline__ u64 cls_sums(u64 x, const ScanCfg& cfg, const u64* tabs) {
    u64 acc;
    if (MM > 0) {
        constexpr u32 mm = MM > 0 ? (u32)MM : 1u;
        acc = tabs[(u32)x & ((1u << (2 * cls_width(mm, 0))) - 1)];
#pragma unroll
        for (u32 c = 1; c < cls_nch(mm); c++) acc += tabs[cls_base(mm, c) + ((u32)(x >> (2 * cls_off(mm, c))) & ((1u << (2 * cls_width(mm, c))) - 1))];
    } else if (NCH > 0) {
        acc = tabs[(u32)x & ((1u << ((cfg.chunk[0] >> 6) & 15)) - 1)];
#pragma unroll
        for (int c = 1; c < NCH; c++) {
            const u32 d = cfg.chunk[c];
            acc += tabs[(d >> 10) + ((u32)(x >> (d & 63)) & ((1u << ((d >> 6) & 15)) - 1))];
        }
    } else {
        acc = tabs[(u32)x & ((1u << ((cfg.chunk[0] >> 6) & 15)) - 1)];
        for (u32 c = 1; c < cfg.nch; c++) {
            const u32 d = cfg.chunk[c];
            acc += tabs[(d >> 10) + ((u32)(x >> (d & 63)) & ((1u << ((d >> 6) & 15)) - 1))];
        }
    }
    return acc;
}
// A sum inside the guard band sets `guard` (the class returned is then not to be used) and the caller goes to the exact fold: a
// lone key behind one wave-uniform branch (decy_class_fast), a loop that can be repeated (the (k-1)-mer's windows) as a whole, so
// that nothing inside it branches.
template <int NCH, int MM>
__device__ __forceinline__ u32 decy_class_guarded(u64 x, const ScanCfg& cfg, const u64* tabs, bool& guard) {
    const u64 acc = cls_sums<NCH, MM>(x, cfg, tabs);
    const int a = (int)(u32)acc, b = (int)(u32)(acc >> 32) - (a >> 31);
    const u32 ua = (u32)(a < 0 ? -a : a), ub = (u32)(b < 0 ? -b : b);
    guard = guard | (ua - (CLS_LO + 1) <= (u32)(CLS_HI - CLS_LO - 2)) | (ub - (CLS_LO + 1) <= (u32)(CLS_HI - CLS_LO - 2));
#ifdef CLS_FORCE_GUARD  // test builds: every pass is repeated with the exact fold (the path a guard-band sum takes once in a blue moon)
    guard = true;
#endif
    const bool c0 = a >= CLS_HI && b <= CLS_LO, c1 = a <= -CLS_HI && b >= -CLS_LO;
    return c0 ? 0u : c1 ? 1u : 2u;
}
template <int NCH, int MM>
__device__ __forceinline__ u32 decy_class_fast(u64 x, u32 m, const ScanCfg& cfg, const u64* tabs, const double* coef) {
#ifdef SCAN_ATTR_NOCLASS  // attribution builds (tools/scan_attribution.py): wrong results, timing only
    return (u32)x & 1u;
#endif
    bool guard = false;
    u32 cls = decy_class_guarded<NCH, MM>(x, cfg, tabs, guard);
    if (__ballot(guard)) {  // wave-uniform and practically never taken: no lane-level branch on the way of the others
        if (guard) cls = decy_class(x, m, coef);
    }
    return cls;
}
template <int NCH = 0, int MM = 0>
__device__ __forceinline__ u64 order_key_fast(u64 x, u32 m, u64 M, const ScanCfg& cfg, const u64* tabs, const double* coef) {
#ifdef SCAN_ATTR_NOMIX
    return ((u64)decy_class_fast<NCH, MM>(x, m, cfg, tabs, coef) << 62) + (x ^ (x >> 7));
#endif
    return ((u64)decy_class_fast<NCH, MM>(x, m, cfg, tabs, coef) << 62) + mix2m(x, M);
}

__global__ void __launch_bounds__(256) k_debug_keys(BriskParams P, ScanCfg cfg, const double* __restrict__ g_tabs, const u64* __restrict__ x, u64 n,
                                                    int exact, u64* __restrict__ out) {
    extern __shared__ double smem_d[];
    for (u32 i = threadIdx.x; i < cfg.n_tab; i += blockDim.x) smem_d[i] = g_tabs[i];
    __syncthreads();
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = exact ? order_key(x[i], P.m, P.m_mask, smem_d) : order_key_fast(x[i], P.m, P.m_mask, cfg, (const u64*)(smem_d + 128), smem_d);
}

// value of a wave-uniform lane, through SGPRs
__device__ __forceinline__ u64 read_lane_u64(u64 v, int L) {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, L), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), L);
    return ((u64)hi << 32) | lo;
}

// ---- long sequences: scanned as overlapping chunks, each chunk a "virtual read" ----------------
// The enumerator's state (minimizer key, its position, its strand) before a step depends on history, so a
// chunk starts SCAN_WARMUP steps early from a fresh state, emits only the vectors that start inside its
// own window [emit_from, emit_until), and exports the state it had reached at emit_from (spec); its
// predecessor exports the state it had at the same step (truth).  Equal state + same nucleotides =>
// identical stream from there on.  A chunk is EXACT when its predecessor is exact and the two states
// match (the first chunk of a sequence is exact by definition); matching against a predecessor that is
// itself wrong proves nothing (two cold starts can agree with each other inside a periodic region and
// both be out of phase with the sequential run).  A chunk whose states do not match is scanned again
// from its window's first step, SEEDED with the exact state its predecessor exported: no warm-up, no
// speculation.  k_chunk_match / k_chunk_commit extend exactness along every sequence as far as it reaches and
// list the chunks to re-scan; the host repeats until every chunk is exact (one round per mismatch along a
// sequence: long runs without a new minimum -- homopolymers, short tandem repeats).
#define SCAN_LONG 8192u     // sequences with more k-mers than this are chunked
#define SCAN_WARMUP 512u    // steps a speculative chunk runs before its window
struct VRead {
    u64 q0;          // stream index of the virtual read's first nt
    u32 len;         // nts
    u32 emit_from;   // local step of the first vector start that belongs to this chunk
    u32 emit_until;  // local step bound (exclusive); ~0u: to the end of the sequence
    u32 read;        // index of the sequence in the batch
    u32 flags;       // 1: first chunk of its sequence, 2: runs to the sequence's last k-mer, 4: seeded start
    u32 slot;        // chunk index: where its states live, and the tag of its records
    u32 first;       // chunk index of its sequence's first chunk
    u32 pad;
};
struct ChunkState {
    u64 hash;
    u32 pos_rev;     // mini_pos | reversed << 31
    u32 set;
};
struct ChunkCtl {
    const VRead* vreads;   // null: whole reads from `starts`
    ChunkState* spec;      // [n_chunks]   state a speculative chunk reached at its emit_from
    ChunkState* truth;     // [n_chunks+1] state the previous chunk had at the same step; seed of a seeded chunk
    u32 long_limit;        // whole-read launch: skip reads with more k-mers than this (0: none)
};
#define CHUNK_EXACT 1u      // chunk status bits
#define CHUNK_RERUN 2u      // its speculative records are void, a seeded scan replaced them

// plan the chunks of long reads (consecutive slots per read): one thread per read reserves the slots, then one
// block per long read writes them (a chromosome is ~10^5 chunks)
struct LongRead {
    u32 read, base, n_chunks, pad;
};
__global__ void __launch_bounds__(256) k_plan_chunks(const u64* __restrict__ starts, u64 n_reads, u32 k, u32 chunk, u32 cap, u32* __restrict__ n_vreads,
                                                     LongRead* __restrict__ longs, u32* __restrict__ n_long) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const u64 len = starts[r + 1] - starts[r];
    if (len < k) return;
    const u64 nk = len - k + 1;
    if (nk <= SCAN_LONG) return;
    const u32 nc = (u32)((nk + chunk - 1) / chunk);
    const u32 base = atomicAdd(n_vreads, nc);
    if (base + nc > cap) return;  // cannot happen: cap is the bound the host computed
    longs[atomicAdd(n_long, 1u)] = LongRead{(u32)r, base, nc, 0u};
}
__global__ void __launch_bounds__(256) k_fill_chunks(const u64* __restrict__ starts, u32 k, u32 w, u32 chunk, const LongRead* __restrict__ longs, u32 n_long,
                                                     VRead* __restrict__ vreads) {
    for (u32 li = blockIdx.x; li < n_long; li += gridDim.x) {
        const LongRead lr = longs[li];
        const u64 q0 = starts[lr.read], nk = starts[lr.read + 1] - q0 - k + 1;
        for (u32 c = threadIdx.x; c < lr.n_chunks; c += blockDim.x) {
            const u64 b0 = (u64)c * chunk, b1 = b0 + chunk;
            const u64 s0 = c == 0 ? 0 : b0 - SCAN_WARMUP;
            const bool last = b1 >= nk;
            const u64 end_step = last ? nk : (b1 + w + 2 < nk ? b1 + w + 2 : nk);
            VRead v;
            v.q0 = q0 + s0;
            v.len = (u32)(end_step - s0 + k - 1);
            v.emit_from = (u32)(b0 - s0);
            v.emit_until = last ? 0xffffffffu : (u32)(b1 - s0);
            v.read = lr.read;
            v.flags = (c == 0 ? 1u : 0u) | (end_step == nk ? 2u : 0u);
            v.slot = lr.base + c;
            v.first = lr.base;
            v.pad = 0;
            vreads[lr.base + c] = v;
        }
    }
}
// One round of the exactness walk, one thread per chunk (a single chromosome is ~10^5 chunks: no serial walk).
// cursor[first] = first chunk of the sequence not yet known exact; stop[first] = first chunk at or after the
// cursor whose speculative state does not match what its predecessor exported.  Chunks in [cursor, stop) are
// exact by induction (each matches the export of an exact predecessor); chunk `stop` is queued for a seeded
// re-scan from the exact state in truth[stop]; the walk resumes behind it in the next round.
__global__ void __launch_bounds__(256) k_chunk_match(const VRead* __restrict__ vreads, const ChunkState* __restrict__ spec,
                                                     const ChunkState* __restrict__ truth, u32 n_chunks, const u32* __restrict__ cursor,
                                                     u32* __restrict__ stop) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks) return;
    const u32 f = vreads[i].first;
    const u32 c = cursor[f] > f + 1 ? cursor[f] : f + 1;
    if (i < c) return;
    const ChunkState a = spec[i], b = truth[i];
    if (!(a.set && b.set && a.hash == b.hash && a.pos_rev == b.pos_rev)) atomicMin(&stop[f], i);
}
__global__ void __launch_bounds__(256) k_chunk_commit(const VRead* __restrict__ vreads, u32 n_chunks, u32 chunk, u32 k, u32 w,
                                                      const u64* __restrict__ starts, const u32* __restrict__ cursor, const u32* __restrict__ stop,
                                                      u32* __restrict__ status, VRead* __restrict__ rerun, u32* __restrict__ n_rerun) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks) return;
    const VRead me = vreads[i];
    const u32 f = me.first;
    const u32 c = cursor[f] > f + 1 ? cursor[f] : f + 1;
    const u32 st = stop[f];
    if (i == f) status[i] |= CHUNK_EXACT;
    if (i < c || i > st) return;
    if (i < st) {
        status[i] |= CHUNK_EXACT;
        return;
    }
    // i == st: scan this chunk again from its window's first step, from the exact state in truth[i]
    const u64 q0 = starts[me.read], nk = starts[me.read + 1] - q0 - k + 1;
    const u64 b0 = (u64)(i - f) * chunk, b1 = b0 + chunk;
    const bool last = b1 >= nk;
    const u64 end_step = last ? nk : (b1 + w + 2 < nk ? b1 + w + 2 : nk);
    VRead v;
    v.q0 = q0 + b0;
    v.len = (u32)(end_step - b0 + k - 1);
    v.emit_from = 0;
    v.emit_until = last ? 0xffffffffu : (u32)(b1 - b0);
    v.read = me.read;
    v.flags = 4u | (end_step == nk ? 2u : 0u);
    v.slot = i;
    v.first = f;
    v.pad = 0;
    rerun[atomicAdd(n_rerun, 1u)] = v;
    status[i] = CHUNK_EXACT | CHUNK_RERUN;  // exact once the re-scan launched after this kernel has run
}
__global__ void __launch_bounds__(256) k_chunk_next(const VRead* __restrict__ vreads, u32 n_chunks, u32* __restrict__ cursor, u32* __restrict__ stop) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks || vreads[i].first != i) return;
    cursor[i] = stop[i] == 0xffffffffu ? 0xffffffffu : stop[i] + 1;
    stop[i] = 0xffffffffu;
}
// ---- a query over chunked sequences (query_sequence stops a sequence at the first super-k-mer, other than its
// first, whose returned minimizer is 0, apps/counter.cpp:304-306).  Chunks cannot know what happened before them,
// so they emit everything, every record carrying where its vector starts and whether its minimizer is 0
// (ScanOut::ret); once the chunks are exact, k_query_break finds each sequence's stop and k_query_filter drops what
// lies at or behind it, along with the speculative records of re-scanned chunks, and tags the rest with their read.
__global__ void __launch_bounds__(256) k_query_break(const u64* __restrict__ ret, const u32* __restrict__ tags, u64 first, u64 n_spec_end, u64 n_rec,
                                                     const VRead* __restrict__ vreads, const u32* __restrict__ status, const u64* __restrict__ starts,
                                                     unsigned long long* __restrict__ brk) {
    const u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u32 slot = tags[i];
    if (i < n_spec_end && (status[slot] & CHUNK_RERUN)) return;
    const u64 r = ret[i], q = r & 0x7fffffffffffffffull;
    const VRead v = vreads[slot];
    if ((r >> 63) && q > starts[v.read]) atomicMin(&brk[v.first], (unsigned long long)q);
}
__global__ void __launch_bounds__(256) k_query_filter(BriskParams P, const u64* __restrict__ rec, const u64* __restrict__ ret, const u32* __restrict__ tags, u64 first,
                                                      u64 n_spec_end, u64 n_rec, const VRead* __restrict__ vreads, const u32* __restrict__ status,
                                                      const unsigned long long* __restrict__ brk, u64* __restrict__ out, u32* __restrict__ tag_out,
                                                      unsigned long long* __restrict__ n_out) {
    const u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u32 slot = tags[i];
    if (i < n_spec_end && (status[slot] & CHUNK_RERUN)) return;
    const VRead v = vreads[slot];
    if ((ret[i] & 0x7fffffffffffffffull) >= brk[v.first]) return;
    const unsigned long long o = atomicAdd(n_out, 1ull);
    for (u32 j = 0; j < P.stride; j++) out[o * P.stride + j] = rec[i * P.stride + j];
    tag_out[o] = v.read;
}
// keep the speculative records of the chunks that were not re-scanned
__global__ void __launch_bounds__(256) k_filter_records(BriskParams P, const u64* __restrict__ rec, const u32* __restrict__ tags, u64 first, u64 n_rec,
                                                        const u32* __restrict__ status, u64* __restrict__ out, unsigned long long* __restrict__ n_out) {
    const u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    if (status[tags[i]] & CHUNK_RERUN) return;
    const unsigned long long slot = atomicAdd(n_out, 1ull);
    for (u32 j = 0; j < P.stride; j++) out[slot * P.stride + j] = rec[i * P.stride + j];
}

// closed form of get_minimizer's fold (Kmers.cpp:377-405) given the first and last window
// holding the minimum key, their `reversed` flags and K-m
__device__ __forceinline__ void resolve_ties(u32 first, u32 last, bool rev_first, bool rev_last, u32 Km, bool canon_if_needed_known, bool canon,
                                             u32* pos, bool* rev, bool* need_canon) {
    *need_canon = false;
    *pos = first;
    *rev = rev_first;
    if (last != first) {
        const u32 dT = Km - last;
        if (dT < first) {
            *pos = dT;
            *rev = rev_last;
        } else if (dT == first) {
            if (!canon_if_needed_known) *need_canon = true;
            else if (!canon) *rev = false;
        }
    }
}

// Emit queue entry (one u64 per closed super-k-mer): low word = the step (k-mer index in the lane's read) of its first
// k-mer; high word = n | idx_end << 8 | reversed << 16 | (returned minimizer == 0) << 17 | lane << 18.  The lane's read
// (stream index of its first nt, and its tag) is looked up in the wave's s_q0 / s_tag when the record is built.
template <bool CLS>
__device__ __forceinline__ void emit_queued(const BriskParams& P, const u32* __restrict__ packed, const ScanOut& out, u64 ent, const u64* s_q0, const u32* s_tag,
                                            unsigned long long slot) {
    const u32 mi = (u32)(ent >> 32), src = (mi >> 18) & 63u;
    const u64 q_start = s_q0[src] + (u32)ent;
    emit_record_at<CLS>(P, packed, q_start, 0, mi & 0xff, (mi >> 16) & 1, (mi >> 8) & 0xff, out, s_tag[src], q_start | ((u64)((mi >> 17) & 1) << 63), slot);
}

// records the wave's queued super-k-mers become
template <bool CLS>
__device__ __forceinline__ u32 queue_records(const BriskParams& P, const u64* q_ent, u32 qcount) {
    if (!(CLS ? P.cls_bits : 0u)) return qcount;
    u32 mine = 0;
    for (u32 e = threadIdx.x & 63; e < qcount; e += 64) {
        const u32 mi = (u32)(q_ent[e] >> 32);
        mine += superkmer_pieces<CLS>(P, mi & 0xff, (mi >> 8) & 0xff);
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    return mine;
}
// the queue's records into slots [base, base + queue_records())
template <bool CLS>
__device__ __forceinline__ void emit_queue(const BriskParams& P, const u32* __restrict__ packed, const ScanOut& out, const u64* q_ent, const u64* s_q0, const u32* s_tag,
                                           u32 qcount, unsigned long long base) {
    const u32 lane = threadIdx.x & 63;
    if (!(CLS ? P.cls_bits : 0u)) {
        for (u32 e = lane; e < qcount; e += 64) emit_queued<CLS>(P, packed, out, q_ent[e], s_q0, s_tag, base + e);
        return;
    }
    for (u32 e0 = 0; e0 < qcount; e0 += 64) {  // wave-uniform: the slots of 64 entries from an inclusive scan of their piece counts
        const u32 e = e0 + lane;
        u64 ent = 0;
        u32 np = 0;
        if (e < qcount) {
            ent = q_ent[e];
            const u32 mi = (u32)(ent >> 32);
            np = superkmer_pieces<CLS>(P, mi & 0xff, (mi >> 8) & 0xff);
        }
        u32 incl = np;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 y = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += y;
        }
        if (e < qcount) emit_queued<CLS>(P, packed, out, ent, s_q0, s_tag, base + incl - np);
        base += (u32)__shfl(incl, 63, 64);
    }
}

// what is left in the waves' queues when their reads end: one slot reservation for the whole block
template <bool CLS>
__device__ __forceinline__ void scan_final_flush(const BriskParams& P, const u32* __restrict__ packed, const ScanOut& out, const u64* q_ent, const u64* s_q0,
                                                 const u32* s_tag, u32 qcount, u32* s_wcnt, unsigned long long* s_wbase) {
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32 n_mine = queue_records<CLS>(P, q_ent, qcount);
    if (out.bins) {  // binned records need no slots: no reservation, no barrier
        if (lane == 0 && n_mine) atomicAdd(out.n_rec, (unsigned long long)n_mine);
        emit_queue<CLS>(P, packed, out, q_ent, s_q0, s_tag, qcount, 0);
        return;
    }
    if (lane == 0) s_wcnt[wid] = n_mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (u32 i = 0; i < nw; i++) total += s_wcnt[i];
        *s_wbase = total ? atomicAdd(out.n_rec, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = *s_wbase;
    for (u32 i = 0; i < wid; i++) base += s_wcnt[i];
    emit_queue<CLS>(P, packed, out, q_ent, s_q0, s_tag, qcount, base);
}

// minimum of x over the lane's 16-lane row, in every lane of the row (DPP row rotations: one instruction per step)
__device__ __forceinline__ u32 row_min_u32(u32 x) {
#define ROW_MIN_STEP(CTRL)                                                                      \
    {                                                                                           \
        const u32 y_ = (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false); \
        x = y_ < x ? y_ : x;                                                                    \
    }
    ROW_MIN_STEP(0x128)  // row_ror:8
    ROW_MIN_STEP(0x124)  // row_ror:4
    ROW_MIN_STEP(0x122)  // row_ror:2
    ROW_MIN_STEP(0x121)  // row_ror:1
#undef ROW_MIN_STEP
    return x;
}

// ---- k <= 32: the window minimum without re-scans ------------------------------------------------------------------------------
// get_minimizer sees the whole k-mer when k <= 32 (no truncation): the re-scan at an expiry is then the minimum over the
// last w + 1 candidates -- keys the step loop has already computed.  Per lane, the monotone queue of a sliding-window minimum, three
// elements deep, in registers: element i = (key, step it was born at mod 256, reversed flag); keys strictly increasing from the front
// (the current minimizer) to the back (newer candidates that can still become the minimizer once everything smaller has left the
// window).  A new candidate removes every element with a larger key and goes to the back; at an expiry the front leaves and the next
// element IS what the re-scan would find -- its key, its position (p - birth), its strand -- provided its key is unique in the window.
// What three elements cannot say sends the lane to the lane-parallel re-scan, which stays exact for everything:
//   * TRUNC: elements may be missing behind the last stored one (a fourth element was refused, or an equal key was seen).  Stored
//     elements are always an exact prefix of the true queue; a candidate is stored behind an unknown tail only if it removes the
//     last stored element (then it removes the tail too: the tail's keys are larger still);
//   * TIE: a candidate equal to a stored key -- the reference's tie rules need the first AND last window holding the minimum
//     (resolve_ties): the element is marked, never served.
// k31 m15 b14, 20 M reads: the re-scans were 14 of the scan's 26 ms (profiles/r03_scan_attribution.txt).
struct MinQueue {
    u64 k0, k1, k2;  // ~0: empty (no key has class 3)
    u32 meta;        // [0,8) [8,16) [16,24): birth steps mod 256; 24..26 reversed flags; 27..29 tie flags; 30 TRUNC
};
#define MQ_EMPTY (~0ull)
#define MQ_TRUNC (1u << 30)
__device__ __forceinline__ void mq_reset(MinQueue& q, u64 key, u32 birth, bool rv, bool trunc) {
    q.k0 = key;
    q.k1 = q.k2 = MQ_EMPTY;
    q.meta = (birth & 0xffu) | (rv ? 1u << 24 : 0u) | (trunc ? MQ_TRUNC : 0u);
}
__device__ __forceinline__ void mq_push(MinQueue& q, u64 h, u32 step, bool rv) {
    const bool l0 = q.k0 < h, l1 = q.k1 < h, l2 = q.k2 < h;  // a prefix: keys increase, empties are never smaller
    const bool e0 = q.k0 == h, e1 = q.k1 == h, e2 = q.k2 == h;
    const u32 cnt = (q.k0 != MQ_EMPTY ? 1u : 0u) + (q.k1 != MQ_EMPTY ? 1u : 0u) + (q.k2 != MQ_EMPTY ? 1u : 0u);
    const u32 keep = (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u);
    if (e0 | e1 | e2) {  // an equal key: mark it, drop what lies behind it (larger keys), and the tail is unknown from here
        const u32 j = e0 ? 0u : e1 ? 1u : 2u;
        if (j < 1) q.k1 = MQ_EMPTY;
        if (j < 2) q.k2 = MQ_EMPTY;
        q.meta |= (1u << (27 + j)) | MQ_TRUNC;
        return;
    }
    const bool store = keep < 3 && (!(q.meta & MQ_TRUNC) || keep < cnt);
    if (!store) {
        q.meta |= MQ_TRUNC;  // (full: the candidate joins the unknown tail; unknown tail and nothing removed: likewise)
        return;
    }
    const u32 sh = 8 * keep;
    q.meta = (q.meta & ~((0xffu << sh) | (1u << (24 + keep)) | (1u << (27 + keep)) | MQ_TRUNC)) | ((step & 0xffu) << sh) | (rv ? 1u << (24 + keep) : 0u);
    if (keep == 0) q.k0 = h;
    if (keep <= 1) q.k1 = keep == 1 ? h : MQ_EMPTY;
    if (keep <= 2) q.k2 = keep == 2 ? h : MQ_EMPTY;
}
__device__ __forceinline__ void mq_pop_front(MinQueue& q) {
    q.k0 = q.k1;
    q.k1 = q.k2;
    q.k2 = MQ_EMPTY;
    q.meta = ((q.meta >> 8) & 0xffffu) | (((q.meta >> 25) & 3u) << 24) | (((q.meta >> 28) & 3u) << 27) | (q.meta & MQ_TRUNC);
}

#ifdef BRISK_PHASE_PROF  // debug builds only (tools/phase_profile.py): event counts of k_scan2
__device__ unsigned long long g_scan_cnt[8];  // [0] wave-steps [1] expiries [2] re-scan rounds [3] of them with two k-mers [4] super-k-mers queued [5] flush passes
#define SCNT(i, v) scan_cnt[i] += (v);
#else
#define SCNT(i, v)
#endif

// MODE 0: reads, insert; 1: reads, query (stops a read at a returned minimizer of 0); 2: virtual reads (chunks of long sequences)
// KK, MM: k and m as compile-time constants for the common parameter sets (0: from P) -- folds the shifts and masks and,
// above all, frees scalar registers: the generic kernel spills 70+ of them into vector lanes and pays a v_readlane per use
// Waves per SIMD the kernel is compiled for: six (three 8-wave blocks per CU, 80 registers) where k and m are compile-time
// constants and there are no minimizer_idx classes -- those bodies fit 80 registers without scratch (or nearly: two of them
// at k31 m15) --, four for the others, which would spill 23-27.
#ifndef SCAN_WAVES_PER_EU
#define SCAN_WAVES_PER_EU 6
#endif
__host__ __device__ constexpr int scan_waves_per_eu(int KK, int MM) { return KK && MM >= 12 ? SCAN_WAVES_PER_EU : 4; }
template <int NCH, int MODE, int KK, int MM>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(scan_waves_per_eu(KK, MM), 8))) k_scan2(BriskParams P, ScanCfg cfg, const u32* __restrict__ packed, const u64* __restrict__ starts,
                                                u64 n_reads, const double* __restrict__ g_tabs, ScanOut out, ChunkCtl cc) {
    constexpr bool VR = MODE == 2, query_mode = MODE == 1;
    constexpr bool CLS = MM < 12;  // minimizer_idx classes in the routing id exist only where 2m < 24 (brisk_hip_create)
    extern __shared__ double smem_d[];
    const double* s_coef = smem_d;                        // 128
    const u64* s_tabs = (const u64*)(smem_d + 128);       // the packed fixed-point chunk tables
    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 n_tab = MM ? 128 + cls_base(MM ? MM : 1, cls_nch(MM ? MM : 1)) : cfg.n_tab;
    for (u32 i = tid; i < n_tab; i += blockDim.x) smem_d[i] = g_tabs[i];
    unsigned long long* s_wbase = (unsigned long long*)(smem_d + n_tab);      // block's slot base at the final flush
    u32* s_wcnt = (u32*)(s_wbase + 1);                                         // [16] records left per wave
    u64* q_ent = (u64*)(smem_d + n_tab + 9) + (size_t)wid * (cfg.qcap + 96);   // [qcap] the wave's emit queue
    u64* s_q0 = q_ent + cfg.qcap;                  // [64] stream index of every lane's first nt
    u32* s_tag = (u32*)(s_q0 + 64);                // [64] every lane's tag (read index, or chunk slot)

    if (KK) {  // the record builder reads these from P: let them fold there too
        P.k = KK;
        P.w = KK - MM;
    }
    if (MM) {
        P.m = MM;
        P.m_mask = (1ull << (2 * MM)) - 1;
    }
    const u32 k = KK ? (u32)KK : P.k, m = MM ? (u32)MM : P.m, w = k - m;
    const u64 M = MM ? ((1ull << (2 * MM)) - 1) : P.m_mask;
    const u32 ksh = 2 * m > 30 ? 2 * m - 30 : 0;  // the mix's bits below its top 30
    const u32 nlow = k < 32 ? k : 32, nlow1 = k - 1 < 32 ? k - 1 : 32;  // nts of a k-mer / (k-1)-mer that get_minimizer sees (F2)
    const u64 r = (u64)blockIdx.x * blockDim.x + tid;
    u64 q0 = 0, len = 0;
    u32 emit_from = 0, emit_until = 0xffffffffu, tagval = (u32)r, vslot = 0;
    bool seq_first = true, seq_last = true, seeded = false;
    if (r < n_reads) {
        if (VR) {
            const VRead v = cc.vreads[r];
            q0 = v.q0;
            len = v.len;
            emit_from = v.emit_from;
            emit_until = v.emit_until;
            tagval = vslot = v.slot;
            seq_first = v.flags & 1u;
            seq_last = v.flags & 2u;
            seeded = v.flags & 4u;
        } else {
            q0 = starts[r];
            len = starts[r + 1] - q0;
            if (cc.long_limit && len >= k && len - k + 1 > cc.long_limit) len = 0;  // a chunked launch takes this one
        }
    }
    s_q0[lane] = q0;
    s_tag[lane] = tagval;
    __syncthreads();
    const bool live = len >= k;  // counter.cpp:233-235
    u32 nk = live ? (u32)(len - k + 1) : 0;
    u32 max_nk = nk;
    for (int o = 32; o > 0; o >>= 1) {
        const u32 y = __shfl_xor(max_nk, o, 64);
        max_nk = y > max_nk ? y : max_nk;
    }
    if (max_nk == 0) {  // nothing to scan in this wave; it still takes part in the block's final reservation
        scan_final_flush<CLS>(P, packed, out, q_ent, s_q0, s_tag, 0, s_wcnt, s_wbase);
        return;
    }
    const u64 KEY0 = order_key_fast<NCH, MM>(0, m, M, cfg, s_tabs, s_coef);
    const u64 KEY0s = read_lane_u64(KEY0, 0);  // the same value, in scalar registers

    // ---- prologue: the low 64 bits of the (k-1)-mer; its last m-mer seeds the rolling candidates
    u64 cr = 0, low64 = 0;
    if (live) low64 = load_nts(packed, q0 + (k - 1) - nlow1, nlow1);
    cr = rc64(low64 & M, m);

    // ---- minimizer of the (k-1)-mer (Kmers.cpp:533): every lane at once, windows in lockstep
    u64 mini_hash;
    u32 mini_pos;
    bool reversed;
#ifndef SCAN_MINQUEUE
#define SCAN_MINQUEUE 1   // 0: every expiry is re-scanned (A/B)
#endif
#ifndef PROLOGUE_UNROLL
#define PROLOGUE_UNROLL 2
#endif
#define BRISK_PRAGMA_(x) _Pragma(#x)
#define BRISK_PRAGMA(x) BRISK_PRAGMA_(x)
    const bool mq_on = SCAN_MINQUEUE && (KK ? (KK <= 32) : (k <= 32));  // wave-uniform: the window minimum from a per-lane queue (MinQueue) instead of re-scans
    MinQueue mq{MQ_EMPTY, MQ_EMPTY, MQ_EMPTY, 0u};
    {
        const u32 Km = k - 1 - m;
        const u32 nwin = Km + 1 < nlow1 ? Km + 1 : nlow1;  // windows inside the low 64 bits; those that stick out of them are zero-padded (F2)
        // One pass over the windows, every lane its own (k-1)-mer.  Nothing in the pass branches: a class sum inside the guard band
        // (decy_class_guarded) only raises a flag, and the wave then repeats the pass with the exact fold (practically never).  Ascending
        // windows roll: window i + 1 drops window i's last nucleotide and takes the next one of the low 64 bits (or an A) in front.
        // (Everything the pass works on is local to it and returned by value: flags captured by reference ended up in scratch.)
        struct WinMin {
            u64 best;
            u32 first, last, flags;  // flags: reversed at first | reversed at last << 1 | a sum in the guard band << 2
        };
        auto pass = [&](auto exact_tag) -> WinMin {
            constexpr bool EXACT = decltype(exact_tag)::value;
            u64 best = ~0ull;
            u32 first = 0, last = 0, rf = 0, rl = 0;
            bool guard = false;
            u64 fw = low64 & M, rw = cr, above = low64 >> (2 * m);
            if (mq_on) mq = MinQueue{MQ_EMPTY, MQ_EMPTY, MQ_EMPTY, 0u};
#ifdef SCAN_ATTR_NOPROLOGUE  // attribution builds: wrong results, timing only
            for (u32 ii = 0; ii < 1; ii++) {
#else
            BRISK_PRAGMA(unroll PROLOGUE_UNROLL)
            for (u32 ii = 0; ii < nwin; ii++) {
#endif
                const u32 i = mq_on ? Km - ii : ii;  // (the queue takes the windows oldest first: window i of the (k-1)-mer is the candidate of step -1 - i)
                u64 fwd, rcv;
                if (mq_on) {
                    fwd = (low64 >> (2 * i)) & M;
                    rcv = rc64(fwd, m);
                } else {
                    fwd = fw;
                    rcv = rw;
                    const u32 t = (u32)above & 3u;
                    above >>= 2;
                    fw = (fw >> 2) | ((u64)t << (2 * m - 2));
                    rw = ((rw << 2) & M) | (t ^ 2u);
                }
                const bool rv = rcv < fwd;
                const u64 x = rv ? rcv : fwd;
                u64 key;
                if (EXACT) key = order_key(x, m, M, s_coef);
                else {
#ifdef SCAN_ATTR_NOCLASS
                    key = order_key_fast<NCH, MM>(x, m, M, cfg, s_tabs, s_coef);
#else
                    key = ((u64)decy_class_guarded<NCH, MM>(x, cfg, s_tabs, guard) << 62) + mix2m(x, M);
#endif
                }
                const bool lt = key < best, eq = key == best;
                best = lt ? key : best;
                if (mq_on) {  // (descending i: an equal key lies at a smaller window index)
                    first = lt || eq ? i : first;
                    rf = lt || eq ? (u32)rv : rf;
                    last = lt ? i : last;
                    rl = lt ? (u32)rv : rl;
                } else {
                    first = lt ? i : first;
                    rf = lt ? (u32)rv : rf;
                    last = lt || eq ? i : last;
                    rl = lt || eq ? (u32)rv : rl;
                }
                if (mq_on) mq_push(mq, key, 0u - 1u - i, rv);
            }
            return WinMin{best, first, last, rf | (rl << 1) | (guard ? 4u : 0u)};
        };
#ifdef SCAN_ATTR_PROLOGUE_TWICE  // attribution builds: the pass' cost, results unchanged
        if (__ballot(pass(std::false_type{}).flags & 4u) == 0x123456789ull) return;
        __builtin_amdgcn_s_barrier();
#endif
        WinMin wm = pass(std::false_type{});
        if (__ballot(wm.flags & 4u)) wm = pass(std::true_type{});
        u64 best = wm.best;
        u32 first = wm.first, last = wm.last;
        bool rf = wm.flags & 1u, rl = wm.flags & 2u;
        if (Km >= nwin) {  // windows nlow1..Km of a (k-1)-mer longer than 32: the all-A m-mer (k - 1 > 32 only: never with the queue)
            if (KEY0 < best) {
                best = KEY0;
                first = nwin;
                last = Km;
                rf = rl = false;
            } else if (KEY0 == best) {
                last = Km;
                rl = false;
            }
        }
        u32 pos;
        bool rev, need_canon;
        resolve_ties(first, last, rf, rl, Km, false, false, &pos, &rev, &need_canon);
        if (need_canon && live) {
            const u64 hi = k - 1 > 32 ? load_nts(packed, q0, k - 1 - 32) : 0;
            if (!canonized_as_executed(mk128(low64, hi), k - 1)) rev = false;
        }
        mini_hash = best;
        mini_pos = pos;
        reversed = rev;
        // a tie among the (k-1)-mer's windows: the tie rules may have put the minimizer where the queue's front is not
        if (mq_on && last != first) mq_reset(mq, best, 0u - 1u - pos, rev, true);
    }
    if (seeded) {  // the exact state the previous chunk had before this step
        const ChunkState st = cc.truth[vslot];
        mini_hash = st.hash;
        mini_pos = st.pos_rev & 0x7fffffffu;
        reversed = st.pos_rev >> 31;
        if (mq_on) mq_reset(mq, mini_hash, 0u - 1u - mini_pos, reversed, true);  // (what else is in the window is not part of a chunk's state)
    }
    bool foreign = seeded;  // the vector open at a seeded start began before it: it is the previous chunk's

    // ---- the stream of k-mers.  Per lane: the open vector started at step p0; the minimizer sits mini_pos nts into the
    // current k-mer (counted while the lane is active only).  A vector's idx values follow from these when it closes:
    // the minimizer_idx of k-mer j of the vector is mp0 + j (forward) or w - mp0 - j (reversed), mp0 = mini_pos at p0.
    u32 qcount = 0;  // wave-uniform
    u32 p0 = 0, n_emitted = 0;
    bool dead = false;
    u64 buf = 0;
    const u32 Km = k - m;
    const u32 lane_bits = lane << 18;
#ifdef BRISK_PHASE_PROF
    unsigned long long scan_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (u32 p = 0; p < max_nk; p++) {
        SCNT(0, 1)
        const bool act = p < nk && !dead;
        if (VR && live) {  // the enumerator state before step p, for the chunk-seam check
            if (p == emit_from && !seq_first && !seeded) cc.spec[vslot] = ChunkState{mini_hash, mini_pos | ((reversed ? 1u : 0u) << 31), 1u};
            if (p == emit_until) cc.truth[vslot + 1] = ChunkState{mini_hash, mini_pos | ((reversed ? 1u : 0u) << 31), 1u};
        }
        if ((p & 31) == 0 && p < nk) {
            const u32 left = (u32)(len - (k - 1 + p));
            const u32 cnt = left < 32 ? left : 32;
            buf = load_nts(packed, q0 + k - 1 + p, cnt) << (64 - 2 * cnt);
        }
        const u32 c = (u32)(buf >> 62);
        buf <<= 2;
        cr = (cr >> 2) + ((u64)(c ^ 2u) << (2 * m - 2));
        low64 = (low64 << 2) | c;
        const u64 cf = low64 & M;
        const bool revf = cr < cf;
        const u64 h = order_key_fast<NCH, MM>(revf ? cr : cf, m, M, cfg, s_tabs, s_coef);
        mini_pos += act ? 1u : 0u;
        if (mq_on) mq_push(mq, h, p, revf);
        const bool expired = act && mini_pos > w;                  // Kmers.cpp:551
        const bool newmin = act && !expired && h < mini_hash;      // Kmers.cpp:564
        const bool closed = expired || newmin;
        // the vector closed by this step (Kmers.cpp:585-588); a close at p == 0 is ignored (:590-592)
        bool push = closed && p > 0 && !foreign && p0 >= emit_from && p0 < emit_until;  // a chunk emits the vectors that start in its window
        if (closed) foreign = false;
        if (query_mode && push && n_emitted > 0 && mini_hash == KEY0) {  // counter.cpp:304-306: returned minimizer == 0
            push = false;
            dead = true;
        }
        {
            const unsigned long long bal = __ballot(push);
            if (push) {
                const u32 n = p - p0;
                // minimizer_idx of the LAST element of the returned vector (reversed vectors are returned back to front)
                const u32 idx_end = reversed ? w + n - mini_pos : mini_pos - 1;
                const u32 hi = n | (idx_end << 8) | ((reversed ? 1u : 0u) << 16) | ((mini_hash == KEY0 ? 1u : 0u) << 17) | lane_bits;
                q_ent[qcount + (u32)__popcll(bal & lanes_below(lane))] = ((u64)hi << 32) | p0;
                if (query_mode) n_emitted++;
            }
            qcount += (u32)__popcll(bal);
            SCNT(4, (u32)__popcll(bal))
        }
        // re-scans (get_minimizer on the low 64 bits, Kmers.cpp:367-408): two lanes' k-mers per round, one
        // window per lane of a half-wave.  Window i of a k-mer is (low64 >> 2i) & M -- zero-padded where it
        // sticks out of the low 64 bits (F2); windows 32.. of a k > 32 are the all-A m-mer (KEY0), folded in
        // below without lanes.
        bool served = false;
        if (mq_on && expired && mq.k0 != MQ_EMPTY && ((p - mq.meta) & 0xffu) == mini_pos) {
            // the queue's front is the minimizer that has just left the window: what follows it is what get_minimizer would find --
            // unless its key has an equal in the window (tie rules), or nothing follows (the queue lost track: TRUNC)
            mq_pop_front(mq);
            if (mq.k0 != MQ_EMPTY && !(mq.meta & (1u << 27))) {
                mini_hash = mq.k0;
                mini_pos = (p - mq.meta) & 0xffu;
                reversed = (mq.meta >> 24) & 1u;
                served = true;
            }
        }
        unsigned long long need = __ballot(expired && !dead && !served);
        SCNT(6, (u32)__popcll(__ballot(served)))
        SCNT(1, (u32)__popcll(need))
#ifdef SCAN_ATTR_NORESCAN
        if (expired) {
            mini_pos = 0;
            mini_hash = h;
        }
        need = 0;
#endif
        while (need) {
            const int LA = __ffsll((long long)need) - 1;
            need &= need - 1;
            const bool two = need != 0;
            SCNT(2, 1)
            SCNT(3, two ? 1 : 0)
            int LB = LA;
            if (two) {
                LB = __ffsll((long long)need) - 1;
                need &= need - 1;
            }
            const u64 lowA = read_lane_u64(low64, LA), lowB = read_lane_u64(low64, LB);
            const u32 wl = lane & 31;
            const u64 lowL = lane < 32 ? lowA : lowB;
            const u64 fwd = (lowL >> (2 * wl)) & M;
            const u64 rcv = rc64(fwd, m);
            const bool rv = rcv < fwd;
            u64 key = order_key_fast<NCH, MM>(rv ? rcv : fwd, m, M, cfg, s_tabs, s_coef);
            // an order-preserving 32-bit prefix of the key (class, then the top 30 bits of the mix): the half's minimum goes
            // over the DPP network on it; two windows agreeing on all 32 bits are told apart by the 64-bit path below
            u32 k32 = ((u32)(key >> 62) << 30) | (u32)((key & M) >> ksh);
            if (wl > Km || wl >= nlow) {
                key = ~0ull;
                k32 = ~0u;
            }
            const u32 rm = row_min_u32(k32);
            const u32 r0 = (u32)__builtin_amdgcn_readlane((int)rm, 0), r1 = (u32)__builtin_amdgcn_readlane((int)rm, 16);
            const u32 r2 = (u32)__builtin_amdgcn_readlane((int)rm, 32), r3 = (u32)__builtin_amdgcn_readlane((int)rm, 48);
            const u32 mA = r0 < r1 ? r0 : r1, mB = r2 < r3 ? r2 : r3;
            unsigned long long tie = __ballot(k32 == (lane < 32 ? mA : mB));
            const unsigned long long rvb = __ballot(rv);
            const bool unique = __popc((u32)tie) == 1 && (!two || __popc((u32)(tie >> 32)) == 1);  // wave-uniform
            u64 hm = key;  // slow path only: minimum of this lane's half on the full keys
            if (!unique) {
                for (int o = 16; o > 0; o >>= 1) {
                    const u64 y = __shfl_xor(hm, o, 64);
                    hm = y < hm ? y : hm;
                }
                tie = __ballot(key == hm);
            }
#ifdef SCAN_ATTR_RS_NOEPI  // attribution builds: wrong results, timing only
            if ((int)lane == LA || (two && (int)lane == LB)) {
                mini_hash = key;
                mini_pos = (u32)tie & 31u;
                reversed = rv;
            }
            if (false)
#endif
            for (int half = 0; half < (two ? 2 : 1); half++) {  // wave-uniform, scalar work
                const int L = half ? LB : LA;
                const u32 t = (u32)(tie >> (32 * half)), rb = (u32)(rvb >> (32 * half));
                u32 first = (u32)__ffs((int)t) - 1, last = 31u - (u32)__clz((int)t);
                u64 hmin = unique ? read_lane_u64(key, 32 * half + (int)first) : read_lane_u64(hm, 32 * half);
                bool rf = (rb >> first) & 1, rl = (rb >> last) & 1;
                if (Km >= 32) {  // windows 32..Km: the all-A m-mer
                    if (KEY0s < hmin) {
                        hmin = KEY0s;
                        first = 32;
                        last = Km;
                        rf = rl = false;
                    } else if (KEY0s == hmin) {
                        last = Km;
                        rl = false;
                    }
                }
                u32 pos;
                bool rev, need_canon;
                resolve_ties(first, last, rf, rl, Km, false, false, &pos, &rev, &need_canon);
                if (need_canon) {  // wave-uniform
                    const u64 lowX = half ? lowB : lowA;
                    const u64 qL = read_lane_u64(q0, L) + p;
                    const u64 hi = k > 32 ? load_nts(packed, qL, k - 32) : 0;
                    const u64 lo = k >= 32 ? lowX : (lowX & ((1ull << (2 * k)) - 1));
                    if (!canonized_as_executed(mk128(lo, hi), k)) rev = false;
                }
                if ((int)lane == L) {
                    mini_hash = hmin;
                    mini_pos = pos;
                    reversed = rev;
                    if (mq_on) mq_reset(mq, hmin, p - pos, rev, pos != 0);  // (what lies between the minimizer and now is unknown, unless it IS now)
                }
            }
        }
        if (newmin) {  // Kmers.cpp:572-576
            mini_hash = h;
            mini_pos = 0;
            reversed = revf;
            if (mq_on) mq_reset(mq, h, p, revf, false);  // (the push above has done the same: everything in the window is larger)
        }
        // a new vector begins with this k-mer (Kmers.cpp:578-592; a close at a seeded start is a real one)
        if (closed && (p > 0 || seeded)) p0 = p;
        // turn queued super-k-mers into records with full waves.  All waves append to one record counter, and
        // same-address atomics serialise device-wide (~15 ns each): one reservation per flush, not per record
        if (qcount + 64 > cfg.qcap) {
#ifndef SCAN_ATTR_NOEMIT
            const u32 n_out = queue_records<CLS>(P, q_ent, qcount);
            unsigned long long base = 0;
            if (out.bins) {  // binned records take their slots from the partitions' counters: the record counter is a total, nobody waits for it
                if (lane == 0) atomicAdd(out.n_rec, (unsigned long long)n_out);
            } else {
                if (lane == 0) base = atomicAdd(out.n_rec, (unsigned long long)n_out);
                base = read_lane_u64(base, 0);
            }
            emit_queue<CLS>(P, packed, out, q_ent, s_q0, s_tag, qcount, base);
#endif
            qcount = 0;
        }
    }
    // the last vector of every read (Kmers.cpp:596-601)
    {
        // the sequence's true end closes the open vector; it belongs to the chunk in whose window it started
        bool push = live && !dead && nk > p0 && seq_last && !foreign && p0 >= emit_from && p0 < emit_until;
        if (push && query_mode && n_emitted > 0 && mini_hash == KEY0) push = false;
        const unsigned long long bal = __ballot(push);
        if (push) {
            const u32 n = nk - p0;
            const u32 idx_end = reversed ? w + n - 1 - mini_pos : mini_pos;  // mini_pos: of the read's last k-mer
            const u32 hi = n | (idx_end << 8) | ((reversed ? 1u : 0u) << 16) | ((mini_hash == KEY0 ? 1u : 0u) << 17) | lane_bits;
            q_ent[qcount + (u32)__popcll(bal & lanes_below(lane))] = ((u64)hi << 32) | p0;
        }
        qcount += (u32)__popcll(bal);
    }
#ifdef BRISK_PHASE_PROF
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_scan_cnt[i], scan_cnt[i]);
#endif
#ifdef SCAN_ATTR_NOEMIT
    qcount = 0;
#endif
    scan_final_flush<CLS>(P, packed, out, q_ent, s_q0, s_tag, qcount, s_wcnt, s_wbase);
}
