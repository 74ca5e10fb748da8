// brisk_kernels.hip -- HIP kernels of the Brisk hot path for gfx950 (MI355X).
//
//   k_scan        reads -> super-k-mer records (SuperKmerEnumerator::next, Kmers.cpp:522-603,
//                 + hash_kmer_minimizer_inplace / get_compacted, Kmers.cpp:138-145,191-200)
//   k_scatter     records -> partition (bucket-range) order        [bucket radix, per-partition atomic cursors]
//   k_insert      per partition: find-all, insert-missing, count++  (DenseMenuYo.hpp:248-310, counter.cpp:262-269)
//   k_enumerate   Brisk::next (Brisk.hpp:166-172)
//   k_lookup      Brisk::get (Brisk.hpp:64-69)
//
// All kernels are integer / byte work bound by HBM traffic and VALU issue; there
// is no MFMA-shaped work on this path.
#include "brisk_device.h"

#define SCAN_BLOCK 256
#define EMPTY_SLOT 0xffffffffu
#define MATCHED_BIT 0x80000000u

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
// out[1] = the share of it in reads of more than 1024 k-mers (a record every ~(w+2)/2 k-mers there, while a
// short read makes a few records whatever its length).  One atomic pair per block.
__global__ void __launch_bounds__(256) k_count_kmers(const u64* __restrict__ starts, u64 n_reads, u32 k, unsigned long long* out) {
    __shared__ unsigned long long s_sum[4], s_long[4];
    unsigned long long acc = 0, lng = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        const u64 len = starts[r + 1] - starts[r];
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
};

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

// Build and append the record of one super-k-mer: k-mers at read positions
// [p0, p0+n), vector reversed if `rev` (Kmers.cpp:554-556,597-599); idx_end is
// the minimizer_idx of the LAST element of the returned vector.
__device__ void emit_record_at(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                               u32 idx_end, const ScanOut& out, u32 tag, u64 ret, unsigned long long slot) {
    if (slot >= out.cap) {
        *out.overflow = 1;
        return;
    }
    const u32 L = P.k + n - 1;
    W4 S = load_span(packed, q0 + p0, L);
    if (rev) S = w4_rc(S, L);
    // minimizer of every k-mer of the vector = the m-mer at suffix offset idx_end
    // of the last one (hash_kmer_minimizer_inplace re-extracts it, Kmers.cpp:191-200)
    const u64 mm = w4_shr(S, 2 * idx_end).w0 & P.m_mask;
    const u64 h = mix2m(mm, P.m_mask);
    const u32 bucket = routing_id(P, h);  // Brisk.hpp:135-137, plus the extra routing bits
    // replace the minimizer by its hash (replace_slice, Kmers.cpp:149-159)
    const W4 hole = w4_shl(W4{P.m_mask, 0, 0, 0}, 2 * idx_end);
    S = w4_or(w4_andn(S, hole), w4_shl(W4{h, 0, 0, 0}, 2 * idx_end));
    // drop the b bucket nts at suffix offset idx_end + suff_reduc (get_compacted, Kmers.cpp:138-145)
    const u32 cut = idx_end + P.suff_reduc;
    const W4 lowm = w4_mask(2 * cut);
    const W4 C = w4_or(w4_andn(w4_shr(S, 2 * P.b), lowm), w4_and(S, lowm));

    u64* r = out.rec + slot * P.stride;
    r[0] = C.w0;
    if (P.nw > 1) r[1] = C.w1;
    if (P.nw > 2) r[2] = C.w2;
    if (P.nw > 3) r[3] = C.w3;
    const u32 idx0p = idx_end - (n - 1) + P.suff_reduc;
    r[P.nw] = rec_header(bucket, n, idx0p);
    if (out.tag) out.tag[slot] = tag;
    if (out.ret) out.ret[slot] = ret;
    if (out.hist) atomicAdd(&out.hist[bucket >> P.shift], 1ull | ((unsigned long long)n << 32));
}
__device__ void emit_record(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                            u32 idx_end, const ScanOut& out, u32 tag, u64 ret = 0) {
    emit_record_at(P, packed, q0, p0, n, rev, idx_end, out, tag, ret, atomicAdd(out.n_rec, 1ull));
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
struct ScanCfg {
    u32 nlow;     // nts of a k-mer that get_minimizer sees: min(32, k)   (F2)
    u32 nlow1;    // same for the (k-1)-mer
    u32 nch;      // 4-nt chunks of a decycling sum: ceil((m-1)/4)
    u32 qcap;     // emit queue entries per wave
};

// class from chunk tables: tabs[c][v] (R) and tabs[nch+c][v] (R of the rotation)
template <int NCH>  // NCH > 0: compile-time chunk count (unrolled lookups); 0: runtime nch
__device__ __forceinline__ u32 decy_class_fast(u64 x, u32 m, u32 nch, const double* tabs, const double* coef) {
    double r = 0.0, rr = 0.0;
    if (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            r += tabs[c * 256 + (u32)((x >> (8 * c)) & 255)];
            rr += tabs[(NCH + c) * 256 + (u32)((x >> (8 * c + 2)) & 255)];
        }
    } else {
        for (u32 c = 0; c < nch; c++) {
            r += tabs[c * 256 + (u32)((x >> (8 * c)) & 255)];
            rr += tabs[(nch + c) * 256 + (u32)((x >> (8 * c + 2)) & 255)];
        }
    }
    const double eps = 0.000001, g = 1e-9;
    // any summation order is within ~1e-13 of the reference's fold; inside the guard band redo it exactly
    if (fabs(fabs(r) - eps) < g || fabs(fabs(rr) - eps) < g) return decy_class(x, m, coef);
    if (r > eps) return rr < eps ? 0u : 2u;
    if (r < -eps) return rr > -eps ? 1u : 2u;
    return 2u;
}
template <int NCH = 0>
__device__ __forceinline__ u64 order_key_fast(u64 x, u32 m, u64 M, u32 nch, const double* tabs, const double* coef) {
    return ((u64)decy_class_fast<NCH>(x, m, nch, tabs, coef) << 62) + mix2m(x, M);
}

__global__ void __launch_bounds__(256) k_debug_keys(BriskParams P, u32 nch, const double* __restrict__ g_tabs, const u64* __restrict__ x, u64 n,
                                                    int exact, u64* __restrict__ out) {
    extern __shared__ double smem_d[];
    const u32 n_tab = 128 + 2 * nch * 256;
    for (u32 i = threadIdx.x; i < n_tab; i += blockDim.x) smem_d[i] = g_tabs[i];
    __syncthreads();
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = exact ? order_key(x[i], P.m, P.m_mask, smem_d) : order_key_fast(x[i], P.m, P.m_mask, nch, smem_d + 128, smem_d);
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

// what is left in the waves' queues when their reads end: one slot reservation for the whole block
__device__ __forceinline__ void scan_final_flush(const BriskParams& P, const u32* __restrict__ packed, const ScanOut& out, const u64* q_start,
                                                 const u32* q_misc, const u32* q_tag, u32 qcount, u32* s_wcnt, unsigned long long* s_wbase) {
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) s_wcnt[wid] = qcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (u32 i = 0; i < nw; i++) total += s_wcnt[i];
        *s_wbase = total ? atomicAdd(out.n_rec, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = *s_wbase;
    for (u32 i = 0; i < wid; i++) base += s_wcnt[i];
    for (u32 e = lane; e < qcount; e += 64) {
        const u32 mi = q_misc[e];
        emit_record_at(P, packed, q_start[e], 0, mi & 0xff, (mi >> 16) & 1, (mi >> 8) & 0xff, out, q_tag[e], q_start[e] | ((u64)((mi >> 17) & 1) << 63), base + e);
    }
}

// MODE 0: reads, insert; 1: reads, query (stops a read at a returned minimizer of 0); 2: virtual reads (chunks of long sequences)
// KK, MM: k and m as compile-time constants for the common parameter sets (0: from P) -- folds the shifts and masks and,
// above all, frees scalar registers: the generic kernel spills 70+ of them into vector lanes and pays a v_readlane per use
template <int NCH, int MODE, int KK, int MM>
__global__ void __launch_bounds__(1024) k_scan2(BriskParams P, ScanCfg cfg, const u32* __restrict__ packed, const u64* __restrict__ starts,
                                                u64 n_reads, const double* __restrict__ g_tabs, ScanOut out, ChunkCtl cc) {
    constexpr bool VR = MODE == 2, query_mode = MODE == 1;
    extern __shared__ double smem_d[];
    double* s_coef = smem_d;             // 128
    double* s_tabs = smem_d + 128;       // 2*nch*256
    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 n_tab = 128 + 2 * (NCH ? (u32)NCH : cfg.nch) * 256;
    for (u32 i = tid; i < n_tab; i += blockDim.x) smem_d[i] = g_tabs[i];
    unsigned long long* s_wbase = (unsigned long long*)(smem_d + n_tab);      // block's slot base at the final flush
    u32* s_wcnt = (u32*)(s_wbase + 1);                                         // [16] records left per wave
    u64* q_start = (u64*)(smem_d + n_tab + 9) + (size_t)wid * (2 * cfg.qcap);  // [qcap] stream index of the super-k-mer's first nt
    u32* q_misc = (u32*)(q_start + cfg.qcap);      // [qcap] n | idx_end<<8 | rev<<16
    u32* q_tag = q_misc + cfg.qcap;                // [qcap] read index
    __syncthreads();

    const u32 k = KK ? (u32)KK : P.k, m = MM ? (u32)MM : P.m, w = k - m, nch = NCH ? (u32)NCH : cfg.nch;
    const u64 M = MM ? ((1ull << (2 * MM)) - 1) : P.m_mask;
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
    const bool live = len >= k;  // counter.cpp:233-235
    u32 nk = live ? (u32)(len - k + 1) : 0;
    u32 max_nk = nk;
    for (int o = 32; o > 0; o >>= 1) {
        const u32 y = __shfl_xor(max_nk, o, 64);
        max_nk = y > max_nk ? y : max_nk;
    }
    if (max_nk == 0) {  // nothing to scan in this wave; it still takes part in the block's final reservation
        scan_final_flush(P, packed, out, q_start, q_misc, q_tag, 0, s_wcnt, s_wbase);
        return;
    }
    const u64 KEY0 = order_key_fast<NCH>(0, m, M, nch, s_tabs, s_coef);
    const u64 KEY0s = read_lane_u64(KEY0, 0);  // the same value, in scalar registers

    // ---- prologue: the low 64 bits of the (k-1)-mer; its last m-mer seeds the rolling candidates
    u64 cf = 0, cr = 0, low64 = 0;
    if (live) low64 = load_nts(packed, q0 + (k - 1) - nlow1, nlow1);
    cf = low64 & M;
    cr = rc64(cf, m);

    // ---- minimizer of the (k-1)-mer (Kmers.cpp:533): every lane at once, windows in lockstep
    u64 mini_hash;
    u32 mini_pos;
    bool reversed;
    {
        const u32 Km = k - 1 - m;
        u64 best = ~0ull;
        u32 first = 0, last = 0;
        bool rf = false, rl = false;
        for (u32 i = 0; i <= Km; i++) {
            u64 key;
            bool rv;
            if (i < nlow1) {  // inside the low 64 bits; windows that stick out of them are zero-padded (F2)
                const u64 fwd = (low64 >> (2 * i)) & M;
                const u64 rcv = rc64(fwd, m);
                rv = rcv < fwd;
                key = order_key_fast<NCH>(rv ? rcv : fwd, m, M, nch, s_tabs, s_coef);
            } else {  // beyond the low 64 bits: the all-A m-mer
                key = KEY0;
                rv = false;
            }
            if (key < best) {
                best = key;
                first = last = i;
                rf = rl = rv;
            } else if (key == best) {
                last = i;
                rl = rv;
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
    }
    if (seeded) {  // the exact state the previous chunk had before this step
        const ChunkState st = cc.truth[vslot];
        mini_hash = st.hash;
        mini_pos = st.pos_rev & 0x7fffffffu;
        reversed = st.pos_rev >> 31;
    }
    bool foreign = seeded;  // the vector open at a seeded start began before it: it is the previous chunk's

    // ---- the stream of k-mers
    u32 qcount = 0;  // wave-uniform
    u32 n = 0, p0 = 0, first_idx = 0, last_idx = 0, n_emitted = 0;
    bool dead = false;
    u64 buf = 0;
    const u32 Km = k - m;
    for (u32 p = 0; p < max_nk; p++) {
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
        cf = ((cf << 2) + c) & M;
        cr = (cr >> 2) + ((u64)(c ^ 2u) << (2 * m - 2));
        low64 = (low64 << 2) | c;
        const bool revf = cr < cf;
        const u64 h = order_key_fast<NCH>(revf ? cr : cf, m, M, nch, s_tabs, s_coef);
        mini_pos++;
        const bool expired = act && mini_pos > w;                  // Kmers.cpp:551
        const bool newmin = act && !expired && h < mini_hash;      // Kmers.cpp:564
        const bool closed = expired || newmin;
        // the vector closed by this step (Kmers.cpp:585-588); a close at p == 0 is ignored (:590-592)
        bool push = closed && p > 0 && !foreign && p0 >= emit_from && p0 < emit_until;  // a chunk emits the vectors that start in its window
        if (closed) foreign = false;
        if (push && query_mode && n_emitted > 0 && mini_hash == KEY0) {  // counter.cpp:304-306: returned minimizer == 0
            push = false;
            dead = true;
        }
        {
            const unsigned long long bal = __ballot(push);
            if (push) {
                const u32 at = qcount + (u32)__popcll(bal & lanes_below(lane));
                q_start[at] = q0 + p0;
                q_misc[at] = n | ((reversed ? first_idx : last_idx) << 8) | ((reversed ? 1u : 0u) << 16) | ((mini_hash == KEY0 ? 1u : 0u) << 17);
                q_tag[at] = tagval;
                n_emitted++;
            }
            qcount += (u32)__popcll(bal);
        }
        // re-scans (get_minimizer on the low 64 bits, Kmers.cpp:367-408): two lanes' k-mers per round, one
        // window per lane of a half-wave.  Window i of a k-mer is (low64 >> 2i) & M -- zero-padded where it
        // sticks out of the low 64 bits (F2); windows 32.. of a k > 32 are the all-A m-mer (KEY0), folded in
        // below without lanes.
        unsigned long long need = __ballot(expired && !dead);
        while (need) {
            const int LA = __ffsll((long long)need) - 1;
            need &= need - 1;
            const bool two = need != 0;
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
            u64 key = order_key_fast<NCH>(rv ? rcv : fwd, m, M, nch, s_tabs, s_coef);
            if (wl > Km || wl >= nlow) key = ~0ull;
            u64 hm = key;  // minimum of this lane's half
            for (int o = 16; o > 0; o >>= 1) {
                const u64 y = __shfl_xor(hm, o, 64);
                hm = y < hm ? y : hm;
            }
            const unsigned long long tie = __ballot(key == hm);
            const unsigned long long rvb = __ballot(rv);
            for (int half = 0; half < (two ? 2 : 1); half++) {  // wave-uniform, scalar work
                const int L = half ? LB : LA;
                const u32 t = (u32)(tie >> (32 * half)), rb = (u32)(rvb >> (32 * half));
                u64 hmin = read_lane_u64(hm, 32 * half);
                u32 first = (u32)__ffs((int)t) - 1, last = 31u - (u32)__clz((int)t);
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
                }
            }
        }
        if (newmin) {  // Kmers.cpp:572-576
            mini_hash = h;
            mini_pos = 0;
            reversed = revf;
        }
        if (act) {
            const u32 idx = reversed ? w - mini_pos : mini_pos;  // Kmers.cpp:578-584
            if (closed && (p > 0 || seeded)) n = 0;  // a close at a seeded start is a real one: a new vector begins here
            if (n == 0) {
                p0 = p;
                first_idx = idx;
            }
            last_idx = idx;
            n++;
        }
        // turn queued super-k-mers into records with full waves.  All waves append to one record counter, and
        // same-address atomics serialise device-wide (~15 ns each): one reservation per flush, not per record
        if (qcount + 64 > cfg.qcap) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(out.n_rec, (unsigned long long)qcount);
            base = read_lane_u64(base, 0);
            for (u32 e = lane; e < qcount; e += 64) {
                const u32 mi = q_misc[e];
                emit_record_at(P, packed, q_start[e], 0, mi & 0xff, (mi >> 16) & 1, (mi >> 8) & 0xff, out, q_tag[e], q_start[e] | ((u64)((mi >> 17) & 1) << 63), base + e);
            }
            qcount = 0;
        }
    }
    // the last vector of every read (Kmers.cpp:596-601)
    {
        // the sequence's true end closes the open vector; it belongs to the chunk in whose window it started
        bool push = live && !dead && n > 0 && seq_last && !foreign && p0 >= emit_from && p0 < emit_until;
        if (push && query_mode && n_emitted > 0 && mini_hash == KEY0) push = false;
        const unsigned long long bal = __ballot(push);
        if (push) {
            const u32 at = qcount + (u32)__popcll(bal & lanes_below(lane));
            q_start[at] = q0 + p0;
            q_misc[at] = n | ((reversed ? first_idx : last_idx) << 8) | ((reversed ? 1u : 0u) << 16) | ((mini_hash == KEY0 ? 1u : 0u) << 17);
            q_tag[at] = tagval;
        }
        qcount += (u32)__popcll(bal);
    }
    scan_final_flush(P, packed, out, q_start, q_misc, q_tag, qcount, s_wcnt, s_wbase);
}

// ===========================================================================
// exclusive prefix sum over the low 32 bits of the 64-bit histogram
#define SCAN_ITEMS 16
__global__ void __launch_bounds__(256) k_psum_block(const unsigned long long* __restrict__ hist, u64 n, u32* __restrict__ block_sums) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * 256 * SCAN_ITEMS;
    u32 acc = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + (u64)i * 256 + threadIdx.x;
        if (j < n) acc += (u32)hist[j];
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
                                                    u32* __restrict__ off, u32* __restrict__ cursor) {
    __shared__ u32 s_wave[4];
    const u64 base = (u64)blockIdx.x * 256 * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + i;
        v[i] = j < n ? (u32)hist[j] : 0;
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

// list of partitions with records, ascending inside a block.  One list-cursor atomic per block of 8192
// partitions: every same-address atomic costs ~15 ns device-wide, whoever issues it.
#define TOUCHED_ITEMS 8
__global__ void __launch_bounds__(1024) k_touched(const unsigned long long* __restrict__ hist, u64 n, u32* __restrict__ list, u32* __restrict__ n_list) {
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
        if (mask >> j & 1) list[at++] = (u32)(p0 + j);
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
__global__ void __launch_bounds__(256) k_need(const unsigned long long* __restrict__ hist, const u32* __restrict__ part_off,
                                              const u32* __restrict__ list, u32 n_list, const DirEnt* __restrict__ dir,
                                              PartDesc* __restrict__ desc, unsigned long long* out) {
    __shared__ unsigned long long s_sum[4];
    unsigned long long need = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {  // grid-stride: few blocks, few atomics
        const u32 p = list[i];
        PartDesc d;
        d.part = p;
        d.r_begin = part_off[p];
        d.n_rec = part_off[p + 1] - d.r_begin;
        d.n_inst = (u32)(hist[p] >> 32);
        const DirEnt de = dir[p];
        d.n_exist = de.cnt;
        d.cap = de.cap;
        d.off = de.off;
        desc[i] = d;
        const u32 tot = d.n_exist + d.n_inst;
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
__device__ __forceinline__ u32 owner_of_record(const BriskParams& P, u64 hdr) {
    return (u32)(((u64)(hdr_bucket(hdr) >> P.shift) * P.n_owners) >> P.part_bits);
}
__global__ void __launch_bounds__(256) k_owner_hist(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u64 chunk, u32* __restrict__ block_cnt,
                                                    unsigned long long* __restrict__ hist) {
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
        const u32 owner = ok ? owner_of_record(P, hdr) : 0xffffffffu;
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
                                                       const u32* __restrict__ tag_in, u32* __restrict__ tag_out) {
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
        const u32 owner = ok ? owner_of_record(P, hdr) : 0xffffffffu;
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
__global__ void __launch_bounds__(256) k_scatter(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u32* __restrict__ cursor,
                                                 u64* __restrict__ out, int by_owner, const u32* __restrict__ tag_in, u32* __restrict__ tag_out,
                                                 u32* __restrict__ err) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u64* src = rec + i * P.stride;
    const u64 hdr = src[P.nw];
    u32 bin = hdr_bucket(hdr) >> P.shift;
    if (by_owner) bin = (u32)(((u64)bin * P.n_owners) >> P.part_bits);
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
#define INSERT_SLOTS 4096u   // persistent waves == private allocator slots (256 CUs x 4 SIMDs x 4 waves)
#endif
#ifndef WI_WAVES_PER_EU
#define WI_WAVES_PER_EU 4
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
};

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

// Record-level de-duplication of the <= 64 records the lanes hold (also in s_rec): the first copy of every distinct record
// survives, s_rmult[its lane] = the multiplicities of all its copies added up; returns whether this lane's record is a
// later copy.  Header bits 48..55 carry a record's multiplicity (mod 256: counts wrap there anyway) once a partition's
// records have been collapsed; they are not part of its identity.
#define HDR_ID_MASK 0x0000ffffffffffffull
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
template <u32 MAXI>
__device__ __forceinline__ void insert_body(const BriskParams& P, u64* __restrict__ rec, const PartDesc* __restrict__ desc, u32 n_touched, const IndexDev& ix,
                                            u32* __restrict__ work_counter) {
    constexpr u32 TABLE = 2 * MAXI, TS = TABLE / 64, NI = MAXI / 64;
    static_assert(MAXI % 256 == 0 && MAXI <= 1024, "chunk size: whole 32-bit words of record marks per lane, 10-bit instance index");
    __shared__ u64 s_key[2 * MAXI];
    __shared__ u64 s_rec[WI_MAX_REC * 5 > MAXI / 2 ? WI_MAX_REC * 5 : MAXI / 2];
    __shared__ u32 s_tab[TABLE];
    __shared__ u32 s_pref[WI_MAX_REC + 1];
    u32* s_list = (u32*)s_rec;  // [MAXI] the new entries' table words: built after the records have been expanded
    __shared__ u32 s_rtab[2 * WI_MAX_REC];
    __shared__ u32 s_rmult[WI_MAX_REC];
    __shared__ __attribute__((aligned(4))) uint8_t s_irec[MAXI];
    __shared__ u32 s_bm[2];

    const u32 lane = threadIdx.x;
    unsigned long long acur = ix.slot_cur[blockIdx.x], aend = ix.slot_end[blockIdx.x], garbage = 0;
    const u32 kbits = 2 * P.kb + 6;

    for (;;) {
        // ---- take the next batch of partitions (one atomic per WI_BATCH partitions)
        u32 t0 = 0;
        if (lane == 0) t0 = atomicAdd(work_counter, WI_BATCH);
        t0 = __shfl(t0, 0, 64);
        if (t0 >= n_touched) break;
        const u32 t_end = min(t0 + WI_BATCH, n_touched);
        PartDesc d = desc[t0];
        RecRegs rr = load_rec_regs(P, rec, d.r_begin, min(d.n_rec, (u32)WI_MAX_REC), lane);

        for (u32 t = t0; t < t_end; t++) {
            // descriptor of the partition after this one: in flight while this one is processed
            const u32 tn = t + 1;
            PartDesc dn{};
            if (tn < t_end) dn = desc[tn];
            RecRegs rn{0, 0, 0, 0, 0};

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
            if (MAXI > WI_MAX_INST && d.n_rec > 2 * WI_MAX_REC) {
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
                    const bool dup = dedupe_records(P.stride, rr, 1u, avail, lane, s_rec, s_rtab, s_rmult);
                    wave_sync();
                    const bool keep = lane < avail && !dup;
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {  // survivors move to the front of the partition's records (never past what is still to be read)
                        u64* dst = rec + (u64)(wr + (u32)__popcll(bal & lanes_below(lane))) * P.stride;
                        const u64 mult = (u64)(s_rmult[lane] & 0xffu) << 48;
                        dst[0] = rr.w0;
                        dst[1] = P.stride == 2 ? rr.w1 | mult : rr.w1;
                        if (P.stride > 2) dst[2] = P.stride == 3 ? rr.w2 | mult : rr.w2;
                        if (P.stride > 3) dst[3] = P.stride == 4 ? rr.w3 | mult : rr.w3;
                        if (P.stride > 4) dst[4] = rr.w4 | mult;
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
                const u32 avail = min(r_end - rc, (u32)WI_MAX_REC);
                if (rc != d.r_begin) rr = load_rec_regs(P, rec, rc, avail, lane);
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
                const u32 my_mult = collapsed ? (u32)(my_hdr >> 48) & 0xffu : 1u;
                const u32 x0 = wave_incl_scan(raw_n, lane);
                // First try every available record: identical records (the same super-k-mer seen in
                // several reads) collapse into one with a multiplicity, so far more raw instances fit.
                // If the collapsed chunk is still too big, shrink to the prefix whose collapsed count fits (counted
                // again on its own it can come out a little higher, once the first copy of a record lies beyond
                // it: hence the loop), at the latest to the raw-count prefix, which always fits.
                const u32 rawfit = (u32)__popcll(__ballot(lane < avail && x0 <= MAXI));  // >= 1; a prefix: x0 is monotone
                u32 nrec = avail, my_n = 0, x = 0, ninst = 0;
                for (int attempt = 0;; attempt++) {
                    const bool dup = dedupe_records(P.stride, rr, my_mult, nrec, lane, s_rec, s_rtab, s_rmult);
                    my_n = (lane < nrec && !dup) ? raw_n : 0;
                    x = wave_incl_scan(my_n, lane);
                    ninst = __shfl(x, 63, 64);
                    if (ninst <= MAXI) break;
                    const u32 fit = (u32)__popcll(__ballot(lane < nrec && x <= MAXI));  // x is monotone too
                    nrec = (attempt >= 2 || fit <= rawfit) ? rawfit : min(fit, nrec - 1);
                    wave_sync();
                }
                const u32 raw_inst = __shfl(x0, nrec - 1, 64);
                u32 tsize = 128;
                while (tsize < 2 * ninst && tsize < TABLE) tsize <<= 1;
                s_pref[lane + 1] = x;
                if (lane == 0) s_pref[0] = 0;
#pragma unroll
                for (u32 w = 0; w < TS; w++)
                    if (w * 64 < tsize) s_tab[w * 64 + lane] = EMPTY_SLOT;
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
                // the next partition's first records: requested now, consumed next iteration
                if (rc == d.r_begin && tn < t_end) rn = load_rec_regs(P, rec, dn.r_begin, min(dn.n_rec, (u32)WI_MAX_REC), lane);

                // ---- 0/1. expand to entry keys and de-duplicate
                if (P.nw == 3) {  // k63/m21/b14 and neighbours: compile-time record width
                    if (ninst <= 64) expand_and_dedupe<1, 3>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else if (ninst <= 128) expand_and_dedupe<2, 3>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else if (ninst <= 192) expand_and_dedupe<3, 3>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else if (NI <= 4 || ninst <= 256) expand_and_dedupe<4, 3>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else expand_and_dedupe<NI, 3>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                } else {
                    if (ninst <= 128) expand_and_dedupe<2, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else if (NI <= 4 || ninst <= 256) expand_and_dedupe<4, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                    else expand_and_dedupe<NI, 0>(P, lane, ninst, tsize, s_rec, s_pref, s_irec, s_rmult, s_key, s_tab);
                }
                wave_sync();

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
                        klo[q] = ix.keys[2 * at];
                        khi[q] = ix.keys[2 * at + 1];
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

                // ---- 3. append the unmatched ones: compact them in LDS, then write them out with
                // full waves (a store instruction costs the same with 3 active lanes as with 64)
                u32 n_new = 0;
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
                wave_sync();
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
                        ix.keys[2 * (noff + e)] = ix.keys[2 * (off + e)];
                        ix.keys[2 * (noff + e) + 1] = ix.keys[2 * (off + e) + 1];
                        ix.counts[noff + e] = ix.counts[off + e];
                    }
                    off = noff;
                }
                for (u32 q = lane; q < n_new; q += 64) {
                    const u32 v = s_list[q];
                    const u32 i = v & WI_IDX_MASK;
                    const u64 klo2 = s_key[2 * i], khi2 = s_key[2 * i + 1];
                    const unsigned long long at = off + n_exist + q;
                    ix.keys[2 * at] = klo2;
                    ix.keys[2 * at + 1] = khi2;
                    ix.counts[at] = (uint8_t)(v >> WI_CNT_SHIFT);
                    // bucket id inside the partition: the key's top `shift` bits (<= 6 of them used here)
                    const u32 bl = P.shift ? ((u32)shr128(mk128(klo2, khi2), kbits).lo & ((1u << P.shift) - 1)) : 0;
                    const u32 bb = P.shift > 6 ? (bl >> (P.shift - 6)) : bl;  // 64 bins at most
                    if (bb < 32) bm0 |= 1u << bb; else bm1 |= 1u << (bb - 32);
                }
                n_exist += n_new;
                rc += nrec;

            }
            if (P.shift <= 6) {  // OR the lanes' bucket bits together through LDS
                if (lane < 2) s_bm[lane] = 0;
                wave_sync();
                if (bm0) atomicOr(&s_bm[0], bm0);
                if (bm1) atomicOr(&s_bm[1], bm1);
                wave_sync();
                bm0 = s_bm[0];
                bm1 = s_bm[1];
            }
            if (lane == 0) ix.dir[part] = DirEnt{off, n_exist, cap};
            // bucket occupancy bits: exact when a partition holds <= 64 buckets (shift <= 6);
            // partitions of more buckets are handled by k_bucket_bits below
            if (P.shift <= 6 && lane < 2) {
                const u32 mask = lane == 0 ? bm0 : bm1;
                const u32 nb = 1u << P.shift;  // buckets per partition
                const u64 first = ((u64)part << P.shift) >> P.ext_bits;  // ext_bits > 0 => shift == 0: the one bucket this partition is a slice of
                // a bit that is already set needs no atomic: with few buckets (small b) every partition of a bucket
                // would otherwise hit the same word, and same-address atomics serialise device-wide
                if (mask) {
                    if (nb >= 32) {
                        if (lane * 32 < nb && (ix.bucket_bits[(first >> 5) + lane] & mask) != mask) atomicOr(&ix.bucket_bits[(first >> 5) + lane], mask);
                    } else if (lane == 0) {
                        const u32 bits = mask << (first & 31);
                        if ((ix.bucket_bits[first >> 5] & bits) != bits) atomicOr(&ix.bucket_bits[first >> 5], bits);
                    }
                }
            }
            d = dn;
            rr = rn;
        }
    }
    if (lane == 0) {
        ix.slot_cur[blockIdx.x] = acur;
        ix.slot_end[blockIdx.x] = aend;
        if (garbage) atomicAdd(&ix.stats[3], garbage);
    }
}


__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WI_WAVES_PER_EU, 8))) k_insert(BriskParams P, u64* __restrict__ rec, const PartDesc* __restrict__ desc,
                                               u32 n_touched, IndexDev ix, u32* __restrict__ work_counter) {
    insert_body<WI_MAX_INST>(P, rec, desc, n_touched, ix, work_counter);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) k_insert_big(BriskParams P, u64* __restrict__ rec, const PartDesc* __restrict__ desc,
                                                                                         u32 n_touched, IndexDev ix, u32* __restrict__ work_counter) {
    insert_body<2 * WI_MAX_INST>(P, rec, desc, n_touched, ix, work_counter);
}

// bucket occupancy for partitions wider than 64 buckets (small part_bits): one pass over all entries
__global__ void __launch_bounds__(256) k_bucket_bits(BriskParams P, IndexDev ix, u32 n_parts) {
    const u32 kbits = 2 * P.kb + 6;
    for (u32 part = blockIdx.x; part < n_parts; part += gridDim.x) {
        const u32 cnt = ix.dir[part].cnt;
        const unsigned long long off = ix.dir[part].off;
        for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
            const u128x key = mk128(ix.keys[2 * (off + e)], ix.keys[2 * (off + e) + 1]);
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
                    const u128x key = mk128(ix.keys[2 * (d.off + e)], ix.keys[2 * (d.off + e) + 1]);
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
            const u128x key = mk128(ix.keys[2 * (de.off + e)], ix.keys[2 * (de.off + e) + 1]);
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
    const u32 pi = blockIdx.x;
    if (pi >= n_parts) return;
    const u32 part = p_begin + pi;
    const u32 cnt = ix.dir[part].cnt;
    const unsigned long long off = ix.dir[part].off;
    const u64 ob = out_base[pi];
    for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
        const u128x key = mk128(ix.keys[2 * (off + e)], ix.keys[2 * (off + e) + 1]);
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
        const u32 bucket = routing_id(P, h);
        km = or128(andn128(km, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(h, 0), 2 * idx));
        const u32 cut = idx + P.suff_reduc;
        const u128x lowm = mask128(2 * cut);
        const u128x comp = or128(andn128(shr128(km, 2 * P.b), lowm), and128(km, lowm));
        const u128x key = make_key(P, bucket, and128(comp, mask128(2 * P.kb)), cut);
        const u32 part = bucket >> P.shift;
        const u32 cnt = ix.dir[part].cnt;
        const unsigned long long off = ix.dir[part].off;
        for (u32 e = lane; e < cnt && !found; e += 64) {
            if (ix.keys[2 * (off + e)] == key.lo && ix.keys[2 * (off + e) + 1] == key.hi) {
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
    const u32 bucket = routing_id(P, h);
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
                                               unsigned long long* __restrict__ id_counter, u32* __restrict__ n_done) {
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
            if (ix.keys[2 * (de.off + e)] == key.lo && ix.keys[2 * (de.off + e) + 1] == key.hi) {
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
                    ix.keys[2 * (noff + e)] = ix.keys[2 * (de.off + e)];
                    ix.keys[2 * (noff + e) + 1] = ix.keys[2 * (de.off + e) + 1];
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
                ix.keys[2 * at] = key.lo;
                ix.keys[2 * at + 1] = key.hi;
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
    if (lane == 0) *n_done = done;
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
