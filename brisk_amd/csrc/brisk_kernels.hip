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
#define INSERT_BLOCK 256
#define INS_TABLE 2048      // LDS hash slots per partition chunk
#define INS_MAX_INST 1280   // k-mer instances per chunk (load <= 0.625)
#define INS_MAX_REC 256     // records per chunk
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

// sum over reads of max(0, len-k+1): the number of k-mer instances (an upper
// bound on records).  One atomic per block.
__global__ void __launch_bounds__(256) k_count_kmers(const u64* __restrict__ starts, u64 n_reads, u32 k, unsigned long long* out) {
    __shared__ unsigned long long s_sum[4];
    unsigned long long acc = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        const u64 len = starts[r + 1] - starts[r];
        if (len >= k) acc += len - k + 1;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
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
    u32* tag;                    // query mode: read index per record (may be null)
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
__device__ void emit_record(const BriskParams& P, const u32* __restrict__ packed, u64 q0, u32 p0, u32 n, bool rev,
                            u32 idx_end, const ScanOut& out, u32 tag) {
    const u32 L = P.k + n - 1;
    W4 S = load_span(packed, q0 + p0, L);
    if (rev) S = w4_rc(S, L);
    // minimizer of every k-mer of the vector = the m-mer at suffix offset idx_end
    // of the last one (hash_kmer_minimizer_inplace re-extracts it, Kmers.cpp:191-200)
    const u64 mm = w4_shr(S, 2 * idx_end).w0 & P.m_mask;
    const u64 h = mix2m(mm, P.m_mask);
    const u32 bucket = (u32)((h >> (2 * P.suff_reduc)) & P.bucket_mask);  // Brisk.hpp:135-137
    // replace the minimizer by its hash (replace_slice, Kmers.cpp:149-159)
    const W4 hole = w4_shl(W4{P.m_mask, 0, 0, 0}, 2 * idx_end);
    S = w4_or(w4_andn(S, hole), w4_shl(W4{h, 0, 0, 0}, 2 * idx_end));
    // drop the b bucket nts at suffix offset idx_end + suff_reduc (get_compacted, Kmers.cpp:138-145)
    const u32 cut = idx_end + P.suff_reduc;
    const W4 lowm = w4_mask(2 * cut);
    const W4 C = w4_or(w4_andn(w4_shr(S, 2 * P.b), lowm), w4_and(S, lowm));

    const unsigned long long slot = atomicAdd(out.n_rec, 1ull);
    if (slot >= out.cap) {
        *out.overflow = 1;
        return;
    }
    u64* r = out.rec + slot * P.stride;
    r[0] = C.w0;
    if (P.nw > 1) r[1] = C.w1;
    if (P.nw > 2) r[2] = C.w2;
    if (P.nw > 3) r[3] = C.w3;
    const u32 idx0p = idx_end - (n - 1) + P.suff_reduc;
    r[P.nw] = rec_header(bucket, n, idx0p);
    if (out.tag) out.tag[slot] = tag;
    if (out.hist) atomicAdd(&out.hist[bucket >> P.shift], 1ull | ((unsigned long long)n << 32));
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
            emit_record(P, packed, q0, p0, n, old_rev, old_rev ? first_idx : last_idx, out, (u32)r);
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
        emit_record(P, packed, q0, p0, n, reversed, reversed ? first_idx : last_idx, out, (u32)r);
    }
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

// list of partitions with records; one wave-aggregated atomic per wave
__global__ void __launch_bounds__(256) k_touched(const unsigned long long* __restrict__ hist, u64 n, u32* __restrict__ list, u32* __restrict__ n_list) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = p < n && (u32)hist[p] != 0;
    const unsigned long long bal = __ballot(hit);
    if (!bal) return;
    const u32 lane = threadIdx.x & 63;
    u32 base = 0;
    if (lane == (u32)__ffsll((long long)bal) - 1) base = atomicAdd(n_list, (u32)__popcll(bal));
    base = __shfl(base, __ffsll((long long)bal) - 1, 64);
    if (hit) list[base + __popcll(bal & ((1ull << lane) - 1))] = (u32)p;
}

// arena space the insert of this batch may need (all instances new), per touched partition
__device__ __forceinline__ u32 grow_cap(u32 n) { return n + (n >> 2) + 8; }
__global__ void __launch_bounds__(256) k_need(const unsigned long long* __restrict__ hist, const u32* __restrict__ list, u32 n_list,
                                              const u32* __restrict__ dir_cnt, const u32* __restrict__ dir_cap, unsigned long long* out) {
    __shared__ unsigned long long s_sum[4];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long need = 0;
    if (i < n_list) {
        const u32 p = list[i];
        const u32 tot = dir_cnt[p] + (u32)(hist[p] >> 32);
        if (tot > dir_cap[p]) need = grow_cap(tot);
    }
    for (int o = 32; o > 0; o >>= 1) need += __shfl_down(need, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = need;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// ===========================================================================
// k_scatter: bucket radix -- move each record to its partition's slice.
// owner mode (n_owners > 1 and by_owner): the bins are owners instead of partitions.
__global__ void __launch_bounds__(256) k_owner_hist(BriskParams P, const u64* __restrict__ rec, u64 n_rec, unsigned long long* __restrict__ hist) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u64 hdr = rec[i * P.stride + P.nw];
    const u32 part = hdr_bucket(hdr) >> P.shift;
    const u32 owner = (u32)(((u64)part * P.n_owners) >> P.part_bits);
    atomicAdd(&hist[owner], 1ull | ((unsigned long long)hdr_n(hdr) << 32));
}
__global__ void __launch_bounds__(256) k_part_hist(BriskParams P, const u64* __restrict__ rec, u64 n_rec, unsigned long long* __restrict__ hist) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u64 hdr = rec[i * P.stride + P.nw];
    atomicAdd(&hist[hdr_bucket(hdr) >> P.shift], 1ull | ((unsigned long long)hdr_n(hdr) << 32));
}
__global__ void __launch_bounds__(256) k_scatter(BriskParams P, const u64* __restrict__ rec, u64 n_rec, u32* __restrict__ cursor,
                                                 u64* __restrict__ out, int by_owner, const u32* __restrict__ tag_in, u32* __restrict__ tag_out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const u64* src = rec + i * P.stride;
    const u64 hdr = src[P.nw];
    u32 bin = hdr_bucket(hdr) >> P.shift;
    if (by_owner) bin = (u32)(((u64)bin * P.n_owners) >> P.part_bits);
    const u32 slot = atomicAdd(&cursor[bin], 1u);
    u64* dst = out + (u64)slot * P.stride;
    for (u32 j = 0; j < P.stride; j++) dst[j] = src[j];
    if (tag_in) tag_out[slot] = tag_in[i];
}

// ===========================================================================
// k_insert: one workgroup per touched partition.
//   1. the chunk's k-mer instances are de-duplicated in an LDS table whose slots
//      hold (record, j) of the first instance and a multiplicity;
//   2. the partition's existing entries stream through the table: a hit adds the
//      multiplicity to the entry's count (uint8_t, wraps; counter.cpp:264-268);
//   3. unmatched table entries are appended as new entries (count = multiplicity).
// Storage per partition: keys[] (u128) and counts[] (u8) in a bump-allocated arena;
// a partition that outgrows its slice moves to a fresh one.
struct IndexDev {
    u64* keys;                   // 2 u64 per entry
    uint8_t* counts;
    unsigned long long* dir_off; // per partition: first entry
    u32* dir_cnt;
    u32* dir_cap;
    unsigned long long* cursor;  // arena entries in use
    u32* bucket_bits;            // one bit per bucket id
    unsigned long long* stats;   // [0] nb_kmers [1] nb_buckets [2] largest partition [3] garbage entries
};

__device__ __forceinline__ u128x chunk_key(const BriskParams& P, const u64* s_rec, u32 rec, u32 j) {
    const u64* c = s_rec + rec * P.stride;
    const u64 hdr = c[P.nw];
    return make_key(P, hdr_bucket(hdr), record_kmer(P, c, hdr_n(hdr), j), hdr_idx0(hdr) + j);
}

__global__ void __launch_bounds__(INSERT_BLOCK) k_insert(BriskParams P, const u64* __restrict__ rec, const u32* __restrict__ part_off,
                                                         const unsigned long long* __restrict__ hist,
                                                         const u32* __restrict__ touched, IndexDev ix) {
    __shared__ u64 s_rec[INS_MAX_REC * 5];
    __shared__ u32 s_tab[INS_TABLE];
    __shared__ u32 s_cnt[INS_TABLE];
    __shared__ u32 s_pref[INS_MAX_REC + 1];
    __shared__ u32 s_wave[INSERT_BLOCK / 64];
    __shared__ u32 s_nrec, s_ninst, s_nnew;
    __shared__ unsigned long long s_off;
    __shared__ u32 s_cap;

    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 part = touched[blockIdx.x];
    const u32 r_begin = part_off[part], r_end = part_off[part + 1];
    u32 n_exist = ix.dir_cnt[part];
    const u32 n_exist0 = n_exist;
    u32 inst_left = (u32)(hist[part] >> 32);  // instances not yet processed: bounds the final size
    if (tid == 0) {
        s_off = ix.dir_off[part];
        s_cap = ix.dir_cap[part];
    }
    __syncthreads();

    for (u32 rc = r_begin; rc < r_end;) {
        // ---- pick the chunk: up to INS_MAX_REC records / INS_MAX_INST instances
        const u32 avail = min(r_end - rc, (u32)INS_MAX_REC);
        u32 my_n = 0;
        if (tid < avail) my_n = hdr_n(rec[(u64)(rc + tid) * P.stride + P.nw]);
        u32 x = my_n;  // inclusive prefix over the block
        for (int o = 1; o < 64; o <<= 1) {
            const u32 y = __shfl_up(x, o, 64);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) s_wave[wid] = x;
        __syncthreads();
        u32 woff = 0;
        for (u32 j = 0; j < wid; j++) woff += s_wave[j];
        x += woff;
        if (tid < INS_MAX_REC) s_pref[tid + 1] = x;
        if (tid == 0) s_pref[0] = 0;
        __syncthreads();
        if (tid == 0) {
            // largest prefix of records whose instances fit (at least one: n <= k-m+1 < INS_MAX_INST)
            u32 lo = 1, hi = avail;
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (s_pref[mid] <= INS_MAX_INST) lo = mid; else hi = mid - 1;
            }
            s_nrec = lo;
            s_ninst = s_pref[lo];
            s_nnew = 0;
        }
        for (u32 i = tid; i < INS_TABLE; i += INSERT_BLOCK) {
            s_tab[i] = EMPTY_SLOT;
            s_cnt[i] = 0;
        }
        __syncthreads();
        const u32 nrec = s_nrec, ninst = s_ninst;
        for (u32 i = tid; i < nrec * P.stride; i += INSERT_BLOCK) s_rec[i] = rec[(u64)rc * P.stride + i];
        __syncthreads();

        // ---- 1. de-duplicate the chunk's instances
        for (u32 i = tid; i < ninst; i += INSERT_BLOCK) {
            u32 lo = 0, hi = nrec - 1;  // record of instance i: last r with s_pref[r] <= i
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (s_pref[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const u32 j = i - s_pref[lo];
            const u128x key = chunk_key(P, s_rec, lo, j);
            const u32 me = (lo << 6) | j;
            u32 h = hash_key32(key) & (INS_TABLE - 1);
            for (;;) {
                const u32 old = atomicCAS(&s_tab[h], EMPTY_SLOT, me);
                if (old == EMPTY_SLOT || eq128(chunk_key(P, s_rec, old >> 6, old & 63), key)) {
                    atomicAdd(&s_cnt[h], 1u);
                    break;
                }
                h = (h + 1) & (INS_TABLE - 1);
            }
        }
        __syncthreads();

        // ---- 2. existing entries probe the table
        const unsigned long long off = s_off;
        for (u32 e = tid; e < n_exist; e += INSERT_BLOCK) {
            const u128x key = mk128(ix.keys[2 * (off + e)], ix.keys[2 * (off + e) + 1]);
            u32 h = hash_key32(key) & (INS_TABLE - 1);
            for (;;) {
                const u32 v = s_tab[h];
                if (v == EMPTY_SLOT) break;
                if (eq128(chunk_key(P, s_rec, (v & ~MATCHED_BIT) >> 6, v & 63), key)) {
                    ix.counts[off + e] = (uint8_t)(ix.counts[off + e] + s_cnt[h]);
                    s_tab[h] = v | MATCHED_BIT;
                    break;
                }
                h = (h + 1) & (INS_TABLE - 1);
            }
        }
        __syncthreads();

        // ---- 3. append the unmatched ones
        u32 mine = 0;
        for (u32 i = tid; i < INS_TABLE; i += INSERT_BLOCK) {
            const u32 v = s_tab[i];
            mine += (v != EMPTY_SLOT && !(v & MATCHED_BIT));
        }
        u32 px = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 y = __shfl_up(px, o, 64);
            if ((int)lane >= o) px += y;
        }
        if (lane == 63) s_wave[wid] = px;
        __syncthreads();
        u32 before = px - mine;
        for (u32 j = 0; j < wid; j++) before += s_wave[j];
        const u32 n_new = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        inst_left -= ninst;
        if (n_exist + n_new > s_cap) {
            // move to a fresh slice, sized so that this partition moves at most once per batch
            __syncthreads();
            if (tid == 0) {
                const u32 want = grow_cap(n_exist + n_new + inst_left);
                const unsigned long long noff = atomicAdd(ix.cursor, (unsigned long long)want);
                atomicAdd(&ix.stats[3], (unsigned long long)s_cap);
                s_off = noff;
                s_cap = want;
            }
            __syncthreads();
            const unsigned long long noff = s_off;
            for (u32 e = tid; e < n_exist; e += INSERT_BLOCK) {
                ix.keys[2 * (noff + e)] = ix.keys[2 * (off + e)];
                ix.keys[2 * (noff + e) + 1] = ix.keys[2 * (off + e) + 1];
                ix.counts[noff + e] = ix.counts[off + e];
            }
        }
        const unsigned long long woffs = s_off;
        u32 pos = n_exist + before;
        for (u32 i = tid; i < INS_TABLE; i += INSERT_BLOCK) {
            const u32 v = s_tab[i];
            if (v != EMPTY_SLOT && !(v & MATCHED_BIT)) {
                const u64 hdr = s_rec[(v >> 6) * P.stride + P.nw];
                const u128x key = chunk_key(P, s_rec, v >> 6, v & 63);
                ix.keys[2 * (woffs + pos)] = key.lo;
                ix.keys[2 * (woffs + pos) + 1] = key.hi;
                ix.counts[woffs + pos] = (uint8_t)s_cnt[i];
                const u32 bucket = hdr_bucket(hdr);
                const u32 bit = 1u << (bucket & 31);
                if (!(ix.bucket_bits[bucket >> 5] & bit)) {
                    const u32 prev = atomicOr(&ix.bucket_bits[bucket >> 5], bit);
                    if (!(prev & bit)) atomicAdd(&ix.stats[1], 1ull);
                }
                pos++;
            }
        }
        n_exist += n_new;
        rc += nrec;
        if (rc < r_end) __threadfence();  // the next chunk re-reads what this one appended
        __syncthreads();
    }
    if (tid == 0) {
        ix.dir_off[part] = s_off;
        ix.dir_cnt[part] = n_exist;
        ix.dir_cap[part] = s_cap;
        if (n_exist != n_exist0) atomicAdd(&ix.stats[0], (unsigned long long)(n_exist - n_exist0));
        atomicMax(&ix.stats[2], (unsigned long long)n_exist);
    }
}

// ===========================================================================
// k_query: same table as k_insert, but existing entries add their count to the
// per-read sums of the instances they match (get_superkmer, Brisk.hpp:102-118).
__global__ void __launch_bounds__(INSERT_BLOCK) k_query(BriskParams P, const u64* __restrict__ rec, const u32* __restrict__ tags,
                                                        const u32* __restrict__ part_off, const u32* __restrict__ touched,
                                                        IndexDev ix, unsigned long long* __restrict__ per_read_sum) {
    __shared__ u64 s_rec[INS_MAX_REC * 5];
    __shared__ u32 s_tab[INS_TABLE];
    __shared__ u32 s_pref[INS_MAX_REC + 1];
    __shared__ u32 s_wave[INSERT_BLOCK / 64];
    __shared__ u32 s_nrec, s_ninst;

    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 part = touched[blockIdx.x];
    const u32 r_begin = part_off[part], r_end = part_off[part + 1];
    const u32 n_exist = ix.dir_cnt[part];
    const unsigned long long off = ix.dir_off[part];
    if (n_exist == 0) return;

    for (u32 rc = r_begin; rc < r_end;) {
        const u32 avail = min(r_end - rc, (u32)INS_MAX_REC);
        u32 my_n = 0;
        if (tid < avail) my_n = hdr_n(rec[(u64)(rc + tid) * P.stride + P.nw]);
        u32 x = my_n;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 y = __shfl_up(x, o, 64);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) s_wave[wid] = x;
        __syncthreads();
        u32 woff = 0;
        for (u32 j = 0; j < wid; j++) woff += s_wave[j];
        x += woff;
        if (tid < INS_MAX_REC) s_pref[tid + 1] = x;
        if (tid == 0) s_pref[0] = 0;
        __syncthreads();
        if (tid == 0) {
            u32 lo = 1, hi = avail;
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (s_pref[mid] <= INS_MAX_INST) lo = mid; else hi = mid - 1;
            }
            s_nrec = lo;
            s_ninst = s_pref[lo];
        }
        for (u32 i = tid; i < INS_TABLE; i += INSERT_BLOCK) s_tab[i] = EMPTY_SLOT;
        __syncthreads();
        const u32 nrec = s_nrec, ninst = s_ninst;
        for (u32 i = tid; i < nrec * P.stride; i += INSERT_BLOCK) s_rec[i] = rec[(u64)rc * P.stride + i];
        __syncthreads();
        // every instance gets its own slot (duplicates chain behind each other)
        for (u32 i = tid; i < ninst; i += INSERT_BLOCK) {
            u32 lo = 0, hi = nrec - 1;
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (s_pref[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const u32 j = i - s_pref[lo];
            const u128x key = chunk_key(P, s_rec, lo, j);
            u32 h = hash_key32(key) & (INS_TABLE - 1);
            while (atomicCAS(&s_tab[h], EMPTY_SLOT, (lo << 6) | j) != EMPTY_SLOT) h = (h + 1) & (INS_TABLE - 1);
        }
        __syncthreads();
        for (u32 e = tid; e < n_exist; e += INSERT_BLOCK) {
            const u128x key = mk128(ix.keys[2 * (off + e)], ix.keys[2 * (off + e) + 1]);
            const u32 cnt = ix.counts[off + e];
            u32 h = hash_key32(key) & (INS_TABLE - 1);
            for (;;) {
                const u32 v = s_tab[h];
                if (v == EMPTY_SLOT) break;
                if (eq128(chunk_key(P, s_rec, v >> 6, v & 63), key)) atomicAdd(&per_read_sum[tags[rc + (v >> 6)]], (unsigned long long)cnt);
                h = (h + 1) & (INS_TABLE - 1);
            }
        }
        rc += nrec;
        __syncthreads();
    }
}

// ===========================================================================
// k_enumerate: entries of partitions [p_begin, p_end) in order; out_base[p - p_begin]
// is the exclusive prefix of dir_cnt over that range (Brisk::next yields unhashed k-mers).
__global__ void __launch_bounds__(256) k_dir_prefix_block(const u32* __restrict__ dir_cnt, u64 n, u32* __restrict__ block_sums) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * 256 * SCAN_ITEMS;
    u32 acc = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const u64 j = base + (u64)i * 256 + threadIdx.x;
        if (j < n) acc += dir_cnt[j];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(64) k_enumerate(BriskParams P, IndexDev ix, u32 p_begin, u32 n_parts, const u64* __restrict__ out_base,
                                                  u64* __restrict__ out_lo, u64* __restrict__ out_hi, uint8_t* __restrict__ out_idx,
                                                  uint8_t* __restrict__ out_cnt) {
    const u32 pi = blockIdx.x;
    if (pi >= n_parts) return;
    const u32 part = p_begin + pi;
    const u32 cnt = ix.dir_cnt[part];
    const unsigned long long off = ix.dir_off[part];
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
    }
}

// k_lookup: one wave per query (Brisk::get: hash the minimizer, find the bucket, compare compacted k-mers)
__global__ void __launch_bounds__(256) k_lookup(BriskParams P, IndexDev ix, const u64* __restrict__ q_lo, const u64* __restrict__ q_hi,
                                                const uint8_t* __restrict__ q_idx, u64 n, uint8_t* __restrict__ out_data,
                                                uint8_t* __restrict__ out_found) {
    const u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63;
    if (qi >= n) return;
    const u32 idx = q_idx[qi];
    u128x km = mk128(q_lo[qi], q_hi[qi]);
    bool found = false;
    u32 data = 0;
    if (idx <= P.w) {
        const u64 mm = shr128(km, 2 * idx).lo & P.m_mask;
        const u64 h = mix2m(mm, P.m_mask);
        const u32 bucket = (u32)((h >> (2 * P.suff_reduc)) & P.bucket_mask);
        km = or128(andn128(km, shl128(mk128(P.m_mask, 0), 2 * idx)), shl128(mk128(h, 0), 2 * idx));
        const u32 cut = idx + P.suff_reduc;
        const u128x lowm = mask128(2 * cut);
        const u128x comp = or128(andn128(shr128(km, 2 * P.b), lowm), and128(km, lowm));
        const u128x key = make_key(P, bucket, and128(comp, mask128(2 * P.kb)), cut);
        const u32 part = bucket >> P.shift;
        const u32 cnt = ix.dir_cnt[part];
        const unsigned long long off = ix.dir_off[part];
        for (u32 e = lane; e < cnt && !found; e += 64) {
            if (ix.keys[2 * (off + e)] == key.lo && ix.keys[2 * (off + e) + 1] == key.hi) {
                found = true;
                data = ix.counts[off + e];
            }
        }
    }
    const unsigned long long bal = __ballot(found);
    if (bal) {
        const int src = __ffsll((long long)bal) - 1;
        data = __shfl(data, src, 64);
    }
    if (lane == 0) {
        out_found[qi] = bal ? 1 : 0;
        out_data[qi] = (uint8_t)data;
    }
}
