// brisk_device.h -- device-side primitives of the Brisk hot path for gfx950.
//
// Behavioural spec: SURVEY.md Appendix A; each helper cites the reference
// file:line whose behaviour it reproduces.  Written for CDNA4 only: 64-wide
// waves, u64 arithmetic on 32-bit VALU pairs, LDS for the lookup tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

// ---------------------------------------------------------------------------
// Index-wide constants, passed by value to every kernel.
struct BriskParams {
    u32 k, m, b;
    u32 w;           // k - m: largest minimizer_idx
    u32 suff_reduc;  // ceil((m-b)/2): minimizer nts dropped on the suffix side (Brisk.hpp:135)
    u32 kb;          // k - b: nts of a compacted k-mer (parameters.hpp:30)
    u32 nw;          // u64 words holding a compacted super-k-mer (2k-m-b nts)
    u32 stride;      // nw + 1: record words, last one is the header
    u32 part_bits;   // log2(#partitions)
    u32 ext_bits;    // routing-id bits beyond the bucket's (0 once 2b >= 24): rid = bucket << ext_bits | extra.  The extra bits are
                     // more bits of the same hashed minimizer and, below them, cls_bits of the k-mer's minimizer_idx class
    u32 cls_bits;    // m so small that the minimizer hash runs out of bits before there are 2^24 routing ids (2m < 24): the
                     // class floor(minimizer_idx / cls_width) of a k-mer extends its routing id.  Few distinct minimizers
                     // mean big partitions; k-mers of different minimizer_idx never meet, so the classes cut every such
                     // partition into 2^cls_bits independent ones, and the scan cuts a super-k-mer's record where the class changes
    u32 cls_width;
    u32 shift;       // 2b + ext_bits - part_bits: routing-id bits kept inside an entry key
    u32 n_owners, owner_rank;
    u64 m_mask;      // 2m ones
    u64 bucket_mask; // 2b ones
};

// record header word: [0,32) routing id, [32,40) n k-mers, [40,48) idx' of k-mer 0,
// [48,56) multiplicity and bit 56 "has a multiplicity" (zero as the scan emits records: one vector each; k_insert_big sets
// them on records it has collapsed, brisk_hip_reallocate on the one-k-mer records that carry an entry's count), rest zero.
// idx' = minimizer_idx + suff_reduc (SuperKmerLight.hpp:98).
// Routing id = the bucket id (Brisk.hpp:135-137) followed by ext_bits more bits of the same hashed minimizer:
// every k-mer of a super-k-mer shares them, so records can be spread over up to 2^24 partitions however small b is
// (a partition is then a slice of a bucket instead of a range of buckets).  ext_bits == 0: the bucket id itself.
__device__ __forceinline__ u64 rec_header(u32 rid, u32 n, u32 idx0p) {
    return (u64)rid | ((u64)n << 32) | ((u64)idx0p << 40);
}
__device__ __forceinline__ u32 hdr_bucket(u64 h) { return (u32)h; }  // the routing id
__device__ __forceinline__ u32 hdr_n(u64 h) { return (u32)(h >> 32) & 0xffu; }
__device__ __forceinline__ u32 hdr_idx0(u64 h) { return (u32)(h >> 40) & 0xffu; }

// ---------------------------------------------------------------------------
// 128-bit values as two u64 (k <= 63 => a k-mer is <= 126 bits)
struct u128x {
    u64 lo, hi;
};
__device__ __forceinline__ u128x mk128(u64 lo, u64 hi) { return u128x{lo, hi}; }
__device__ __forceinline__ bool eq128(u128x a, u128x b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ bool lt128(u128x a, u128x b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ u128x shr128(u128x a, u32 s) {  // s in [0,127]
    if (s == 0) return a;
    if (s >= 64) return u128x{a.hi >> (s - 64), 0};
    return u128x{(a.lo >> s) | (a.hi << (64 - s)), a.hi >> s};
}
__device__ __forceinline__ u128x shl128(u128x a, u32 s) {  // s in [0,127]
    if (s == 0) return a;
    if (s >= 64) return u128x{0, a.lo << (s - 64)};
    return u128x{a.lo << s, (a.hi << s) | (a.lo >> (64 - s))};
}
__device__ __forceinline__ u128x mask128(u32 bits) {  // low `bits` ones, bits in [0,128]
    if (bits >= 128) return u128x{~0ull, ~0ull};
    if (bits >= 64) return u128x{~0ull, bits == 64 ? 0ull : ((1ull << (bits - 64)) - 1)};
    return u128x{bits == 0 ? 0ull : ((1ull << bits) - 1), 0};
}
__device__ __forceinline__ u128x and128(u128x a, u128x b) { return u128x{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ u128x or128(u128x a, u128x b) { return u128x{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ u128x andn128(u128x a, u128x b) { return u128x{a.lo & ~b.lo, a.hi & ~b.hi}; }

// 256-bit value for a whole super-k-mer (<= 2k-m <= 125 nts).  Named fields, never
// an indexed array: runtime-indexed register arrays go to scratch on hipcc.
struct W4 {
    u64 w0, w1, w2, w3;  // w0 least significant
};
__device__ __forceinline__ W4 w4_shr(W4 a, u32 s) {  // s in [0,255]
    const u32 ws = s >> 6, bs = s & 63;
    u64 t0 = ws == 0 ? a.w0 : ws == 1 ? a.w1 : ws == 2 ? a.w2 : a.w3;
    u64 t1 = ws == 0 ? a.w1 : ws == 1 ? a.w2 : ws == 2 ? a.w3 : 0;
    u64 t2 = ws == 0 ? a.w2 : ws == 1 ? a.w3 : 0;
    u64 t3 = ws == 0 ? a.w3 : 0;
    if (bs) {
        t0 = (t0 >> bs) | (t1 << (64 - bs));
        t1 = (t1 >> bs) | (t2 << (64 - bs));
        t2 = (t2 >> bs) | (t3 << (64 - bs));
        t3 = t3 >> bs;
    }
    return W4{t0, t1, t2, t3};
}
__device__ __forceinline__ W4 w4_shl(W4 a, u32 s) {  // s in [0,255]
    const u32 ws = s >> 6, bs = s & 63;
    u64 t3 = ws == 0 ? a.w3 : ws == 1 ? a.w2 : ws == 2 ? a.w1 : a.w0;
    u64 t2 = ws == 0 ? a.w2 : ws == 1 ? a.w1 : ws == 2 ? a.w0 : 0;
    u64 t1 = ws == 0 ? a.w1 : ws == 1 ? a.w0 : 0;
    u64 t0 = ws == 0 ? a.w0 : 0;
    if (bs) {
        t3 = (t3 << bs) | (t2 >> (64 - bs));
        t2 = (t2 << bs) | (t1 >> (64 - bs));
        t1 = (t1 << bs) | (t0 >> (64 - bs));
        t0 = t0 << bs;
    }
    return W4{t0, t1, t2, t3};
}
__device__ __forceinline__ u64 ones_upto(int bits) {  // low clamp(bits,0,64) ones
    return bits <= 0 ? 0ull : bits >= 64 ? ~0ull : ((1ull << bits) - 1);
}
__device__ __forceinline__ W4 w4_mask(u32 bits) {  // low `bits` ones, bits in [0,256]
    return W4{ones_upto((int)bits), ones_upto((int)bits - 64), ones_upto((int)bits - 128), ones_upto((int)bits - 192)};
}
__device__ __forceinline__ W4 w4_and(W4 a, W4 b) { return W4{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2, a.w3 & b.w3}; }
__device__ __forceinline__ W4 w4_andn(W4 a, W4 b) { return W4{a.w0 & ~b.w0, a.w1 & ~b.w1, a.w2 & ~b.w2, a.w3 & ~b.w3}; }
__device__ __forceinline__ W4 w4_or(W4 a, W4 b) { return W4{a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2, a.w3 | b.w3}; }

// ---------------------------------------------------------------------------
// 2-bit packed sequence stream: 16 nts per u32, first nt in the top two bits
// (A0 C1 T2 G3, Kmers.cpp:442-444).  The stream must be readable two words past
// the last nt touched.
__device__ __forceinline__ u32 nt_at(const u32* __restrict__ packed, u64 q) {
    return (packed[q >> 4] >> (30 - 2 * (u32)(q & 15))) & 3u;
}
// cnt (1..32) nts starting at stream index q, first nt most significant, right-aligned
__device__ __forceinline__ u64 load_nts(const u32* __restrict__ packed, u64 q, u32 cnt) {
    const u64 wi = q >> 4;
    const u32 o = (u32)(q & 15) * 2;
    const u64 top = ((u64)packed[wi] << 32) | packed[wi + 1];
    const u64 nxt = packed[wi + 2];
    const u64 x = o ? ((top << o) | (nxt >> (32 - o))) : top;
    return x >> (64 - 2 * cnt);
}
// nts [q, q+len) as a right-aligned 256-bit value, len in [1,128]
__device__ __forceinline__ W4 load_span(const u32* __restrict__ packed, u64 q, u32 len) {
    // chunk c (from the END) covers nts [len-32(c+1), len-32c) clipped at 0
    W4 r{0, 0, 0, 0};
    {
        const u32 c = len >= 32 ? 32 : len;
        r.w0 = load_nts(packed, q + len - c, c);
    }
    if (len > 32) {
        const u32 c = len >= 64 ? 32 : len - 32;
        r.w1 = load_nts(packed, q + len - 32 - c, c);
    }
    if (len > 64) {
        const u32 c = len >= 96 ? 32 : len - 64;
        r.w2 = load_nts(packed, q + len - 64 - c, c);
    }
    if (len > 96) {
        const u32 c = len >= 128 ? 32 : len - 96;
        r.w3 = load_nts(packed, q + len - 96 - c, c);
    }
    return r;
}

// ---------------------------------------------------------------------------
// reverse complements
__device__ __forceinline__ u64 rev_nts64(u64 x) {  // reverse the order of the 32 2-bit groups
    u64 r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}
// true reverse complement of an n-mer, n <= 32 (rcbc, Kmers.cpp:320-332)
__device__ __forceinline__ u64 rc64(u64 x, u32 n) { return rev_nts64(x ^ 0xaaaaaaaaaaaaaaaaull) >> (64 - 2 * n); }
// true reverse complement of a len-nt 256-bit value
__device__ __forceinline__ W4 w4_rc(W4 a, u32 len) {
    const u64 c = 0xaaaaaaaaaaaaaaaaull;
    W4 r{rev_nts64(a.w3 ^ c), rev_nts64(a.w2 ^ c), rev_nts64(a.w1 ^ c), rev_nts64(a.w0 ^ c)};
    return w4_shr(r, 256 - 2 * len);
}
// the reference's 128-bit "rc" AS EXECUTED (Kmers.cpp:293-316, F4): nts reversed
// inside each byte only, complemented, then shifted right by 128-2n.
__device__ __forceinline__ u64 swap_in_bytes(u64 x) {
    const u64 c1 = 0x0f0f0f0f0f0f0f0full, c2 = 0x3333333333333333ull;
    x = ((x & c1) << 4) | ((x >> 4) & c1);
    x = ((x & c2) << 2) | ((x >> 2) & c2);
    return x;
}
__device__ __forceinline__ bool canonized_as_executed(u128x x, u32 n) {  // Kmers.cpp:342-353
    u128x r{swap_in_bytes(x.lo) ^ 0xaaaaaaaaaaaaaaaaull, swap_in_bytes(x.hi) ^ 0xaaaaaaaaaaaaaaaaull};
    r = shr128(r, 128 - 2 * n);
    return !lt128(r, x);  // x == min(x, r)
}

// ---------------------------------------------------------------------------
// order key (hashing.cpp:8-19): invertible mixer on 2m bits + decycling class << 62
__device__ __forceinline__ u64 mix2m(u64 x, u64 M) {
    x = (~x + (x << 21)) & M;
    x ^= x >> 24;
    x = (x + (x << 3) + (x << 8)) & M;
    x ^= x >> 14;
    x = (x + (x << 2) + (x << 4)) & M;
    x ^= x >> 28;
    x = (x + (x << 31)) & M;
    return x;
}
// inverse on the low 2m bits (hashing.cpp:23-48)
__device__ __forceinline__ u64 mix2m_inv(u64 y, u64 M) {
    u64 t;
    t = y - (y << 31);
    y = (y - (t << 31)) & M;
    t = y ^ (y >> 28);
    y = y ^ (t >> 28);
    y = (y * 14933078535860113213ull) & M;
    t = y ^ (y >> 14);
    t = y ^ (t >> 14);
    t = y ^ (t >> 14);
    y = y ^ (t >> 14);
    y = (y * 15244667743933553977ull) & M;
    t = y ^ (y >> 24);
    y = y ^ (t >> 24);
    t = ~y;
    t = ~(y - (t << 21));
    t = ~(y - (t << 21));
    y = ~(y - (t << 21)) & M;
    return y;
}
// exact decycling fold (Decycling.cpp:17-24): same adds, same order, IEEE f64.
// coef: the host-built table, staged in LDS (4 values per position, so a wave's
// 64 lookups hit at most 4 distinct addresses on 8 distinct banks).
__device__ __forceinline__ double fold_r(u64 x, u32 m, const double* coef) {
    double r = 0.0;
    for (u32 j = 0; j + 1 < m; j++) {
        r += coef[4 * (m - 1 - j) + (u32)(x & 3)];
        x >>= 2;
    }
    return r;
}
__device__ __forceinline__ u32 decy_class(u64 x, u32 m, const double* coef) {  // Decycling.cpp:38-52
    const double eps = 0.000001;
    const double r = fold_r(x, m, coef);
    const u64 rot = ((x & 3) << (2 * (m - 1))) + (x >> 2);
    if (r > eps) {
        if (fold_r(rot, m, coef) < eps) return 0;
    } else if (r < -eps) {
        if (fold_r(rot, m, coef) > -eps) return 1;
    }
    return 2;
}
__device__ __forceinline__ u64 order_key(u64 x, u32 m, u64 M, const double* coef) {
    return ((u64)decy_class(x, m, coef) << 62) + mix2m(x, M);
}

// ---------------------------------------------------------------------------
// routing id of a k-mer with hashed minimizer h (2m bits) and minimizer_idx: [bucket (2b bits, Brisk.hpp:135-137) | bits of what is
// left of h: the 2*suff_reduc bits below the bucket first, then the bits above it | cls_bits: the class of minimizer_idx]
__device__ __forceinline__ u32 cls_of(const BriskParams& P, u32 minimizer_idx) {
    const u32 c = minimizer_idx / P.cls_width, top = (1u << P.cls_bits) - 1;
    return c < top ? c : top;
}
// the routing id without its class bits
__device__ __forceinline__ u32 routing_base(const BriskParams& P, u64 h) {
    const u32 bucket = (u32)((h >> (2 * P.suff_reduc)) & P.bucket_mask);
    const u32 he = P.ext_bits - P.cls_bits;  // bits taken from what is left of the hash
    if (!he) return bucket;
    const u64 rest = ((h >> (2 * (P.suff_reduc + P.b))) << (2 * P.suff_reduc)) | (h & ((1ull << (2 * P.suff_reduc)) - 1));
    return (bucket << he) | (u32)(rest & ((1u << he) - 1));
}
__device__ __forceinline__ u32 routing_id(const BriskParams& P, u64 h, u32 minimizer_idx) {
    const u32 base = routing_base(P, h);
    return P.cls_bits ? (base << P.cls_bits) | cls_of(P, minimizer_idx) : base;
}
// entry key inside a partition: [routing id low `shift` bits | compacted k-mer (2kb) | idx' (6)]
__device__ __forceinline__ u128x make_key(const BriskParams& P, u32 bucket, u128x comp, u32 idxp) {
    u128x key = shl128(comp, 6);
    key.lo |= idxp;
    if (P.shift) key = or128(key, shl128(mk128(bucket & ((1u << P.shift) - 1), 0), 2 * P.kb + 6));
    return key;
}
// k-mer j of a record: compacted_j = (C >> 2(n-1-j)) & ones(2kb)   (SuperKmerLight.hpp:301-312)
__device__ __forceinline__ u128x record_kmer(const BriskParams& P, const u64* c, u32 n, u32 j) {
    const u32 s = 2 * (n - 1 - j);
    const u32 ws = s >> 6, bs = s & 63;
    u64 a0 = ws < P.nw ? c[ws] : 0, a1 = ws + 1 < P.nw ? c[ws + 1] : 0, a2 = ws + 2 < P.nw ? c[ws + 2] : 0;
    u128x r;
    if (bs) {
        r.lo = (a0 >> bs) | (a1 << (64 - bs));
        r.hi = (a1 >> bs) | (a2 << (64 - bs));
    } else {
        r.lo = a0;
        r.hi = a1;
    }
    return and128(r, mask128(2 * P.kb));
}
// inverse of make_key + compaction: the hashed k-mer and minimizer_idx of an entry
__device__ __forceinline__ u128x entry_hashed_kmer(const BriskParams& P, u32 part, u128x key, u32* idx_out) {
    const u32 idxp = (u32)key.lo & 63u;
    const u128x comp = and128(shr128(key, 6), mask128(2 * P.kb));
    u32 bucket = part << P.shift;
    if (P.shift) bucket |= (u32)shr128(key, 2 * P.kb + 6).lo & ((1u << P.shift) - 1);
    bucket >>= P.ext_bits;  // routing id -> bucket id
    const u128x suffix = and128(comp, mask128(2 * idxp));
    const u128x prefix = shr128(comp, 2 * idxp);
    u128x r = or128(shl128(prefix, 2 * (idxp + P.b)), suffix);
    r = or128(r, shl128(mk128(bucket, 0), 2 * idxp));
    *idx_out = idxp - P.suff_reduc;
    return r;
}
// slot hash of an entry key for the LDS tables: rotate-xor fold of the four 32-bit words, one 32-bit multiply
// (64-bit multiplies are four quarter-rate instructions each, and this runs once per k-mer instance and per
// streamed entry)
__device__ __forceinline__ u32 hash_key32(u128x k) {
    const u32 a = (u32)k.lo, b = (u32)(k.lo >> 32), c = (u32)k.hi, d = (u32)(k.hi >> 32);
    u32 z = a ^ ((b << 7) | (b >> 25)) ^ ((c << 13) | (c >> 19)) ^ ((d << 21) | (d >> 11));
    z ^= z >> 16;
    z *= 0x85EBCA6Bu;
    z ^= z >> 13;
    return z;
}
