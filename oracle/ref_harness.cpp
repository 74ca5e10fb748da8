/*
 * ref_harness.cpp -- C-ABI driver around the REAL reference sources.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no
 * reference code: it #includes the reference's headers where they lie under
 * $(REF)/brisk and is linked (oracle/Makefile) with the reference's own
 * Kmers.cpp, hashing.cpp and Decycling.cpp into oracle/_ref/libbrisk_ref.so.
 * It exists so that the restatement in brisk_oracle.c and the golden vectors
 * in tests/golden/ are pinned by outputs of the reference itself.
 *
 * What is and is not the reference here:
 *   - L1 (enumerator, get_minimizer, rc, hashing, decycling, kmer_full) and
 *     L2 (Bucket<DATA>, SKL: insert_kmer / find_kmer_vector / next_kmer) are
 *     the reference's code, unmodified.
 *   - Brisk.hpp / DenseMenuYo.hpp / writer.hpp cannot be compiled in this
 *     image: they include un-vendored submodules (ankerl/unordered_dense.h,
 *     kff_io.hpp; .gitmodules:1-6, both directories empty) and stand-ins for
 *     missing headers are not allowed.  The ~40 lines of directory logic they
 *     contribute to the path (Brisk.hpp:123-147, DenseMenuYo.hpp:248-310,
 *     358-396, 476-521) are driven here by `RefIndex` below, written against
 *     the behaviour described in SURVEY.md Appendix A.7.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>
#include <omp.h>

/* expose DecyclingSet::coef for the table fixture; std headers are already in */
#define private public
#include "Decycling.h"
#undef private
#include "Kmers.hpp"
#include "buckets.hpp"
#include "hashing.hpp"

typedef unsigned __int128 u128;
#define REF_EXPORT extern "C" __attribute__((visibility("default")))

static inline u128 mk128(uint64_t lo, uint64_t hi) { return ((u128)hi << 64) | lo; }

/* ---- L1 unit entry points ------------------------------------------------ */
REF_EXPORT void ref_coef_table(unsigned m, double *out)
{
    DecyclingSet d(m);
    for (unsigned i = 0; i < 4 * m; i++) out[i] = d.coef[i];
}
REF_EXPORT unsigned ref_class(uint64_t x, unsigned m)
{
    DecyclingSet d(m);
    return d.memDouble(x);
}
REF_EXPORT void ref_class_many(const uint64_t *x, uint64_t n, unsigned m, uint8_t *out)
{
    DecyclingSet d(m);
    for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)d.memDouble(x[i]);
}
REF_EXPORT void ref_key_many(const uint64_t *x, uint64_t n, unsigned m, uint64_t *out)
{
    DecyclingSet d(m);
    const uint64_t mask = ((uint64_t)1 << (2 * m)) - 1;
    for (uint64_t i = 0; i < n; i++) out[i] = bfc_hash_64(x[i], mask, &d);
}
REF_EXPORT void ref_mix_inv_many(const uint64_t *x, uint64_t n, unsigned m, uint64_t *out)
{
    const uint64_t mask = ((uint64_t)1 << (2 * m)) - 1;
    for (uint64_t i = 0; i < n; i++) out[i] = bfc_hash_64_inv(x[i], mask);
}
REF_EXPORT uint64_t ref_rcbc(uint64_t x, unsigned n) { return rcbc(x, n); }
REF_EXPORT void ref_rcb(uint64_t lo, uint64_t hi, unsigned n, uint64_t *olo, uint64_t *ohi)
{
    u128 r = rcb(mk128(lo, hi), n);
    *olo = (uint64_t)r;
    *ohi = (uint64_t)(r >> 64);
}
REF_EXPORT int ref_canonized(uint64_t lo, uint64_t hi, unsigned n) { return canonized(mk128(lo, hi), n) ? 1 : 0; }
REF_EXPORT uint64_t ref_get_minimizer(uint64_t lo, uint64_t hi, unsigned K, unsigned m, uint8_t *pos, int *rev)
{
    DecyclingSet d(m);
    bool r = false;
    uint8_t p = 0;
    uint64_t mini = get_minimizer(mk128(lo, hi), (uint8_t)K, p, (uint8_t)m, r, ((uint64_t)1 << (2 * m)) - 1, &d);
    *pos = p;
    *rev = r ? 1 : 0;
    return mini;
}
REF_EXPORT void ref_compacted(uint64_t lo, uint64_t hi, unsigned b, unsigned idxp, uint64_t *olo, uint64_t *ohi)
{
    kmer_full km;
    km.kmer_s = mk128(lo, hi);
    u128 r = km.get_compacted((uint8_t)b, (uint8_t)idxp);
    *olo = (uint64_t)r;
    *ohi = (uint64_t)(r >> 64);
}

/* same contract as bo_enumerate in brisk_oracle.c */
REF_EXPORT int64_t ref_enumerate(const char *seq, uint64_t len, unsigned k, unsigned m,
                                 uint64_t *skm_ret, uint32_t *skm_n, uint64_t skm_cap,
                                 uint64_t *km_lo, uint64_t *km_hi, uint8_t *km_idx, uint64_t *km_mini,
                                 uint64_t km_cap, uint64_t *n_km_out)
{
    if (len < k) return -1;
    DecyclingSet d(m);
    std::string s(seq, len);
    SuperKmerEnumerator en(s, (uint8_t)k, (uint8_t)m, &d);
    std::vector<kmer_full> v;
    uint64_t n_skm = 0, n_km = 0;
    kint ret = en.next(v);
    while (!v.empty()) {
        if (n_skm >= skm_cap || n_km + v.size() > km_cap) return -1;
        skm_ret[n_skm] = (uint64_t)ret;
        skm_n[n_skm] = (uint32_t)v.size();
        n_skm++;
        for (auto &km : v) {
            km_lo[n_km] = (uint64_t)km.kmer_s;
            km_hi[n_km] = (uint64_t)(km.kmer_s >> 64);
            km_idx[n_km] = km.minimizer_idx;
            if (km_mini) km_mini[n_km] = (uint64_t)km.minimizer;
            n_km++;
        }
        v.clear();
        ret = en.next(v);
    }
    if (n_km_out) *n_km_out = n_km;
    return (int64_t)n_skm;
}

/* ---- index driver over the reference's Bucket<uint8_t> ---------------- */
struct RefIndex {
    Parameters params;
    unsigned suff_reduc;
    uint64_t n_buckets, n_stripes;
    std::vector<uint32_t> dir;                                    /* 1-based, per bucket id */
    std::vector<std::vector<std::unique_ptr<Bucket<uint8_t>>>> rows; /* per stripe */
    std::vector<omp_lock_t> locks;
    uint64_t nb_kmers = 0;
    RefIndex(unsigned k, unsigned m, unsigned b) : params((uint8_t)k, (uint8_t)m, (uint8_t)b)
    {
        Bucket<uint8_t>::setParameters(params);
        suff_reduc = (params.m_reduc + 1) / 2;
        n_buckets = (uint64_t)1 << (2 * b);
        n_stripes = (uint64_t)1 << (2 * std::min(12u, b));
        dir.assign(n_buckets, 0);
        rows.resize(n_stripes);
        locks.resize(n_stripes);
        for (auto &l : locks) omp_init_lock(&l);
    }
    uint64_t small_minimizer(const kmer_full &hashed0) const
    {
        uint64_t sm = (uint64_t)(hashed0.minimizer >> (2 * suff_reduc));
        return sm & (n_buckets - 1);
    }
    Bucket<uint8_t> *bucket(uint64_t sm, bool create)
    {
        uint32_t &slot = dir[sm];
        auto &row = rows[sm % n_stripes];
        if (slot == 0) {
            if (!create) return nullptr;
            row.emplace_back(new Bucket<uint8_t>());
            slot = (uint32_t)row.size();
        }
        return row[slot - 1].get();
    }
    /* find-all, insert-missing, then the counter app's update */
    void count_superkmer(std::vector<kmer_full> &skm)
    {
        if (skm.empty()) return;
        for (auto &km : skm) km.hash_kmer_minimizer_inplace(params.m);
        const uint64_t sm = small_minimizer(skm[0]);
        omp_lock_t *lk = &locks[sm % n_stripes];
        omp_set_lock(lk);
        /* a bucket only exists once it holds a k-mer: find_kmer_vector is never
         * run on an empty Bucket (DenseMenuYo.hpp:372-375 returns early) */
        Bucket<uint8_t> *bk = bucket(sm, false);
        std::vector<uint8_t *> found(skm.size(), nullptr);
        if (bk) found = bk->find_kmer_vector(skm);
        std::vector<bool> fresh(skm.size(), false);
        uint64_t created = 0;
        for (size_t i = 0; i < skm.size(); i++) {
            if (found[i] == nullptr) {
                if (!bk) bk = bucket(sm, true);
                bk->insert_kmer(skm[i]);
                fresh[i] = true;
                created++;
            }
        }
        found = bk->find_kmer_vector(skm);
        for (size_t i = 0; i < skm.size(); i++) {
            if (fresh[i])
                *found[i] = 1;
            else
                (*found[i])++;
        }
        omp_unset_lock(lk);
#pragma omp atomic
        nb_kmers += created;
    }
};

REF_EXPORT void *ref_index_new(unsigned k, unsigned m, unsigned b)
{
    if (!(b >= 1 && b <= m && m < k && k <= 63 && (m & 1))) return nullptr;
    return new RefIndex(k, m, b);
}
REF_EXPORT void ref_index_free(void *h) { delete (RefIndex *)h; }

REF_EXPORT int ref_index_insert_reads(void *h, const char *bases, const uint64_t *offs, uint64_t n_reads, int threads)
{
    RefIndex *ix = (RefIndex *)h;
    const unsigned k = ix->params.k, m = ix->params.m;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
    for (int64_t r = 0; r < (int64_t)n_reads; r++) {
        const uint64_t len = offs[r + 1] - offs[r];
        if (len < k) continue;
        std::string s(bases + offs[r], len);
        SuperKmerEnumerator en(s, (uint8_t)k, (uint8_t)m, ix->params.dede);
        std::vector<kmer_full> v;
        en.next(v);
        while (!v.empty()) {
            ix->count_superkmer(v);
            v.clear();
            en.next(v);
        }
    }
    return 0;
}
REF_EXPORT uint64_t ref_index_nb_kmers(void *h) { return ((RefIndex *)h)->nb_kmers; }
REF_EXPORT uint64_t ref_index_nb_buckets(void *h)
{
    RefIndex *ix = (RefIndex *)h;
    uint64_t n = 0;
    for (uint32_t d : ix->dir) n += (d != 0);
    return n;
}
REF_EXPORT uint64_t ref_index_nb_skmers(void *h)
{
    RefIndex *ix = (RefIndex *)h;
    uint64_t n = 0;
    for (auto &row : ix->rows)
        for (auto &b : row) n += b->skml.size();
    return n;
}

/* enumerate every entry (bucket id ascending, storage order) with its count,
 * k-mers unhashed -- the next()+get() walk of counter.cpp:90-126 */
REF_EXPORT uint64_t ref_index_dump(void *h, uint64_t *lo, uint64_t *hi, uint8_t *idx, uint8_t *cnt, uint64_t cap)
{
    RefIndex *ix = (RefIndex *)h;
    const uint8_t m = ix->params.m;
    uint64_t n = 0;
    for (uint64_t sm = 0; sm < ix->n_buckets; sm++) {
        Bucket<uint8_t> *bk = ix->bucket(sm, false);
        if (!bk) continue;
        uint32_t si = 0, ki = 0;
        while (bk->has_next_kmer(si, ki)) {
            kmer_full km((kint)0, 0, m, ix->params.dede);
            bk->next_kmer(km, (kint)sm, si, ki);
            km.minimizer_idx -= (uint8_t)ix->suff_reduc;
            km.compute_mini(m);
            /* km is hashed here; look its count up, then unhash */
            uint8_t *c = bk->find_kmer(km);
            km.unhash_kmer_minimizer(m);
            if (n >= cap) return n;
            lo[n] = (uint64_t)km.kmer_s;
            hi[n] = (uint64_t)(km.kmer_s >> 64);
            idx[n] = km.minimizer_idx;
            cnt[n] = c ? *c : 0;
            n++;
        }
    }
    return n;
}

/* Brisk::get on an unhashed (kmer_s, minimizer_idx) */
REF_EXPORT int ref_index_get(void *h, uint64_t lo, uint64_t hi, unsigned idx)
{
    RefIndex *ix = (RefIndex *)h;
    kmer_full km(mk128(lo, hi), (uint8_t)idx, ix->params.m, ix->params.dede);
    km.hash_kmer_minimizer_inplace(ix->params.m);
    Bucket<uint8_t> *bk = ix->bucket(ix->small_minimizer(km), false);
    if (!bk) return -1;
    uint8_t *c = bk->find_kmer(km);
    return c ? (int)*c : -1;
}

/* query_sequence of counter.cpp:281-310, per read */
REF_EXPORT int ref_index_query_reads(void *h, const char *bases, const uint64_t *offs, uint64_t n_reads, uint64_t *sums)
{
    RefIndex *ix = (RefIndex *)h;
    const unsigned k = ix->params.k, m = ix->params.m;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t len = offs[r + 1] - offs[r];
        sums[r] = 0;
        if (len < k) continue;
        std::string s(bases + offs[r], len);
        SuperKmerEnumerator en(s, (uint8_t)k, (uint8_t)m, ix->params.dede);
        std::vector<kmer_full> v;
        kint ret = en.next(v);
        uint64_t total = 0;
        while (!v.empty()) {
            for (auto &km : v) km.hash_kmer_minimizer_inplace(ix->params.m);
            Bucket<uint8_t> *bk = ix->bucket(ix->small_minimizer(v[0]), false);
            if (bk) {
                std::vector<uint8_t *> f = bk->find_kmer_vector(v);
                for (uint8_t *p : f)
                    if (p) total += *p;
            }
            v.clear();
            ret = en.next(v);
            if (ret == 0) break;
        }
        sums[r] = total;
    }
    return 0;
}
