// Brisk.hpp -- facade header with the reference's name and public surface (brisk/Brisk.hpp:23-42),
// over the C-ABI of include/brisk_hip.h.  Everything that computes runs on the GPU; this header only
// marshals vectors of kmer_full to flat arrays and keeps DATA (the caller's payload) in host memory,
// addressed by the dense entry ids the device assigns.
#ifndef BRISK_AMD_BRISK_HPP
#define BRISK_AMD_BRISK_HPP
#include <sys/resource.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "DenseMenuYo.hpp"
#include "Kmers.hpp"
#include "buckets.hpp"
#include "parameters.hpp"

template <class DATA>
class Brisk {
  public:
    DenseMenuYo<DATA>* menu;
    Parameters params;

    explicit Brisk(Parameters& parameters) : menu(nullptr), params(parameters), enum_cursor_(0), enum_pos_(0) {
        menu = new DenseMenuYo<DATA>(params);  // throws std::invalid_argument on a bad (k, m, b)
    }
    ~Brisk() { delete menu; }
    Brisk(const Brisk&) = delete;
    Brisk& operator=(const Brisk&) = delete;

    // brisk/Brisk.hpp:64-69
    DATA* get(kmer_full& kmer) {
        {
            // The walk the reference's apps run -- next() for every entry, get() on what it returned (apps/counter.cpp:90-126) --
            // needs no look-up: the enumeration brought the entry's id along, and ids are for life (entries are never removed).
            std::lock_guard<std::mutex> g(call_mu_);
            if (have_last_ && kmer.kmer_s == last_kmer_ && kmer.minimizer_idx == last_idx_) return slot(last_id_);
        }
        std::vector<kmer_full> one(1, kmer);
        return find(one)[0];
    }
    // README.md:49-50,63,136 of the reference document `insert(kmer)` ("if absent, allocate a DATA space and return a
    // pointer to it; if present, similar to get") although brisk/Brisk.hpp never defines it: a thin extra over
    // insert_superkmer.  The DATA of a new entry is uninitialised, as with insert_superkmer.
    DATA* insert(kmer_full& kmer) {
        std::vector<kmer_full> one(1, kmer);
        std::vector<bool> fresh;
        return insert_superkmer(one, fresh)[0];
    }
    // brisk/Brisk.hpp:123-147: entries that were absent are created; their DATA is uninitialised and
    // newly_inserted[i] tells the caller to initialise it.  Pointers stay valid for the life of the index.
    std::vector<DATA*> insert_superkmer(std::vector<kmer_full>& superkmer, std::vector<bool>& newly_inserted) {
        std::vector<DATA*> result;
        if (superkmer.empty()) return result;
        std::lock_guard<std::mutex> g(call_mu_);  // before flatten(): lo_/hi_/idx_ are shared by every caller of this index
        flatten(superkmer);
        ids_.resize(superkmer.size());
        new_.resize(superkmer.size());
        check(brisk_hip_upsert_kmers(menu->handle, lo_.data(), hi_.data(), idx_.data(), superkmer.size(), ids_.data(), new_.data()));
        for (size_t i = 0; i < superkmer.size(); i++) {
            newly_inserted.push_back(new_[i] != 0);
            result.push_back(slot(ids_[i]));
        }
        return result;
    }
    // brisk/Brisk.hpp:102-118: NULL for absent k-mers
    std::vector<DATA*> get_superkmer(std::vector<kmer_full>& superkmer) { return find(superkmer); }
    // declared by the reference (brisk/Brisk.hpp:27-28) and never defined there
    std::vector<DATA*> insert_sequence(const std::string& str, std::vector<bool>& newly_inserted) {
        std::vector<DATA*> all;
        std::string s(str);
        SuperKmerEnumerator en(s, params.k, params.m, params.dede);
        std::vector<kmer_full> v;
        en.next(v);
        while (!v.empty()) {
            std::vector<DATA*> part = insert_superkmer(v, newly_inserted);
            all.insert(all.end(), part.begin(), part.end());
            v.clear();
            en.next(v);
        }
        return all;
    }
    std::vector<DATA*> get_sequence(const std::string& str) {
        std::vector<DATA*> all;
        std::string s(str);
        SuperKmerEnumerator en(s, params.k, params.m, params.dede);
        std::vector<kmer_full> v;
        en.next(v);
        while (!v.empty()) {
            std::vector<DATA*> part = get_superkmer(v);
            all.insert(all.end(), part.begin(), part.end());
            v.clear();
            en.next(v);
        }
        return all;
    }
    // brisk/Brisk.hpp:152-161: the reference takes a lock stripe the caller holds while it touches DATA
    void protect_data(const kmer_full&) { data_mu_.lock(); }
    void unprotect_data(const kmer_full&) { data_mu_.unlock(); }
    // brisk/Brisk.hpp:166-179: every entry once, k-mer unhashed, ascending bucket range
    bool next(kmer_full& kmer) {
        std::lock_guard<std::mutex> g(call_mu_);
        if (enum_pos_ >= e_lo_.size()) {
            const uint64_t cap = 1u << 16;
            e_lo_.resize(cap);
            e_hi_.resize(cap);
            e_idx_.resize(cap);
            e_ids_.resize(cap);
            uint64_t n = 0;
            int rc = brisk_hip_enumerate_ids(menu->handle, &enum_cursor_, e_lo_.data(), e_hi_.data(), e_idx_.data(), e_ids_.data(), cap, &n);
            for (uint64_t bigger = cap * 16; rc == BRISK_HIP_ECAPACITY; bigger *= 16) {  // one bucket range larger than the buffer
                e_lo_.resize(bigger);
                e_hi_.resize(bigger);
                e_idx_.resize(bigger);
                e_ids_.resize(bigger);
                rc = brisk_hip_enumerate_ids(menu->handle, &enum_cursor_, e_lo_.data(), e_hi_.data(), e_idx_.data(), e_ids_.data(), bigger, &n);
            }
            check(rc);
            e_lo_.resize(n);
            e_hi_.resize(n);
            enum_pos_ = 0;
            if (n == 0) return false;
        }
        kmer.kmer_s = ((kint)e_hi_[enum_pos_] << 64) | e_lo_[enum_pos_];
        kmer.minimizer_idx = e_idx_[enum_pos_];
        kmer.compute_mini(params.m);
        kmer.interleaved.clear();
        last_kmer_ = kmer.kmer_s;
        last_idx_ = kmer.minimizer_idx;
        last_id_ = e_ids_[enum_pos_];
        have_last_ = true;
        enum_pos_++;
        return true;
    }
    // Every entry once, with its DATA (the BriskWriter's walk, brisk/writer.hpp:99-176: there over the bucket matrix and
    // the SKLs of every bucket; here over the device's enumeration, buckets ascending).  f(const kmer_full&, const DATA*).
    template <class F>
    void visit_entries(F&& f) {
        std::lock_guard<std::mutex> g(call_mu_);
        uint64_t cursor = 0, n = 0, cap = 1u << 16;
        std::vector<uint64_t> lo(cap), hi(cap);
        std::vector<uint8_t> idx(cap);
        std::vector<uint32_t> ids(cap);
        kmer_full km((kint)0, 0, params.m, params.dede);
        for (;;) {
            int rc = brisk_hip_enumerate_ids(menu->handle, &cursor, lo.data(), hi.data(), idx.data(), ids.data(), cap, &n);
            if (rc == BRISK_HIP_ECAPACITY) {  // one bucket range larger than the buffer
                cap *= 16;
                lo.resize(cap);
                hi.resize(cap);
                idx.resize(cap);
                ids.resize(cap);
                continue;
            }
            check(rc);
            if (n == 0) break;
            for (uint64_t i = 0; i < n; i++) {
                km.kmer_s = ((kint)hi[i] << 64) | lo[i];
                km.minimizer_idx = idx[i];
                f(km, slot(ids[i]));
            }
        }
    }
    void restart_kmer_enumeration() {
        std::lock_guard<std::mutex> g(call_mu_);
        enum_cursor_ = 0;
        enum_pos_ = 0;
        e_lo_.clear();
        e_hi_.clear();
    }
    // brisk/Brisk.hpp:194-197.  memory_usage is the process' peak RSS in kB, as in the reference.
    void stats(uint64_t& nb_buckets, uint64_t& nb_skmers, uint64_t& nb_kmers, uint64_t& memory_usage, uint64_t& largest_bucket) const {
        uint64_t dev_bytes = 0;
        std::lock_guard<std::mutex> g(call_mu_);
        check(brisk_hip_stats(menu->handle, &nb_buckets, &nb_skmers, &nb_kmers, &dev_bytes, &largest_bucket));
        memory_usage = getMemorySelfMaxUsed();
    }
    uint64_t getMemorySelfMaxUsed() const {
        struct rusage usage;
        return getrusage(RUSAGE_SELF, &usage) == 0 ? (uint64_t)usage.ru_maxrss : 0;
    }
    // brisk/Brisk.hpp:202-224: the index re-bucketed to (m + 2, b + 2); the reference's only call site is commented out
    // (:124-129) and its loop calls an update_kmer overload that no file defines, so "the k-mer under the new m" is
    // taken from the path itself: what SuperKmerEnumerator yields for the k-mer as a sequence of k nts at the new m
    // (include/brisk_hip.h, brisk_hip_reallocate, does the same on the device for bulk-count indexes).  DATA lives on
    // the host here, so the walk is the reference's: every entry, in turn, into the new index, its DATA copied over
    // (`*value = *old_value`, :217; entries that merge keep the DATA of the last one, as that line does).  One launch
    // per k-mer: the per-call API's cost, on a path the reference never runs.  Throws std::invalid_argument when
    // (k, m + 2, b + 2) is not a valid triple (m + 2 >= k).
    void reallocate() {
        Parameters grown(params.k, (uint8_t)(params.m + 2), (uint8_t)(params.b + 2));
        std::unique_ptr<DenseMenuYo<DATA>> big(new DenseMenuYo<DATA>(grown));
        struct Old {
            kint kmer_s;
            const DATA* data;
        };
        std::vector<Old> old;
        visit_entries([&](const kmer_full& km, const DATA* d) { old.push_back(Old{km.kmer_s, d}); });
        std::lock_guard<std::mutex> g(call_mu_);
        std::vector<std::unique_ptr<DATA[]>> fresh_chunks;
        auto fresh_slot = [&](uint32_t id) -> DATA* {
            const size_t c = id >> 16;
            while (fresh_chunks.size() <= c) fresh_chunks.emplace_back(new DATA[1u << 16]);
            return &fresh_chunks[c][id & 0xffff];
        };
        std::vector<kmer_full> v;
        for (const Old& o : old) {
            std::string s = kmer2str(o.kmer_s, params.k);
            SuperKmerEnumerator en(s, grown.k, grown.m, grown.dede);
            v.clear();
            en.next(v);  // a sequence of k nts: one vector of one k-mer
            if (v.size() != 1) throw std::runtime_error("Brisk::reallocate: a k-mer did not come back as one k-mer");
            const uint64_t lo = (uint64_t)v[0].kmer_s, hi = (uint64_t)(v[0].kmer_s >> 64);
            uint32_t id = 0;
            uint8_t is_new = 0;
            if (brisk_hip_upsert_kmers(big->handle, &lo, &hi, &v[0].minimizer_idx, 1, &id, &is_new) != BRISK_HIP_OK)
                throw std::runtime_error(std::string("brisk_hip: ") + brisk_hip_last_error(big->handle));
            *fresh_slot(id) = *o.data;
        }
        delete menu;
        menu = big.release();
        params = grown;
        have_last_ = false;
        chunks_ = std::move(fresh_chunks);
        enum_cursor_ = 0;
        enum_pos_ = 0;
        e_lo_.clear();
        e_hi_.clear();
    }

  private:
    void check(int rc) const {
        if (rc != BRISK_HIP_OK) throw std::runtime_error(std::string("brisk_hip: ") + brisk_hip_last_error(menu->handle));
    }
    void flatten(const std::vector<kmer_full>& v) {
        lo_.resize(v.size());
        hi_.resize(v.size());
        idx_.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) {
            lo_[i] = (uint64_t)v[i].kmer_s;
            hi_[i] = (uint64_t)(v[i].kmer_s >> 64);
            idx_[i] = v[i].minimizer_idx;
        }
    }
    std::vector<DATA*> find(std::vector<kmer_full>& v) {
        std::vector<DATA*> result(v.size(), nullptr);
        if (v.empty()) return result;
        std::lock_guard<std::mutex> g(call_mu_);
        flatten(v);
        ids_.resize(v.size());
        check(brisk_hip_find_kmers(menu->handle, lo_.data(), hi_.data(), idx_.data(), v.size(), ids_.data()));
        for (size_t i = 0; i < v.size(); i++)
            if (ids_[i] != 0xffffffffu) result[i] = slot(ids_[i]);
        return result;
    }
    // host DATA store: fixed-size chunks so that pointers never move
    DATA* slot(uint32_t id) {
        const size_t c = id >> 16;
        while (chunks_.size() <= c) chunks_.emplace_back(new DATA[1u << 16]);
        return &chunks_[c][id & 0xffff];
    }
    mutable std::mutex call_mu_;  // the C-ABI is thread-compatible: one call at a time per handle
    std::mutex data_mu_;          // protect_data / unprotect_data
    std::vector<std::unique_ptr<DATA[]>> chunks_;
    std::vector<uint64_t> lo_, hi_, e_lo_, e_hi_;
    std::vector<uint8_t> idx_, new_, e_idx_;
    std::vector<uint32_t> ids_, e_ids_;
    uint64_t enum_cursor_;
    size_t enum_pos_;
    kint last_kmer_ = 0;  // what next() returned last, with its entry id (get() right behind next())
    uint8_t last_idx_ = 0;
    uint32_t last_id_ = 0;
    bool have_last_ = false;
};

#endif
