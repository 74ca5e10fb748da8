// brisk_count -- this repo's own small k-mer counter over the facade / C-ABI (not the reference's
// apps/counter.cpp, which compiles unchanged against brisk_amd/include and is built by the tests).
//   brisk_count --facade FASTA k m b [dump.txt]   per-call API: SuperKmerEnumerator + Brisk<uint8_t>
//   brisk_count --bulk   FASTA k m b [dump.txt]   bulk C-ABI: brisk_hip_insert_reads, FASTA or FASTA.gz streamed in batches
//   brisk_count --mixed  FASTA k m b              BASELINE config #5's protocol (apps/counter.cpp:197-227,314-346): one thread streams the
//                                                 file into brisk_hip_insert_reads while a second thread issues brisk_hip_get_reads on
//                                                 the first reads of batches that are already in; prints what every get saw
// Prints nb_kmers / nb_buckets / sum of counts; optionally dumps "KMER idx count" lines (dump.txt, "-" for none) and
// writes the index as a KFF file (a 7th argument: BriskWriter in --facade mode, brisk_write_kff in --bulk mode).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <atomic>
#include <chrono>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "Brisk.hpp"
#include "brisk_fasta.hpp"
#include "writer.hpp"

static std::vector<std::string> read_fasta(const char* path) {
    std::vector<std::string> out;
    FastaReader rd(path);
    FastaBatch b;
    b.clear();
    while (rd.fill(b, (size_t)1 << 26)) {
        if (rd.done()) break;
    }
    for (size_t i = 0; i < b.size(); i++) out.push_back(b.flat.substr(b.offs[i], b.offs[i + 1] - b.offs[i]));
    return out;
}

// --mixed: concurrent insert + get on ONE handle from two host threads (include/brisk_hip.h, "Threads").
// Output, one line each: "batch J N" (batch J of N reads is in), "get J s0 s1 ..." (the per-read sums a get of batch J's
// first reads returned while later batches were being inserted), "final J s0 ..." (the same get after the last batch),
// "digest ENTRIES SUM DIGEST" (brisk_hip_checksum of the final index).
static int run_mixed(const char* path, uint8_t k, uint8_t m, uint8_t b, const double* coef, size_t batch_bases) {
    const size_t sample_reads = getenv("BRISK_MIXED_SAMPLE") ? (size_t)atoll(getenv("BRISK_MIXED_SAMPLE")) : 50;
    brisk_hip_options o{};
    o.struct_size = sizeof o;
    brisk_hip_index* h = nullptr;
    if (brisk_hip_create(&h, k, m, b, 1, coef, &o) != BRISK_HIP_OK) return 1;
    struct Sample {
        std::string flat;
        std::vector<uint64_t> offs;
    };
    std::mutex mu;
    std::vector<Sample> samples;  // sample j = the first reads of batch j, published once batch j is in
    std::vector<std::string> log;
    std::atomic<bool> done{false}, failed{false};
    auto query = [&](size_t j, const char* tag) {
        Sample s;
        {
            std::lock_guard<std::mutex> g(mu);
            s = samples[j];
        }
        std::vector<uint64_t> sums(s.offs.size() - 1);
        if (brisk_hip_get_reads(h, s.flat.data(), s.offs.data(), sums.size(), sums.data()) != BRISK_HIP_OK) {
            failed = true;
            return;
        }
        std::string line = std::string(tag) + " " + std::to_string(j);
        for (uint64_t v : sums) line += " " + std::to_string(v);
        std::lock_guard<std::mutex> g(mu);
        log.push_back(line);
    };
    std::thread getter([&]() {  // keeps asking about batches that are in -- the newest one first, then all of them in turn
        size_t newest_asked = (size_t)-1, turn = 0, asked = 0;
        const size_t max_gets = getenv("BRISK_MIXED_MAX_GETS") ? (size_t)atoll(getenv("BRISK_MIXED_MAX_GETS")) : 2000;
        while (!done && !failed && asked < max_gets) {
            size_t have;
            {
                std::lock_guard<std::mutex> g(mu);
                have = samples.size();
            }
            if (have == 0) {
                std::this_thread::yield();
                continue;
            }
            size_t which = turn++ % have;
            if (have - 1 != newest_asked) which = newest_asked = have - 1;
            query(which, "get");
            asked++;
        }
    });
    FastaBatcher batches(path, batch_bases);
    FastaBatch bt;
    size_t j = 0;
    while (!failed && batches.next(bt)) {
        if (brisk_hip_insert_reads(h, bt.flat.data(), bt.offs.data(), bt.size()) != BRISK_HIP_OK) {
            std::cerr << brisk_hip_last_error(h) << std::endl;
            failed = true;
            break;
        }
        Sample s;
        const size_t ns = std::min(sample_reads, bt.size());
        s.flat = bt.flat.substr(0, bt.offs[ns]);
        s.offs.assign(bt.offs.begin(), bt.offs.begin() + ns + 1);
        std::lock_guard<std::mutex> g(mu);
        samples.push_back(s);
        log.push_back("batch " + std::to_string(j++) + " " + std::to_string(bt.size()));
    }
    done = true;
    getter.join();
    for (size_t q = 0; q < samples.size() && !failed; q++) query(q, "final");
    uint64_t ck[3] = {0, 0, 0};
    if (!failed && brisk_hip_checksum(h, ck) != BRISK_HIP_OK) failed = true;
    brisk_hip_destroy(h);
    for (const std::string& l : log) std::cout << l << "\n";
    std::cout << "digest " << ck[0] << " " << ck[1] << " " << ck[2] << std::endl;
    return failed ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 6) {
        std::cerr << "usage: brisk_count --facade|--bulk|--mixed FASTA k m b [dump.txt]" << std::endl;
        return 2;
    }
    const bool bulk = !strcmp(argv[1], "--bulk");
    const size_t batch_bases = getenv("BRISK_BATCH_BASES") ? (size_t)atoll(getenv("BRISK_BATCH_BASES")) : ((size_t)256 << 20);
    const bool mixed = !strcmp(argv[1], "--mixed");
    const std::vector<std::string> seqs = (bulk || mixed) ? std::vector<std::string>() : read_fasta(argv[2]);
    const uint8_t k = (uint8_t)atoi(argv[3]), m = (uint8_t)atoi(argv[4]), b = (uint8_t)atoi(argv[5]);
    const char* dump = argc > 6 && strcmp(argv[6], "-") ? argv[6] : nullptr;
    const char* kff = argc > 7 ? argv[7] : nullptr;
    Parameters params(k, m, b);
    if (mixed) return run_mixed(argv[2], k, m, b, params.dede->coef(), batch_bases);
    std::vector<std::string> lines;
    uint64_t nb_buckets = 0, nb_skmers = 0, nb_kmers = 0, mem = 0, largest = 0, sum = 0;
    try {
        if (!bulk) {
            Brisk<uint8_t> index(params);
            for (const std::string& s0 : seqs) {
                if (s0.size() < k) continue;
                std::string s(s0);
                SuperKmerEnumerator en(s, k, m, params.dede);
                std::vector<kmer_full> v;
                std::vector<bool> fresh;
                en.next(v);
                while (!v.empty()) {
                    fresh.clear();
                    index.protect_data(v[0]);
                    std::vector<uint8_t*> ptr = index.insert_superkmer(v, fresh);
                    for (size_t i = 0; i < ptr.size(); i++) {
                        if (fresh[i]) *ptr[i] = 1;
                        else ++*ptr[i];
                    }
                    index.unprotect_data(v[0]);
                    v.clear();
                    en.next(v);
                }
            }
            if (getenv("BRISK_REALLOCATE")) {  // Brisk::reallocate (brisk/Brisk.hpp:202-224): the same entries under (k, m + 2, b + 2)
                index.reallocate();
                std::cout << "reallocated k " << (int)index.params.k << " m " << (int)index.params.m << " b " << (int)index.params.b << std::endl;
            }
            kmer_full km((kint)0, 0, index.params.m, index.params.dede);
            bool first_entry = true;
            while (index.next(km)) {
                uint8_t* c = index.get(km);
                if (!c) {
                    std::cerr << "entry without data" << std::endl;
                    return 1;
                }
                if (first_entry) {  // Brisk::insert on a k-mer that is present behaves like get (README.md:49 of the reference)
                    first_entry = false;
                    if (index.insert(km) != c) {
                        std::cerr << "insert(kmer) of a present k-mer is not its get" << std::endl;
                        return 1;
                    }
                }
                sum += *c;
                if (dump) lines.push_back(kmer2str(km.kmer_s, k) + " " + std::to_string(km.minimizer_idx) + " " + std::to_string(*c));
            }
            index.stats(nb_buckets, nb_skmers, nb_kmers, mem, largest);
            if (kff) {  // apps/counter.cpp:407-411
                BriskWriter writer(kff);
                writer.write(index);
                writer.close();
            }
        } else {
            brisk_hip_options o{};
            o.struct_size = sizeof o;
            brisk_hip_index* h = nullptr;
            int rc = brisk_hip_create(&h, k, m, b, 1, params.dede->coef(), &o);
            if (rc != BRISK_HIP_OK) {
                std::cerr << "brisk_hip_create failed: " << rc << std::endl;
                return 1;
            }
            // stream the file: a background thread inflates and segments batch i+1 while the GPU counts batch i
            const bool e2e = getenv("BRISK_E2E_JSON") != nullptr;  // bench.py's end-to-end leg: the stage split as one JSON line
            if (e2e) brisk_hip_profile_enable(h, 1);
            const auto t_start = std::chrono::steady_clock::now();
            double insert_calls_s = 0.0;
            uint64_t n_reads_in = 0, n_bases_in = 0, n_batches = 0;
            {
                FastaBatcher batches(argv[2], batch_bases);
                FastaBatch bt;
                while (batches.next(bt)) {
                    const auto c0 = std::chrono::steady_clock::now();
                    rc = brisk_hip_insert_reads(h, bt.flat.data(), bt.offs.data(), bt.size());
                    insert_calls_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
                    if (rc != BRISK_HIP_OK) {
                        std::cerr << brisk_hip_last_error(h) << std::endl;
                        return 1;
                    }
                    n_reads_in += bt.size();
                    n_bases_in += bt.flat.size();
                    n_batches++;
                }
                const auto c0 = std::chrono::steady_clock::now();
                brisk_hip_sync(h);  // (completes deferred inserts: the index is whole when the clock stops)
                const double sync_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
                const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                brisk_hip_stats(h, &nb_buckets, &nb_skmers, &nb_kmers, &mem, &largest);
                if (e2e) {
                    uint32_t ns = 0;
                    const char* names[BRISK_HIP_PROFILE_SLOTS];
                    uint64_t launches[BRISK_HIP_PROFILE_SLOTS];
                    double ms[BRISK_HIP_PROFILE_SLOTS];
                    brisk_hip_profile_read(h, &ns, names, launches, ms);
                    std::cout << "E2E {\"wall_s\": " << wall_s << ", \"reads\": " << n_reads_in << ", \"bases\": " << n_bases_in << ", \"batches\": " << n_batches
                              << ", \"entries\": " << nb_kmers << ", \"reader_produce_s\": " << batches.produce_seconds() << ", \"main_waits_for_reader_s\": "
                              << batches.consumer_wait_seconds() << ", \"insert_reads_calls_s\": " << insert_calls_s << ", \"final_sync_s\": " << sync_s << ", \"library_ms\": {";
                    bool first = true;
                    for (uint32_t i = 0; i < ns; i++)
                        if (launches[i]) {
                            std::cout << (first ? "" : ", ") << "\"" << names[i] << "\": " << ms[i];
                            first = false;
                        }
                    std::cout << "}}" << std::endl;
                }
            }
            uint64_t cursor = 0, n = 0;
            const uint64_t cap = 1u << 20;
            std::vector<uint64_t> lo(cap), hi(cap);
            std::vector<uint8_t> idx(cap), cnt(cap);
            for (;;) {
                rc = brisk_hip_enumerate(h, &cursor, lo.data(), hi.data(), idx.data(), cnt.data(), cap, &n);
                if (rc != BRISK_HIP_OK || n == 0) break;
                for (uint64_t i = 0; i < n; i++) {
                    sum += cnt[i];
                    if (dump) lines.push_back(kmer2str(((kint)hi[i] << 64) | lo[i], k) + " " + std::to_string(idx[i]) + " " + std::to_string(cnt[i]));
                }
            }
            if (kff && brisk_write_kff(h, kff) != BRISK_HIP_OK) {
                std::cerr << "KFF: " << brisk_hip_last_error(h) << std::endl;
                return 1;
            }
            brisk_hip_destroy(h);
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "nb_kmers " << nb_kmers << " nb_buckets " << nb_buckets << " sum_counts " << sum << std::endl;
    if (dump) {
        std::sort(lines.begin(), lines.end());
        std::ofstream out(dump);
        for (auto& l : lines) out << l << "\n";
    }
    return 0;
}
