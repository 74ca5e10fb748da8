// brisk_count -- this repo's own small k-mer counter over the facade / C-ABI (not the reference's
// apps/counter.cpp, which compiles unchanged against brisk_amd/include and is built by the tests).
//   brisk_count --facade FASTA k m b [dump.txt]   per-call API: SuperKmerEnumerator + Brisk<uint8_t>
//   brisk_count --bulk   FASTA k m b [dump.txt]   bulk C-ABI: brisk_hip_insert_reads, FASTA or FASTA.gz streamed in batches
// Prints nb_kmers / nb_buckets / sum of counts; optionally dumps "KMER idx count" lines.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "Brisk.hpp"
#include "brisk_fasta.hpp"

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

int main(int argc, char** argv) {
    if (argc < 6) {
        std::cerr << "usage: brisk_count --facade|--bulk FASTA k m b [dump.txt]" << std::endl;
        return 2;
    }
    const bool bulk = !strcmp(argv[1], "--bulk");
    const size_t batch_bases = getenv("BRISK_BATCH_BASES") ? (size_t)atoll(getenv("BRISK_BATCH_BASES")) : ((size_t)256 << 20);
    const std::vector<std::string> seqs = bulk ? std::vector<std::string>() : read_fasta(argv[2]);
    const uint8_t k = (uint8_t)atoi(argv[3]), m = (uint8_t)atoi(argv[4]), b = (uint8_t)atoi(argv[5]);
    const char* dump = argc > 6 ? argv[6] : nullptr;
    Parameters params(k, m, b);
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
            kmer_full km((kint)0, 0, m, params.dede);
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
            FastaBatcher batches(argv[2], batch_bases);
            FastaBatch bt;
            while (batches.next(bt)) {
                rc = brisk_hip_insert_reads(h, bt.flat.data(), bt.offs.data(), bt.size());
                if (rc != BRISK_HIP_OK) {
                    std::cerr << brisk_hip_last_error(h) << std::endl;
                    return 1;
                }
            }
            brisk_hip_stats(h, &nb_buckets, &nb_skmers, &nb_kmers, &mem, &largest);
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
