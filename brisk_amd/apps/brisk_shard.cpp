// brisk_shard -- the multi-GPU counting job from C++ (north_star: "host code stays C++"): one process per GPU, reads sharded
// by index, super-k-mer records routed to the owner of their bucket range with ONE all-to-all over RCCL (counts first,
// then the payload -- records and the scan's per-partition counts in one group of ncclSend/ncclRecv), insert purely local.  The Python twin of this flow is
// brisk_amd/exchange.py (torch.distributed); both sit on the same C-ABI calls: brisk_hip_scan_packed -> route_records ->
// export_hist -> [exchange] -> insert_records_hist.  The reference is single-process: nothing here has a counterpart in it.
//
//   brisk_shard RANK WORLD DIR TRANSPORT TOTAL_READS k m b [coverage]
//     DIR        a directory all ranks see (rank 0 leaves the ncclUniqueId there; the "files" transport exchanges through it).
//                Every file carries the run's token (environment BRISK_SHARD_RUN, the same for all ranks of a run; "" if unset)
//                and is removed by its last reader, so a second run in the same directory never meets the first one's id or
//                payloads -- also not those a crashed run left behind, as long as launchers give every run its own token
//     TRANSPORT  rccl : device = RANK, ncclSend/ncclRecv between device buffers over xGMI
//                files: every rank on device 0, buffers staged through DIR -- the rehearsal of the N > 1 flow on a one-GPU
//                       box, where RCCL refuses two ranks on one device (what "gloo --share-gpu" is to bench.py)
//   Input: the bench's synthetic reads, strong scaling: rank r scans reads [r, r+1) * TOTAL / WORLD.
//   Output (every rank): "rank R entries E sum_counts S digest D ms T"; digests of the shards add up to the single-index one.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "Decycling.h"
#include "brisk_hip.h"

#define HIPOK(call)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::cerr << "rank " << g_rank << ": " #call ": " << hipGetErrorString(e_) << std::endl; \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)
#define NCCLOK(call)                                                                          \
    do {                                                                                      \
        ncclResult_t e_ = (call);                                                             \
        if (e_ != ncclSuccess) {                                                              \
            std::cerr << "rank " << g_rank << ": " #call ": " << ncclGetErrorString(e_) << std::endl; \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)
#define BRISKOK(h, call)                                                                      \
    do {                                                                                      \
        int e_ = (call);                                                                      \
        if (e_ != BRISK_HIP_OK) {                                                             \
            std::cerr << "rank " << g_rank << ": " #call ": " << e_ << " " << brisk_hip_last_error(h) << std::endl; \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

static int g_rank = 0;

static bool exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}
static void wait_for(const std::string& p) {
    for (int i = 0; !exists(p); i++) {
        if (i > 600000) {
            std::cerr << "rank " << g_rank << ": timed out waiting for " << p << std::endl;
            exit(1);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}
static void publish(const std::string& p, const void* data, size_t bytes) {  // write under a temporary name, then rename: readers never see half a file
    const std::string tmp = p + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || (bytes && fwrite(data, 1, bytes, f) != bytes) || fclose(f) != 0 || rename(tmp.c_str(), p.c_str()) != 0) {
        std::cerr << "rank " << g_rank << ": cannot write " << p << std::endl;
        exit(1);
    }
}
static void slurp(const std::string& p, void* data, size_t bytes, bool last_reader = true) {
    wait_for(p);
    FILE* f = fopen(p.c_str(), "rb");
    if (!f || (bytes && fread(data, 1, bytes, f) != bytes)) {
        std::cerr << "rank " << g_rank << ": cannot read " << p << std::endl;
        exit(1);
    }
    fclose(f);
    if (last_reader) unlink(p.c_str());  // every step file has exactly one reader
}

// The exchange step: every rank hands `send_counts[d]` items of `words` u64 each to rank d (its send buffer is grouped by
// destination) and receives `recv_counts[s]` items from rank s into `recv` (grouped by source).  Device buffers.
struct Transport {
    int rank, world;
    std::string dir;
    bool rccl;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int step = 0;
    std::string run;  // BRISK_SHARD_RUN: part of every file name

    std::string path(const char* kind, int a, int b) const { return dir + "/" + run + kind + std::to_string(step) + "_" + std::to_string(a) + "_" + std::to_string(b); }
    void init() {
        if (const char* t = getenv("BRISK_SHARD_RUN")) run = std::string(t) + "_";
        if (!rccl) return;
        ncclUniqueId id;
        const std::string f = dir + "/" + run + "nccl_id";
        if (rank == 0) {
            NCCLOK(ncclGetUniqueId(&id));
            publish(f, &id, sizeof id);
        } else {
            slurp(f, &id, sizeof id, /*last_reader=*/false);  // world - 1 readers: rank 0 removes it once every rank holds the communicator
        }
        NCCLOK(ncclCommInitRank(&comm, world, id, rank));
        barrier();
        if (rank == 0) unlink(f.c_str());
    }
    // counts first: how many items every rank is about to send me
    std::vector<uint64_t> exchange_counts(const std::vector<uint64_t>& send_counts) {
        std::vector<uint64_t> recv_counts(world, 0);
        step++;
        if (rccl) {
            uint64_t *d_s, *d_r;
            HIPOK(hipMalloc((void**)&d_s, world * 8));
            HIPOK(hipMalloc((void**)&d_r, world * 8));
            HIPOK(hipMemcpyAsync(d_s, send_counts.data(), world * 8, hipMemcpyHostToDevice, stream));
            NCCLOK(ncclGroupStart());
            for (int p = 0; p < world; p++) {
                NCCLOK(ncclSend(d_s + p, 1, ncclUint64, p, comm, stream));
                NCCLOK(ncclRecv(d_r + p, 1, ncclUint64, p, comm, stream));
            }
            NCCLOK(ncclGroupEnd());
            HIPOK(hipMemcpyAsync(recv_counts.data(), d_r, world * 8, hipMemcpyDeviceToHost, stream));
            HIPOK(hipStreamSynchronize(stream));
            HIPOK(hipFree(d_s));
            HIPOK(hipFree(d_r));
        } else {
            for (int p = 0; p < world; p++) publish(path("c", rank, p), &send_counts[p], 8);
            for (int p = 0; p < world; p++) slurp(path("c", p, rank), &recv_counts[p], 8);
        }
        return recv_counts;
    }
    // then the payload
    void exchange(const uint64_t* d_send, const std::vector<uint64_t>& send_counts, uint64_t* d_recv, const std::vector<uint64_t>& recv_counts, uint64_t words) {
        step++;
        if (rccl) {
            NCCLOK(ncclGroupStart());
            uint64_t so = 0, ro = 0;
            for (int p = 0; p < world; p++) {
                if (send_counts[p]) NCCLOK(ncclSend(d_send + so * words, send_counts[p] * words, ncclUint64, p, comm, stream));
                if (recv_counts[p]) NCCLOK(ncclRecv(d_recv + ro * words, recv_counts[p] * words, ncclUint64, p, comm, stream));
                so += send_counts[p];
                ro += recv_counts[p];
            }
            NCCLOK(ncclGroupEnd());
            HIPOK(hipStreamSynchronize(stream));
        } else {
            uint64_t so = 0, ro = 0;
            std::vector<uint64_t> buf;
            for (int p = 0; p < world; p++) {
                buf.resize(send_counts[p] * words);
                if (!buf.empty()) HIPOK(hipMemcpy(buf.data(), d_send + so * words, buf.size() * 8, hipMemcpyDeviceToHost));
                publish(path("p", rank, p), buf.data(), buf.size() * 8);
                so += send_counts[p];
            }
            for (int p = 0; p < world; p++) {
                buf.resize(recv_counts[p] * words);
                slurp(path("p", p, rank), buf.data(), buf.size() * 8);
                if (!buf.empty()) HIPOK(hipMemcpy(d_recv + ro * words, buf.data(), buf.size() * 8, hipMemcpyHostToDevice));
                ro += recv_counts[p];
            }
        }
    }
    // records AND histogram slices in one exchange step: with RCCL one group of sends and receives (to every peer its records and the
    // slice of its partition range, from every peer mine), i.e. one launch on the wire where there used to be two collectives
    void exchange_with_slices(const uint64_t* d_send, const std::vector<uint64_t>& send_counts, uint64_t* d_recv, const std::vector<uint64_t>& recv_counts, uint64_t words,
                              const uint64_t* d_hist, const std::vector<uint64_t>& lens, uint64_t* d_slices, uint64_t my_len) {
        if (!rccl) {
            exchange(d_send, send_counts, d_recv, recv_counts, words);
            exchange(d_hist, lens, d_slices, std::vector<uint64_t>(world, my_len), 1);
            return;
        }
        step++;
        NCCLOK(ncclGroupStart());
        uint64_t so = 0, ro = 0, ho = 0;
        for (int p = 0; p < world; p++) {
            if (send_counts[p]) NCCLOK(ncclSend(d_send + so * words, send_counts[p] * words, ncclUint64, p, comm, stream));
            if (recv_counts[p]) NCCLOK(ncclRecv(d_recv + ro * words, recv_counts[p] * words, ncclUint64, p, comm, stream));
            if (lens[p]) NCCLOK(ncclSend(d_hist + ho, lens[p], ncclUint64, p, comm, stream));
            if (my_len) NCCLOK(ncclRecv(d_slices + (uint64_t)p * my_len, my_len, ncclUint64, p, comm, stream));
            so += send_counts[p];
            ro += recv_counts[p];
            ho += lens[p];
        }
        NCCLOK(ncclGroupEnd());
        HIPOK(hipStreamSynchronize(stream));
    }
    void barrier() {
        std::vector<uint64_t> one(world, 1);
        exchange_counts(one);
    }
    void finish() {
        if (comm) ncclCommDestroy(comm);
    }
};

int main(int argc, char** argv) {
    if (argc < 9) {
        std::cerr << "usage: brisk_shard RANK WORLD DIR rccl|files TOTAL_READS k m b [coverage]" << std::endl;
        return 2;
    }
    const int rank = g_rank = atoi(argv[1]), world = atoi(argv[2]);
    const std::string dir = argv[3];
    const bool rccl = !strcmp(argv[4], "rccl");
    const uint64_t total = strtoull(argv[5], nullptr, 10);
    const uint8_t k = (uint8_t)atoi(argv[6]), m = (uint8_t)atoi(argv[7]), b = (uint8_t)atoi(argv[8]);
    const double coverage = argc > 9 ? atof(argv[9]) : 15.0;
    const uint32_t L = 150;
    if (rank < 0 || rank >= world || world < 1) return 2;
    const int device = rccl ? rank : 0;
    HIPOK(hipSetDevice(device));
    hipStream_t stream;
    HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    DecyclingSet dede(m);
    brisk_hip_options o{};
    o.struct_size = sizeof o;
    o.device = device;
    o.stream = stream;
    o.owner_rank = (uint32_t)rank;
    o.n_owners = (uint32_t)world;
    // about three reads of the batch per partition (include/brisk_hip.h, brisk_hip_options.part_bits; the same rule as
    // brisk_amd/exchange.py, suggest_part_bits): beyond the default 2^24 only for jobs of more than ~70 M reads
    if (world > 1) {
        const uint32_t bits = std::min<uint32_t>((uint32_t)std::lround(std::log2((double)std::max<uint64_t>(total, 3) / 3.0)), 2u * b);
        if (bits > 24) o.part_bits = bits;
    }
    brisk_hip_index* h = nullptr;
    int rc = brisk_hip_create(&h, k, m, b, 1, dede.coef(), &o);
    if (rc != BRISK_HIP_OK) {
        std::cerr << "rank " << rank << ": brisk_hip_create failed: " << rc << std::endl;
        return 1;
    }
    brisk_hip_layout lay{};
    BRISKOK(h, brisk_hip_get_layout(h, &lay));
    const uint64_t W = lay.record_words, n_parts = 1ull << lay.part_bits;

    Transport tp{rank, world, dir, rccl};
    tp.stream = stream;
    tp.init();

    // this rank's contiguous share of the job's reads, generated on its device (below); ownership first
    const uint64_t first = total * (uint64_t)rank / (uint64_t)world, n_reads = total * (uint64_t)(rank + 1) / (uint64_t)world - first;
    const uint64_t genome = std::max<uint64_t>((uint64_t)(total * (double)L / coverage), L + 1);
    uint32_t* d_packed;
    uint64_t* d_starts;
    // (the fills go on the index's stream: a fill on the null stream is not ordered with a non-blocking stream, and one that lands
    // behind brisk_hip_synth_reads leaves a rank with reads of length zero -- seen as an intermittent short count in round 3's tests)
    HIPOK(hipMalloc((void**)&d_packed, ((n_reads * L + 15) / 16 + 4) * 4));
    HIPOK(hipMemsetAsync(d_packed, 0, ((n_reads * L + 15) / 16 + 4) * 4, stream));
    HIPOK(hipMalloc((void**)&d_starts, (n_reads + 1) * 8));
    HIPOK(hipMemsetAsync(d_starts, 0, (n_reads + 1) * 8, stream));
    HIPOK(hipStreamSynchronize(stream));
    if (n_reads) BRISKOK(h, brisk_hip_synth_reads(h, genome, first, n_reads, L, 1, 2, d_packed, d_starts));
    BRISKOK(h, brisk_hip_sync(h));
    tp.barrier();

    // Ownership: equal partition ranges, unless a scan of this rank's first reads shows the most loaded owner more
    // than 1.3x above the mean: then every rank installs the same histogram-balanced cut points (brisk_hip_set_owner_cuts; the twin of
    // brisk_amd/exchange.py: ShardedCounter.balance).  2^14 block sums of k-mer instances per rank are all the ranks exchange.
    // BRISK_SHARD_BALANCE=0 keeps equal ranges.  Once per job, before the clock starts.
    if (world > 1 && !(getenv("BRISK_SHARD_BALANCE") && atoi(getenv("BRISK_SHARD_BALANCE")) == 0)) {
        const uint32_t cb = std::min<uint32_t>(14, lay.part_bits);
        const uint64_t nblk = 1ull << cb, per = n_parts >> cb;
        const uint64_t sample = std::min<uint64_t>(n_reads, 2000000);
        std::vector<uint64_t> mine(nblk, 0);
        if (sample) {
            uint64_t cap_s = 0, n_s = 0;
            BRISKOK(h, brisk_hip_scan_bound(h, d_starts, sample, &cap_s));
            uint64_t *d_r, *d_h;
            HIPOK(hipMalloc((void**)&d_r, (cap_s + 1) * W * 8));
            HIPOK(hipMalloc((void**)&d_h, n_parts * 8));
            BRISKOK(h, brisk_hip_scan_packed(h, d_packed, d_starts, sample, d_r, cap_s, &n_s));
            std::vector<uint64_t> ignore(world);
            BRISKOK(h, brisk_hip_export_hist(h, d_h, ignore.data()));
            std::vector<uint64_t> hh(n_parts);
            HIPOK(hipMemcpy(hh.data(), d_h, n_parts * 8, hipMemcpyDeviceToHost));
            for (uint64_t p = 0; p < n_parts; p++) mine[p / per] += hh[p] >> 32;
            HIPOK(hipFree(d_r));
            HIPOK(hipFree(d_h));
        }
        // all-gather of the block sums through the exchange step, then the same arithmetic on every rank
        uint64_t *d_m, *d_all;
        HIPOK(hipMalloc((void**)&d_m, (uint64_t)world * nblk * 8));
        HIPOK(hipMalloc((void**)&d_all, (uint64_t)world * nblk * 8));
        for (int p = 0; p < world; p++) HIPOK(hipMemcpy(d_m + (uint64_t)p * nblk, mine.data(), nblk * 8, hipMemcpyHostToDevice));
        tp.exchange(d_m, std::vector<uint64_t>(world, nblk), d_all, std::vector<uint64_t>(world, nblk), 1);
        std::vector<uint64_t> all((uint64_t)world * nblk);
        HIPOK(hipMemcpy(all.data(), d_all, all.size() * 8, hipMemcpyDeviceToHost));
        HIPOK(hipFree(d_m));
        HIPOK(hipFree(d_all));
        std::vector<double> cum(nblk + 1, 0.0);
        for (uint64_t i = 0; i < nblk; i++) {
            double v = 0;
            for (int p = 0; p < world; p++) v += (double)all[(uint64_t)p * nblk + i];
            cum[i + 1] = cum[i] + v;
        }
        const double tot = cum[nblk];
        double worst = 0;
        for (int o = 0; o < world; o++) {  // equal ranges: the smallest p with p * N >> part_bits == o
            const uint64_t lo = (((uint64_t)o << lay.part_bits) + world - 1) / world, hi = (((uint64_t)(o + 1) << lay.part_bits) + world - 1) / world;
            worst = std::max(worst, cum[hi / per] - cum[lo / per]);
        }
        if (tot > 0 && worst * world / tot > 1.3) {
            std::vector<uint64_t> cuts(world + 1, 0);
            for (int o = 1; o < world; o++) {
                const double want = tot * o / world;
                uint64_t i = (uint64_t)(std::lower_bound(cum.begin() + 1, cum.end(), want) - cum.begin());  // cum[i] >= want
                if (i > 1 && std::fabs(cum[i - 1] - want) <= std::fabs(cum[std::min<uint64_t>(i, nblk)] - want)) i--;
                cuts[o] = std::max<uint64_t>(cuts[o - 1], std::min<uint64_t>(i, nblk) * per);
            }
            cuts[world] = n_parts;
            BRISKOK(h, brisk_hip_set_owner_cuts(h, cuts.data()));
            if (rank == 0) std::cerr << "brisk_shard: equal ranges carry " << worst * world / tot << "x the mean: histogram-balanced cut points installed" << std::endl;
        }
        tp.barrier();
    }

    const auto t0 = std::chrono::steady_clock::now();
    // scan -> records; route them by owner; the scan's per-partition histogram travels with them
    uint64_t cap = 0, n_rec = 0;
    BRISKOK(h, brisk_hip_scan_bound(h, d_starts, n_reads, &cap));
    cap = std::min<uint64_t>(cap, 6 * n_reads + 4096) + 1;
    uint64_t *d_rec, *d_out, *d_hist;
    HIPOK(hipMalloc((void**)&d_rec, cap * W * 8));
    HIPOK(hipMalloc((void**)&d_out, cap * W * 8));
    HIPOK(hipMalloc((void**)&d_hist, n_parts * 8));
    rc = brisk_hip_scan_packed(h, d_packed, d_starts, n_reads, d_rec, cap, &n_rec);
    if (rc == BRISK_HIP_ECAPACITY) {  // more super-k-mers per read than the first guess: the exact bound
        BRISKOK(h, brisk_hip_scan_bound(h, d_starts, n_reads, &cap));
        HIPOK(hipFree(d_rec));
        HIPOK(hipFree(d_out));
        HIPOK(hipMalloc((void**)&d_rec, (cap + 1) * W * 8));
        HIPOK(hipMalloc((void**)&d_out, (cap + 1) * W * 8));
        rc = brisk_hip_scan_packed(h, d_packed, d_starts, n_reads, d_rec, cap, &n_rec);
    }
    BRISKOK(h, rc);
    std::vector<uint64_t> send_counts(world), lens(world);
    BRISKOK(h, brisk_hip_route_records(h, d_rec, n_rec, d_out, send_counts.data()));
    BRISKOK(h, brisk_hip_export_hist(h, d_hist, lens.data()));
    // the all-to-all: counts, then ONE payload exchange -- the records and the histogram slices (every rank sends owner d the slice of d's range)
    const std::vector<uint64_t> recv_counts = tp.exchange_counts(send_counts);
    uint64_t n_in = 0;
    for (uint64_t c : recv_counts) n_in += c;
    uint64_t *d_inbox, *d_slices;
    HIPOK(hipMalloc((void**)&d_inbox, (n_in + 1) * W * 8));
    const uint64_t my_len = lens[rank];
    HIPOK(hipMalloc((void**)&d_slices, ((uint64_t)world * my_len + 1) * 8));
    tp.exchange_with_slices(d_out, send_counts, d_inbox, recv_counts, W, d_hist, lens, d_slices, my_len);
    // insert what this rank owns
    BRISKOK(h, brisk_hip_insert_records_hist(h, d_inbox, n_in, d_slices, (uint32_t)world));
    BRISKOK(h, brisk_hip_sync(h));
    tp.barrier();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    uint64_t ck[3] = {0, 0, 0};
    BRISKOK(h, brisk_hip_checksum(h, ck));
    std::cout << "rank " << rank << " entries " << ck[0] << " sum_counts " << ck[1] << " digest " << ck[2] << " ms " << ms << " records_out " << n_rec << " records_in " << n_in
              << std::endl;
    tp.finish();
    brisk_hip_destroy(h);
    return 0;
}
