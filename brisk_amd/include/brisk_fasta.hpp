// brisk_fasta.hpp -- host FASTA / FASTA.gz front-end for the bulk path.
// Same record and segment rules as the reference's harness (apps/counter.cpp:130-190):
//   * a record is its first line (the header, whatever it holds) plus every following line up
//     to the next line that starts with '>', concatenated;
//   * the record is cut at every character outside [ACGTacgt]; each maximal run of valid bases
//     is a sequence of its own; sequences are upper-cased.
// Input is inflated with zlib (gzread reads plain files transparently, as zstr does for the
// reference).  FastaBatcher parses in the background while the GPU counts the previous batch (the
// reference serialises parsing in one omp critical section, counter.cpp:217-220); a plain (not
// gzipped) file is memory-mapped and every batch is parsed by several threads, each on a run of
// whole records.
#ifndef BRISK_AMD_FASTA_HPP
#define BRISK_AMD_FASTA_HPP
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct FastaBatch {
    std::string flat;            // concatenated sequences
    std::vector<uint64_t> offs;  // offs[n+1]
    size_t size() const { return offs.empty() ? 0 : offs.size() - 1; }
    void clear() {
        flat.clear();
        offs.assign(1, 0);
    }
};

class FastaReader {
  public:
    explicit FastaReader(const std::string& path) : gz_(gzopen(path.c_str(), "rb")), pos_(0), len_(0), eof_(false), in_header_(true), at_line_start_(true) {
        if (!gz_) throw std::runtime_error("cannot open " + path);
        gzbuffer(gz_, 1 << 20);
        buf_.resize(8 << 20);  // zstr's buffer size (zstr.hpp:222)
    }
    ~FastaReader() {
        if (gz_) gzclose(gz_);
    }
    FastaReader(const FastaReader&) = delete;
    FastaReader& operator=(const FastaReader&) = delete;

    // Appends sequences to `out` until it holds at least max_bases bases or the file ends.
    // Returns false when nothing was appended and the file is exhausted.
    bool fill(FastaBatch& out, size_t max_bases) {
        if (out.offs.empty()) out.offs.assign(1, 0);
        const size_t before = out.size();
        for (;;) {
            if (pos_ == len_ && !refill()) break;
            // one line (or what the buffer holds of it) per turn: memchr for its end, whole runs of bases appended at once
            const char* p = &buf_[pos_];
            const char* end = &buf_[0] + len_;
            if (at_line_start_) {
                at_line_start_ = false;
                if (*p == '>' && !in_header_) {  // next record: close the running sequence
                    close_run(out);
                    in_header_ = true;
                }
            }
            // A full batch ends where no sequence is running -- in front of the next record's header, or behind a
            // character that cut the sequence.  (Checked here, after the header of the next record has closed the run:
            // with one sequence line per record, as in a file of reads, the run is open at the end of EVERY line, and a
            // check at the end of the line would never let a batch end before the end of the file.)
            if (out.flat.size() >= max_bases && out.flat.size() == out.offs.back()) break;
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* stop = nl ? nl : end;
            if (!in_header_) append_bases(out, p, stop);
            pos_ = (size_t)(stop - &buf_[0]);
            if (nl) {
                pos_++;
                at_line_start_ = true;
                in_header_ = false;  // the first line of a record is its header
            }
        }
        if (eof_ && pos_ == len_) close_run(out);
        return out.size() > before || !(eof_ && pos_ == len_);
    }
    bool done() const { return eof_ && pos_ == len_; }

    // Whole records in memory: [p, end) starts at the first character of a record's header line (or of the file,
    // whose first line is a header whatever it holds) and ends where the next record starts (or at the end of the file).
    static void parse_records(const char* p, const char* end, FastaBatch& out) {
        if (out.offs.empty()) out.offs.assign(1, 0);
        bool in_header = true;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* stop = nl ? nl : end;
            if (!in_header) {
                if (*p == '>') {  // a line that starts with '>' begins the next record
                    if (out.flat.size() > out.offs.back()) out.offs.push_back(out.flat.size());
                    in_header = true;
                } else {
                    append_bases(out, p, stop);
                }
            }
            if (!nl) break;
            in_header = false;  // the first line of a record is its header; the lines after it are not
            p = nl + 1;
        }
        if (out.flat.size() > out.offs.back()) out.offs.push_back(out.flat.size());
    }
    // first position >= p where a record starts (a '>' at the start of a line), or end
    static const char* next_record(const char* base, const char* p, const char* end) {
        if (p <= base) return base;
        const char* q = p - 1;  // look for "\n>" with the '>' at or after p
        for (;;) {
            q = (const char*)memchr(q, '\n', (size_t)(end - q));
            if (!q || q + 1 >= end) return end;
            if (q[1] == '>') return q + 1;
            q++;
        }
    }

  private:
    static void close_run(FastaBatch& out) {
        if (out.flat.size() > out.offs.back()) out.offs.push_back(out.flat.size());
    }
    // [p, stop) holds no newline: maximal runs of [ACGTacgt] are appended upper-cased, anything else closes the run.
    // One pass: room for the whole stretch is made first, a 256-entry table maps a byte to its upper-case base or 0.
    struct BaseTable {
        unsigned char t[256];
        BaseTable() {
            std::memset(t, 0, sizeof t);
            for (const char* c = "ACGT"; *c; c++) t[(unsigned char)*c] = t[(unsigned char)(*c | 0x20)] = (unsigned char)*c;
        }
    };
    static void append_bases(FastaBatch& out, const char* p, const char* stop) {
        static const BaseTable tab;
        size_t o = out.flat.size();
        out.flat.resize(o + (size_t)(stop - p));
        char* d = &out.flat[0];
        for (; p < stop; p++) {
            const unsigned char u = tab.t[(unsigned char)*p];
            if (u) {
                d[o++] = (char)u;
            } else if (o > out.offs.back()) {
                out.offs.push_back(o);
            }
        }
        out.flat.resize(o);
    }
    bool refill() {
        if (eof_) return false;
        const int n = gzread(gz_, &buf_[0], (unsigned)buf_.size());
        if (n < 0) throw std::runtime_error("gzread failed");
        pos_ = 0;
        len_ = (size_t)n;
        if (n == 0) eof_ = true;
        return n > 0;
    }
    gzFile gz_;
    std::string buf_;
    size_t pos_, len_;
    bool eof_, in_header_, at_line_start_;
};

// Double-buffered batches: a background thread parses batch i+1 while the caller consumes batch i.
class FastaBatcher {
  public:
    FastaBatcher(const std::string& path, size_t batch_bases, unsigned n_threads = 0)
        : reader_(path), batch_bases_(batch_bases), ready_(false), finished_(false), stop_(false), map_(nullptr), map_len_(0) {
        n_threads_ = n_threads ? n_threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        map_plain_file(path);
        worker_ = std::thread([this] { map_ ? run_mapped() : run(); });
    }
    ~FastaBatcher() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        if (map_) munmap((void*)map_, map_len_);
    }
    // seconds the worker spent producing batches (inflate + parse; the time it waited for its slot to be taken is not in it), and
    // seconds next() made its caller wait for a batch
    double produce_seconds() const { return produce_s_; }
    double consumer_wait_seconds() const { return wait_s_; }
    // Moves the next batch into `out`; false when the file is exhausted.
    bool next(FastaBatch& out) {
        const auto w0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return ready_ || finished_; });
        wait_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (!ready_) {
            if (!error_.empty()) throw std::runtime_error(error_);
            return false;
        }
        std::swap(out, slot_);
        ready_ = false;
        lk.unlock();
        cv_.notify_all();
        return true;
    }

  private:
    void run() {
        try {
            FastaBatch b;  // buffers circulate: what next() swapped back into the slot is filled again (no fresh pages per batch)
            for (;;) {
                const auto p0 = std::chrono::steady_clock::now();
                b.clear();
                if (b.flat.capacity() < batch_bases_) b.flat.reserve(batch_bases_ + (batch_bases_ >> 4) + (1 << 20));
                // a batch may end inside a sequence only at the end of the file: keep reading until a run closes
                bool more = reader_.fill(b, batch_bases_);
                while (more && !reader_.done() && b.flat.size() > b.offs.back()) more = reader_.fill(b, b.flat.size() + (1 << 16));
                if (b.flat.size() > b.offs.back()) b.offs.push_back(b.flat.size());
                produce_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !ready_ || stop_; });
                if (stop_) return;
                if (b.size() > 0) {
                    std::swap(slot_, b);
                    ready_ = true;
                }
                if (reader_.done()) {
                    finished_ = true;
                    lk.unlock();
                    cv_.notify_all();
                    return;
                }
                lk.unlock();
                cv_.notify_all();
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> g(mu_);
            error_ = e.what();
            finished_ = true;
            cv_.notify_all();
        }
    }
    // a file that does not start with the gzip magic is mapped; anything else goes through zlib
    void map_plain_file(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = (const char*)m;
                map_len_ = (size_t)st.st_size;
                madvise(m, map_len_, MADV_SEQUENTIAL);
            }
        }
        close(fd);
    }
    // hand a finished batch to next(); false when the consumer is gone
    bool publish(FastaBatch& b, bool last) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !ready_ || stop_; });
        if (stop_) return false;
        if (b.size() > 0) {
            std::swap(slot_, b);
            ready_ = true;
        }
        if (last) finished_ = true;
        lk.unlock();
        cv_.notify_all();
        return true;
    }
    // Plain file: every batch is a run of whole records of about batch_bases_ bytes, cut into n_threads_ runs of whole
    // records that are parsed side by side and then copied, side by side again, into one batch.
    void run_mapped() {
        try {
            const char* base = map_;
            const char* end = map_ + map_len_;
            const char* pos = base;
            std::vector<FastaBatch> part(n_threads_);
            FastaBatch b;
            while (pos < end) {
                const auto p0 = std::chrono::steady_clock::now();
                const size_t want = batch_bases_ + (batch_bases_ >> 4) + 1;
                const char* stop = (size_t)(end - pos) <= want ? end : FastaReader::next_record(base, pos + want, end);
                // cut points at record starts
                std::vector<const char*> cut(n_threads_ + 1, stop);
                cut[0] = pos;
                for (unsigned t = 1; t < n_threads_; t++) {
                    const char* c = FastaReader::next_record(base, pos + (size_t)(stop - pos) / n_threads_ * t, stop);
                    cut[t] = std::max(c, cut[t - 1]);
                }
                std::vector<std::thread> th;
                for (unsigned t = 0; t < n_threads_; t++)
                    th.emplace_back([&, t] {
                        part[t].clear();
                        if (cut[t] < cut[t + 1]) FastaReader::parse_records(cut[t], cut[t + 1], part[t]);
                    });
                for (auto& x : th) x.join();
                std::vector<size_t> fo(n_threads_ + 1, 0), so(n_threads_ + 1, 0);
                for (unsigned t = 0; t < n_threads_; t++) {
                    fo[t + 1] = fo[t] + part[t].flat.size();
                    so[t + 1] = so[t] + part[t].size();
                }
                b.flat.resize(fo[n_threads_]);
                b.offs.resize(so[n_threads_] + 1);
                b.offs[0] = 0;
                th.clear();
                for (unsigned t = 0; t < n_threads_; t++)
                    th.emplace_back([&, t] {
                        if (!part[t].flat.empty()) std::memcpy(&b.flat[fo[t]], part[t].flat.data(), part[t].flat.size());
                        for (size_t i = 0; i < part[t].size(); i++) b.offs[so[t] + i + 1] = fo[t] + part[t].offs[i + 1];
                    });
                for (auto& x : th) x.join();
                pos = stop;
                produce_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
                if (!publish(b, pos >= end)) return;
                b.clear();
            }
            std::lock_guard<std::mutex> g(mu_);
            finished_ = true;
            cv_.notify_all();
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> g(mu_);
            error_ = e.what();
            finished_ = true;
            cv_.notify_all();
        }
    }
    FastaReader reader_;
    size_t batch_bases_;
    unsigned n_threads_;
    const char* map_;
    size_t map_len_;
    FastaBatch slot_;
    double produce_s_ = 0.0, wait_s_ = 0.0;
    bool ready_, finished_, stop_;
    std::string error_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread worker_;
};

#endif
