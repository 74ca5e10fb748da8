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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#define BRISK_FASTA_AVX2 1
#endif

// The bytes of a batch: the subset of std::string the readers and their callers use, with a resize() that does NOT fill what it
// adds (a fresh 300 MB batch was zero-filled by one thread, 150-200 ms, before the parser threads wrote it; now the parser
// threads are the first to touch its pages, side by side).
class FastaBytes {
  public:
    FastaBytes() = default;
    FastaBytes(const FastaBytes&) = delete;
    FastaBytes& operator=(const FastaBytes&) = delete;
    FastaBytes(FastaBytes&& o) noexcept { swap(o); }
    FastaBytes& operator=(FastaBytes&& o) noexcept {
        swap(o);
        return *this;
    }
    ~FastaBytes() { release(p_, cap_, mapped_); }
    void swap(FastaBytes& o) noexcept {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        std::swap(cap_, o.cap_);
        std::swap(mapped_, o.mapped_);
    }
    const char* data() const { return p_; }
    char* data() { return p_; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    char& operator[](size_t i) { return p_[i]; }
    const char& operator[](size_t i) const { return p_[i]; }
    void clear() { n_ = 0; }
    // A large buffer is a mapping of its own that asks for transparent huge pages: a batch's first touch is 150 faults of 2 MiB
    // instead of 75,000 of 4 KiB (which cost more than parsing the batch).
    void reserve(size_t c) {
        if (c <= cap_) return;
        c = std::max(c, cap_ + (cap_ >> 1));
        char* q;
        bool mapped = false;
        if (c >= kMapFrom) {
            c = (c + kHuge - 1) / kHuge * kHuge;
            void* m = mmap(nullptr, c + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) throw std::bad_alloc();
            // keep the 2 MiB-aligned part (the slack in front and behind goes back)
            char* a = (char*)(((uintptr_t)m + kHuge - 1) / kHuge * kHuge);
            if (a > (char*)m) munmap(m, (size_t)(a - (char*)m));
            if (a + c < (char*)m + c + kHuge) munmap(a + c, (size_t)((char*)m + c + kHuge - (a + c)));
            madvise(a, c, MADV_HUGEPAGE);
            q = a;
            mapped = true;
        } else {
            q = (char*)std::malloc(c);
            if (!q) throw std::bad_alloc();
        }
        if (n_) std::memcpy(q, p_, n_);
        release(p_, cap_, mapped_);
        p_ = q;
        cap_ = c;
        mapped_ = mapped;
    }
    void resize(size_t n) {  // new bytes are NOT initialised
        reserve(n);
        n_ = n;
    }
    std::string substr(size_t pos, size_t len) const { return std::string(p_ + pos, p_ + pos + len); }

  private:
    static constexpr size_t kHuge = (size_t)2 << 20, kMapFrom = (size_t)32 << 20;
    static void release(char* p, size_t cap, bool mapped) {
        if (!p) return;
        if (mapped) munmap(p, cap);
        else std::free(p);
    }
    char* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool mapped_ = false;
};
inline void swap(FastaBytes& a, FastaBytes& b) noexcept { a.swap(b); }  // (found by argument-dependent lookup; std::swap works too, through the moves)

struct FastaBatch {
    FastaBytes flat;             // concatenated sequences
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

    // ---- clean input, the common case: every sequence line holds [ACGTacgt] only (no '\r', no N).  Then every record is exactly
    // one sequence, the sequences can be counted before anything is copied, and several threads can write one batch in place.
    struct RangeCount {
        size_t n_seqs = 0, n_bases = 0;
        bool clean = true;
    };
    static bool has_avx2() {
#ifdef BRISK_FASTA_AVX2
        static const bool v = __builtin_cpu_supports("avx2");
        return v;
#else
        return false;
#endif
    }
#ifdef BRISK_FASTA_AVX2
    __attribute__((target("avx2"))) static bool all_bases_avx2(const char* p, const char* stop) {
        const __m256i up = _mm256_set1_epi8((char)0xDF), a = _mm256_set1_epi8('A'), c = _mm256_set1_epi8('C'), g = _mm256_set1_epi8('G'), t = _mm256_set1_epi8('T');
        for (; p + 32 <= stop; p += 32) {
            const __m256i u = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)p), up);
            const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, a), _mm256_cmpeq_epi8(u, c)), _mm256_or_si256(_mm256_cmpeq_epi8(u, g), _mm256_cmpeq_epi8(u, t)));
            if (_mm256_movemask_epi8(ok) != -1) return false;
        }
        return all_bases_scalar(p, stop);
    }
    __attribute__((target("avx2"))) static void copy_upper_avx2(char* d, const char* p, size_t n) {
        const __m256i up = _mm256_set1_epi8((char)0xDF);
        size_t i = 0;
        for (; i + 32 <= n; i += 32) _mm256_storeu_si256((__m256i*)(d + i), _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(p + i)), up));
        for (; i < n; i++) d[i] = (char)(p[i] & 0xDF);
    }
#endif
    static bool all_bases_scalar(const char* p, const char* stop) {
        static const BaseTable tab;
        for (; p < stop; p++)
            if (!tab.t[(unsigned char)*p]) return false;
        return true;
    }
    static bool all_bases(const char* p, const char* stop) {
#ifdef BRISK_FASTA_AVX2
        if (has_avx2()) return all_bases_avx2(p, stop);
#endif
        return all_bases_scalar(p, stop);
    }
    static void copy_upper(char* d, const char* p, size_t n) {  // valid bases only: & 0xDF is the upper case
#ifdef BRISK_FASTA_AVX2
        if (has_avx2()) return copy_upper_avx2(d, p, n);
#endif
        for (size_t i = 0; i < n; i++) d[i] = (char)(p[i] & 0xDF);
    }
    // sequences and bases of whole records [p, end) (the rules of parse_records); clean = false as soon as a sequence line holds anything else
    static RangeCount count_records(const char* p, const char* end) {
        RangeCount r;
        bool in_header = true;
        size_t cur = 0;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* stop = nl ? nl : end;
            if (!in_header) {
                if (*p == '>') {
                    r.n_seqs += cur > 0;
                    cur = 0;
                    in_header = true;
                } else {
                    if (!all_bases(p, stop)) {
                        r.clean = false;
                        return r;
                    }
                    cur += (size_t)(stop - p);
                    r.n_bases += (size_t)(stop - p);
                }
            }
            if (!nl) break;
            in_header = false;
            p = nl + 1;
        }
        r.n_seqs += cur > 0;
        return r;
    }
    // the same walk, writing: bases to flat + o (upper-cased), the end of every sequence to ends[0 .. n_seqs) as an offset into flat
    static void copy_records(const char* p, const char* end, char* flat, size_t o, uint64_t* ends) {
        bool in_header = true;
        size_t cur = 0;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* stop = nl ? nl : end;
            if (!in_header) {
                if (*p == '>') {
                    if (cur > 0) *ends++ = o;
                    cur = 0;
                    in_header = true;
                } else {
                    copy_upper(flat + o, p, (size_t)(stop - p));
                    o += (size_t)(stop - p);
                    cur += (size_t)(stop - p);
                }
            }
            if (!nl) break;
            in_header = false;
            p = nl + 1;
        }
        if (cur > 0) *ends++ = o;
    }

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
        if (all_bases(p, stop)) {  // the common line: nothing but bases, copied (upper-cased) 32 bytes at a time
            copy_upper(d + o, p, (size_t)(stop - p));
            return;
        }
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
                const auto q0 = std::chrono::steady_clock::now();
                // pass 1, side by side: count (clean ranges) or parse into a part of their own (ranges with anything but bases in a sequence line)
                std::vector<FastaReader::RangeCount> rc(n_threads_);
                std::vector<std::thread> th;
                for (unsigned t = 0; t < n_threads_; t++)
                    th.emplace_back([&, t] {
                        part[t].clear();
                        if (cut[t] >= cut[t + 1]) return;
                        rc[t] = FastaReader::count_records(cut[t], cut[t + 1]);
                        if (!rc[t].clean) {
                            FastaReader::parse_records(cut[t], cut[t + 1], part[t]);
                            rc[t].n_seqs = part[t].size();
                            rc[t].n_bases = part[t].flat.size();
                        }
                    });
                for (auto& x : th) x.join();
                std::vector<size_t> fo(n_threads_ + 1, 0), so(n_threads_ + 1, 0);
                for (unsigned t = 0; t < n_threads_; t++) {
                    fo[t + 1] = fo[t] + rc[t].n_bases;
                    so[t + 1] = so[t] + rc[t].n_seqs;
                }
                const auto q1 = std::chrono::steady_clock::now();
                // (the buffer circulates with the consumer's: it is not cleared, so growing it zero-fills only what is new)
                b.flat.resize(fo[n_threads_]);
                b.offs.resize(so[n_threads_] + 1);
                b.offs[0] = 0;
                const auto q2 = std::chrono::steady_clock::now();
                // pass 2, side by side again: clean ranges are copied from the file to their place, the others from their part
                th.clear();
                for (unsigned t = 0; t < n_threads_; t++)
                    th.emplace_back([&, t] {
                        if (cut[t] >= cut[t + 1]) return;
                        if (rc[t].clean) {
                            FastaReader::copy_records(cut[t], cut[t + 1], &b.flat[0], fo[t], &b.offs[so[t] + 1]);
                            return;
                        }
                        if (!part[t].flat.empty()) std::memcpy(&b.flat[fo[t]], part[t].flat.data(), part[t].flat.size());
                        for (size_t i = 0; i < part[t].size(); i++) b.offs[so[t] + i + 1] = fo[t] + part[t].offs[i + 1];
                    });
                for (auto& x : th) x.join();
                if (getenv("BRISK_FASTA_DEBUG")) {
                    const auto q3 = std::chrono::steady_clock::now();
                    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                    fprintf(stderr, "[brisk_fasta] batch of %zu bases: cut %.1f ms, count %.1f, resize %.1f, copy %.1f\n", (size_t)fo[n_threads_], ms(p0, q0), ms(q0, q1), ms(q1, q2), ms(q2, q3));
                }
                pos = stop;
                produce_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
                if (!publish(b, pos >= end)) return;
                b.offs.assign(1, 0);  // (what came back from the consumer keeps its bytes: see the resize above)
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
