// brisk_kff.hpp -- a from-scratch emitter of the K-mer File Format (KFF 1.0) for what BriskWriter writes
// (reference: brisk/writer.hpp:22-30 header + encoding, :75-179 sections).  The reference drives the un-vendored
// kff-cpp-api (brisk/lib/kff is an empty submodule: SURVEY.md 8(c)); nothing of it is available here, so the bytes
// below follow the published KFF 1.0 layout as restated in this header, and PARITY WITH THE REFERENCE'S OUTPUT IS
// UNPINNED (no kff library, no KFF fixture, no KMC in the reference tree).  tests/kff_reader.py reads the same layout
// back; tests/test_kff.py round-trips an index to its (k-mer, minimizer_idx, count) multiset.
//
// Layout written (all integers big-endian):
//   header   "KFF" | major 1 | minor 0 | encoding byte (A<<6 | C<<4 | G<<2 | T) | uniqueness 0 | canonicity 0 |
//            u32 metadata size | metadata
//   'v'      u64 n | n x (name, NUL, u64 value)                     global variables: k, m, data_size, max
//   'm'      minimizer, ceil(m/4) bytes | u64 n_blocks | n_blocks x block
//            block = n_kmers (ceil(ceil(log2 max)/8) bytes, absent when max == 1)
//                  | minimizer position (ceil(ceil(log2(k+max-1))/8) bytes): nts of the block's sequence in front of it
//                  | sequence WITHOUT the minimizer, (k + n_kmers - 1 - m) nts, 2 bits each, first nt in the highest
//                    bits, the unused highest bits of the first byte zero (writer.hpp:42-71)
//                  | data, n_kmers x data_size bytes
//   footer   'v' section {first_index = 0, footer_size} | "KFF"
// Nucleotide codes are the index's own (A0 C1 T2 G3 = the reference's encoding 0,1,3,2 for A,C,G,T, writer.hpp:26).
#ifndef BRISK_AMD_KFF_HPP
#define BRISK_AMD_KFF_HPP
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

class KffOut {
  public:
    explicit KffOut(const std::string& path) : f_(fopen(path.c_str(), "wb")), k_(0), m_(0), data_size_(0), max_(1), in_section_(false), n_blocks_(0) {
        if (!f_) throw std::runtime_error("KFF: cannot open " + path);
        const uint8_t head[8] = {'K', 'F', 'F', 1, 0, (uint8_t)((0u << 6) | (1u << 4) | (3u << 2) | 2u), 0, 0};  // encoding A0 C1 G3 T2
        put(head, 8);
        const std::string meta = "File generated with Brisk v1. See https://github.com/Malfoy/Brisk";  // writer.hpp:28
        be(meta.size(), 4);
        put(meta.data(), meta.size());
    }
    ~KffOut() {
        if (f_) fclose(f_);
    }
    KffOut(const KffOut&) = delete;
    KffOut& operator=(const KffOut&) = delete;

    // Section_GV (writer.hpp:77-91).  The variables in force for the sections that follow are remembered.
    void global_variables(const std::vector<std::pair<std::string, uint64_t>>& vars) {
        end_section();
        put("v", 1);
        be(vars.size(), 8);
        for (const auto& v : vars) {
            put(v.first.c_str(), v.first.size() + 1);
            be(v.second, 8);
            if (v.first == "k") k_ = v.second;
            else if (v.first == "m") m_ = v.second;
            else if (v.first == "data_size") data_size_ = v.second;
            else if (v.first == "max") max_ = v.second;
        }
    }
    // Section_Minimizer (writer.hpp:138-153): `minimizer` = the m-mer as a 2-bit value, first nt in the highest bits
    void begin_minimizer_section(uint64_t minimizer) {
        end_section();
        if (!k_ || !m_ || m_ >= k_) throw std::logic_error("KFF: minimizer section without k and m");
        in_section_ = true;
        n_blocks_ = 0;
        body_.clear();
        uint8_t buf[16];
        pack_nts(buf, minimizer, 0, m_);
        head_.assign(buf, buf + (m_ + 3) / 4);
    }
    // write_compacted_sequence_without_mini (writer.hpp:166-173): a super-k-mer of n_kmers k-mers whose minimizer sits
    // `mini_pos` nts from its left end; seq_hi:seq_lo = its k + n_kmers - 1 - m other nts (128-bit, first nt highest)
    void block(uint64_t n_kmers, uint64_t mini_pos, uint64_t seq_lo, uint64_t seq_hi, const uint8_t* data) {
        if (!in_section_ || n_kmers == 0 || n_kmers > max_) throw std::logic_error("KFF: bad block");
        const uint64_t nts = k_ + n_kmers - 1 - m_;
        if (nts > 64) throw std::logic_error("KFF: block sequence longer than 64 nts");
        if (max_ > 1) be_vec(n_kmers, bytes_for(max_));
        be_vec(mini_pos, bytes_for(k_ + max_ - 1));
        uint8_t buf[16];
        pack_nts(buf, seq_lo, seq_hi, nts);
        body_.insert(body_.end(), buf, buf + (nts + 3) / 4);
        body_.insert(body_.end(), data, data + n_kmers * data_size_);
        n_blocks_++;
    }
    void close() {
        end_section();
        // footer: a global-variable section telling where the (absent) index starts and how long the footer is, then the signature
        const long at = ftell(f_);
        global_variables({{"first_index", 0}, {"footer_size", 0}});
        const long len = ftell(f_) - at + 3;
        fseek(f_, at, SEEK_SET);
        global_variables({{"first_index", 0}, {"footer_size", (uint64_t)len}});
        put("KFF", 3);
        if (fclose(f_) != 0) {
            f_ = nullptr;
            throw std::runtime_error("KFF: write failed");
        }
        f_ = nullptr;
    }

    static uint64_t bytes_for(uint64_t max_value) {  // bytes of a field that holds values up to max_value: ceil(ceil(log2 max) / 8)
        uint64_t bits = 0;
        while ((1ull << bits) < max_value && bits < 63) bits++;
        return (bits + 7) / 8;
    }

  private:
    void end_section() {
        if (!in_section_) return;
        in_section_ = false;
        put("m", 1);
        put(head_.data(), head_.size());
        be(n_blocks_, 8);
        put(body_.data(), body_.size());
    }
    // n nts of hi:lo (first nt in the highest used bits) -> ceil(n/4) bytes, right-aligned, unused top bits zero
    static void pack_nts(uint8_t* out, uint64_t lo, uint64_t hi, uint64_t n) {
        const uint64_t nb = (n + 3) / 4;
        for (uint64_t i = 0; i < nb; i++) {  // byte i from the END holds nts [4i, 4i+4) counted from the last nt
            const unsigned s = (unsigned)(8 * i);
            const uint64_t v = s >= 64 ? (hi >> (s - 64)) : s == 0 ? lo : ((lo >> s) | (hi << (64 - s)));
            uint8_t byte = (uint8_t)v;
            const uint64_t have = n - 4 * i;  // nts left for this byte
            if (have < 4) byte &= (uint8_t)((1u << (2 * have)) - 1);
            out[nb - 1 - i] = byte;
        }
    }
    void put(const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f_) != n) throw std::runtime_error("KFF: write failed");
    }
    void be(uint64_t v, int bytes) {
        uint8_t b[8];
        for (int i = 0; i < bytes; i++) b[i] = (uint8_t)(v >> (8 * (bytes - 1 - i)));
        put(b, bytes);
    }
    void be_vec(uint64_t v, uint64_t bytes) {
        for (uint64_t i = 0; i < bytes; i++) body_.push_back((uint8_t)(v >> (8 * (bytes - 1 - i))));
    }
    FILE* f_;
    uint64_t k_, m_, data_size_, max_;
    bool in_section_;
    uint64_t n_blocks_;
    std::vector<uint8_t> head_, body_;
};

// One entry of the index as the writer sees it: the k-mer (hi:lo, 2k bits, minimizer UNHASHED: what Brisk::next returns),
// its minimizer_idx (nts of the k-mer behind the minimizer) and a pointer to its DATA.
struct KffEntry {
    uint64_t lo, hi;
    uint8_t minimizer_idx;
    const uint8_t* data;
};

// The sections of BriskWriter::write (writer.hpp:75-179) from a stream of entries grouped by bucket (ascending, as
// brisk_hip_enumerate yields them): two global-variable sections, then one minimizer section per run of entries that
// share a minimizer.  The device index stores k-mers, not super-k-mers, so every block holds ONE
// k-mer (n_kmers = 1 <= max); readers see the same (k-mer, data) multiset as from the reference's multi-k-mer blocks.
class KffIndexWriter {
  public:
    KffIndexWriter(const std::string& path, uint32_t k, uint32_t m, uint32_t data_size) : out_(path), k_(k), m_(m), have_(false), cur_(0) {
        out_.global_variables({{"k", k}, {"data_size", data_size}, {"max", 1}});                          // writer.hpp:77-81
        out_.global_variables({{"k", k}, {"m", m}, {"data_size", data_size}, {"max", 2ull * (k - m)}});  // writer.hpp:85-90
    }
    void add(const KffEntry& e) {
        const unsigned idx = e.minimizer_idx;
        const __uint128_t km = ((__uint128_t)e.hi << 64) | e.lo;
        const uint64_t mm = (uint64_t)(km >> (2 * idx)) & ((1ull << (2 * m_)) - 1);
        if (!have_ || mm != cur_) {
            out_.begin_minimizer_section(mm);
            have_ = true;
            cur_ = mm;
        }
        // the k-mer without its minimizer: prefix (k - m - idx nts) then suffix (idx nts)
        const __uint128_t suffix = idx ? (km & ((((__uint128_t)1) << (2 * idx)) - 1)) : 0;
        const __uint128_t prefix = km >> (2 * (idx + m_));
        const __uint128_t seq = (prefix << (2 * idx)) | suffix;
        out_.block(1, k_ - m_ - idx, (uint64_t)seq, (uint64_t)(seq >> 64), e.data);
    }
    void close() { out_.close(); }

  private:
    KffOut out_;
    uint32_t k_, m_;
    bool have_;
    uint64_t cur_;
};

#ifdef BRISK_HIP_H
// A bulk-count index (DATA = the uint8_t counter on the device) straight from the C-ABI: brisk_hip_enumerate in chunks.
inline int brisk_write_kff(brisk_hip_index* h, const std::string& path) {
    brisk_hip_layout lay{};
    int rc = brisk_hip_get_layout(h, &lay);
    if (rc != BRISK_HIP_OK) return rc;
    KffIndexWriter w(path, lay.k, lay.m, 1);
    uint64_t cursor = 0, n = 0, cap = 1u << 20;
    std::vector<uint64_t> lo(cap), hi(cap);
    std::vector<uint8_t> idx(cap), cnt(cap);
    for (;;) {
        rc = brisk_hip_enumerate(h, &cursor, lo.data(), hi.data(), idx.data(), cnt.data(), cap, &n);
        if (rc == BRISK_HIP_ECAPACITY) {  // one bucket range larger than the buffer
            cap *= 8;
            lo.resize(cap); hi.resize(cap); idx.resize(cap); cnt.resize(cap);
            continue;
        }
        if (rc != BRISK_HIP_OK) return rc;
        if (n == 0) break;
        for (uint64_t i = 0; i < n; i++) w.add(KffEntry{lo[i], hi[i], idx[i], &cnt[i]});
    }
    w.close();
    return BRISK_HIP_OK;
}
#endif

#endif
