// writer.hpp -- facade header with the reference's name and surface (brisk/writer.hpp:11-19): BriskWriter dumps an
// index as a KFF file.  The reference walks DenseMenuYo / Bucket / SKL internals and calls the un-vendored kff-cpp-api;
// here the entries come from the device (brisk_hip_enumerate_ids through Brisk::visit_entries) and the bytes from this
// repo's own emitter (brisk_kff.hpp, which also says why parity with the reference's output is unpinned).
#ifndef BRISK_AMD_WRITER_HPP
#define BRISK_AMD_WRITER_HPP
#include <memory>
#include <string>

#include "Brisk.hpp"
#include "brisk_kff.hpp"

class BriskWriter {
  public:
    explicit BriskWriter(std::string filename) : filename_(std::move(filename)) {}
    // brisk/writer.hpp:75-179: global variables (k, data_size, max = 1), then (k, m, data_size, max = 2(k - m)), then the
    // minimizer sections, buckets ascending
    template <class DATA>
    void write(Brisk<DATA>& index) {
        KffIndexWriter w(filename_, index.params.k, index.params.m, (uint32_t)sizeof(DATA));
        index.visit_entries([&](const kmer_full& km, const DATA* data) {
            w.add(KffEntry{(uint64_t)km.kmer_s, (uint64_t)(km.kmer_s >> 64), km.minimizer_idx, reinterpret_cast<const uint8_t*>(data)});
        });
        w.close();
    }
    void close() {}  // brisk/writer.hpp:183-186: the file is complete when write() returns

  private:
    std::string filename_;
};

#endif
