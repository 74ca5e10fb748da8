// writer.hpp -- facade header with the reference's name (brisk/writer.hpp:11-19).  KFF output
// needs the un-vendored kff-cpp-api and is out of scope for this round:
// write() reports that and produces no file.
#ifndef BRISK_AMD_WRITER_HPP
#define BRISK_AMD_WRITER_HPP
#include <iostream>
#include <string>

template <class DATA>
class Brisk;

class BriskWriter {
  public:
    explicit BriskWriter(std::string filename) : filename_(std::move(filename)) {}
    template <class DATA>
    void write(Brisk<DATA>&) {
        std::cerr << "BriskWriter: KFF output is not implemented in the MI355X build; " << filename_ << " not written" << std::endl;
    }
    void close() {}

  private:
    std::string filename_;
};

#endif
