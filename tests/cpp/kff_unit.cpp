// Drives brisk_kff.hpp without a GPU (tests/test_kff.py): writes entries given on stdin as "KMER idx count" lines.
//   kff_unit OUT.kff k m < lines
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <string>

#include "brisk_kff.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const uint32_t k = (uint32_t)atoi(argv[2]), m = (uint32_t)atoi(argv[3]);
    KffIndexWriter w(argv[1], k, m, 1);
    std::string km;
    unsigned idx, cnt;
    while (std::cin >> km >> idx >> cnt) {
        __uint128_t v = 0;
        for (char c : km) v = (v << 2) | (((unsigned)c >> 1) & 3u);  // A0 C1 T2 G3 (brisk/Kmers.cpp:442-444)
        const uint8_t data = (uint8_t)cnt;
        w.add(KffEntry{(uint64_t)v, (uint64_t)(v >> 64), (uint8_t)idx, &data});
    }
    w.close();
    return 0;
}
