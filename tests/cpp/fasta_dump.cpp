// prints every sequence the host front-end yields for a FASTA / FASTA.gz file, one per line
#include <iostream>

#include "brisk_fasta.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FastaBatcher batches(argv[1], (size_t)atoll(argv[2]), argc > 3 ? (unsigned)atoi(argv[3]) : 0u);  // threads for a plain file (0: default)
    FastaBatch b;
    const bool marks = argc > 4;  // a 4th argument: a "#batch N" line in front of every batch of N sequences
    while (batches.next(b)) {
        if (marks) std::cout << "#batch " << b.size() << "\n";
        for (size_t i = 0; i < b.size(); i++) std::cout << b.flat.substr(b.offs[i], b.offs[i + 1] - b.offs[i]) << "\n";
    }
    return 0;
}
