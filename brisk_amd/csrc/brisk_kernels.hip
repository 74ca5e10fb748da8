// brisk_kernels.hip -- HIP kernels of the Brisk hot path for gfx950 (MI355X).
//
// One translation unit, four parts:
//   brisk_scan.hip       reads -> super-k-mer records (SuperKmerEnumerator::next, Kmers.cpp:522-603,
//                        + hash_kmer_minimizer_inplace / get_compacted, Kmers.cpp:138-145,191-200); long sequences in chunks
//   brisk_partition.hip  records -> partition order (histogram prefix, k_scatter) and -> owner order (multi-GPU routing)
//   brisk_insert.hip     per partition: find-all, insert-missing, count++  (DenseMenuYo.hpp:248-310, counter.cpp:262-269)
//   brisk_readout.hip    Brisk::get / get_superkmer / next / stats and the per-call upsert of the facade
//
// All of it is integer / byte work (plus the FP64 decycling class); there is no MFMA-shaped work on this path.
#include "brisk_device.h"

#define SCAN_BLOCK 256
#define EMPTY_SLOT 0xffffffffu
#define MATCHED_BIT 0x80000000u

#include "brisk_scan.hip"
#include "brisk_partition.hip"
#include "brisk_insert.hip"
#include "brisk_readout.hip"
