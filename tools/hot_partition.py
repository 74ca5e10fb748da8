"""One hot partition: every read carries the same minimizer X between random flanks, so that X's partition receives 17 distinct
k-mers per read (k=31 m=15 b=14).  python tools/hot_partition.py N_READS   (BRISK_HUGE_AT=0: without k_insert_huge)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import brisk_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k, m, b = 31, 15, 14
rng = random.Random(5)
ix = brisk_amd.BriskHip(k, m, b)
cands = np.array([rng.getrandbits(2 * m) for _ in range(1 << 20)], dtype=np.uint64)
best = int(cands[np.argmin(ix.debug_order_keys(cands))])
X = "".join("ACTG"[(best >> (2 * (m - 1 - i))) & 3] for i in range(m))
reads = ["".join(rng.choice("ACGT") for _ in range(40)) + X + "".join(rng.choice("ACGT") for _ in range(40)) for _ in range(n)]
ix.insert_reads(reads[:10]); ix.stats(); ix.clear()   # warm-up
for rep in range(2):
    ix.clear()
    t0 = time.perf_counter(); ix.insert_reads(reads); st = ix.stats(); dt = time.perf_counter() - t0
    print("HUGE_AT=%s %d reads: %.1f ms  entries %d, largest partition %d" % (os.environ.get("BRISK_HUGE_AT", "default"), n, dt * 1e3, st["nb_kmers"], st["largest_bucket"]), flush=True)
t0 = time.perf_counter(); sums = ix.get_reads(reads); dt = time.perf_counter() - t0
print("get of the same reads: %.1f ms, sum %d" % (dt * 1e3, int(sums.sum())), flush=True)
print("checksum", ix.checksum())
