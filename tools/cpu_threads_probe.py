import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, oracle
O = oracle.Oracle(); R = oracle.Ref()
k, m, b, L = 63, 21, 14, 150
n = 1_000_000
G = int(n * L / 15)
reads = O.synth_reads(G, 0, n, L)
flat = np.ascontiguousarray(reads.reshape(-1)); offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), flush=True)
for t in (16, 32, 64, 128, 256):
    h = R.index_new(k, m, b)
    t0 = time.perf_counter(); R.index_insert_reads(h, flat, offs, threads=t); dt = time.perf_counter() - t0
    nk, _ = R.index_stats(h); R.index_free(h)
    print(t, "threads:", round(nk / dt / 1e6, 2), "M entries/s", round(dt, 2), "s", flush=True)
