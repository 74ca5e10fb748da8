"""The reference's CPU path (oracle/_ref) by thread count on this host: 1 M synthetic 150 bp reads, k63 m21 b14.  On a GPU box the
affinity mask lists every core of the host while the cgroup grants a share of them; past that share the OpenMP path with its lock
stripes collapses (profiles/r02_cpu_threads.txt).  bench.py's cpu_baseline takes its thread count from the cgroup quota for that reason.
    python3 tools/cpu_threads_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
