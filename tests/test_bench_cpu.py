"""bench.py's CPU baseline leg (no GPU): the checker timed beside the GPU number -- thread count from the host's CPU share,
bounded in reads and in time, every sample read accounted for."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_host_cpu_share(monkeypatch):
    monkeypatch.delenv("BRISK_CPU_THREADS", raising=False)
    n = bench.host_cpu_share()
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    assert 1 <= n <= aff
    monkeypatch.setenv("BRISK_CPU_THREADS", "3")
    assert bench.host_cpu_share() == 3


def test_cpu_baseline_is_bounded_and_counts_what_it_ran(monkeypatch):
    monkeypatch.setenv("BRISK_CPU_THREADS", "2")
    out = bench.cpu_baseline(31, 15, 14, 150, 15.0, 30_000)
    assert out["kind"] in ("reference", "port") and out["unit"] == "k-mers/s" and out["value"] > 0
    assert out["cores"] == (2 if out["kind"] == "reference" else 1)
    assert out["sample"].startswith("30000 of 30000 synthetic 150 bp reads")
    # the same sample through the oracle's own index: the entries the leg reports are the sample's entries
    import numpy as np
    import oracle
    O = oracle.Oracle()
    G = max(int(30_000 * 150 / 15.0), 151)
    reads = O.synth_reads(G, 0, 30_000, 150)
    h = O.index_new(31, 15, 14)
    O.index_insert_reads(h, np.ascontiguousarray(reads.reshape(-1)), np.arange(30_001, dtype=np.uint64) * np.uint64(150))
    nk, _ = O.index_stats(h)
    O.index_free(h)
    assert f": {nk} entries in " in out["sample"]
