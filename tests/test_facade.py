"""The drop-in boundary one level up: the C++ facade with the reference's names
(brisk_amd/include) over the C-ABI.  CPU side: it compiles -- including the reference's own
apps/counter.cpp, UNCHANGED, where the reference tree exists.  GPU side: the built binaries
count the reference's fixture and agree with the goldens."""
import hashlib
import os
import re
import subprocess
import sys

import pytest

import brisk_amd
from conftest import GOLDEN, load_golden

APPS = os.path.join(os.path.dirname(brisk_amd.__file__), "apps")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(brisk_amd.__file__)))


def test_facade_compiles_and_counter_cpp_links_unchanged():
    brisk_amd.build_library()
    apps = brisk_amd.build_apps()
    assert os.path.exists(apps["brisk_count"])
    if os.path.isdir("/root/reference"):
        assert os.path.exists(apps["counter_ref"]), "apps/counter.cpp did not compile against the facade headers"


def test_facade_headers_carry_the_reference_surface():
    inc = os.path.join(os.path.dirname(brisk_amd.__file__), "include")
    brisk = open(os.path.join(inc, "Brisk.hpp")).read()
    for name in ("insert_superkmer", "get_superkmer", "protect_data", "unprotect_data", "restart_kmer_enumeration",
                 "insert_sequence", "get_sequence", "reallocate", "stats", "DATA* get(kmer_full& kmer)", "bool next(kmer_full& kmer)",
                 "DenseMenuYo<DATA>* menu", "Parameters params"):
        assert name in brisk, name
    for f in ("Kmers.hpp", "parameters.hpp", "Decycling.h", "hashing.hpp", "buckets.hpp", "DenseMenuYo.hpp", "writer.hpp"):
        assert os.path.exists(os.path.join(inc, f))


def _md5_of_dump(path):
    lines = [l for l in open(path).read().split("\n") if l]
    return hashlib.md5("".join("{} idx={} {}\n".format(*l.split()) for l in lines).encode()).hexdigest(), lines


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--facade", "--bulk"])
def test_brisk_count_binary_on_reference_fixture(tmp_path, mode):
    exe = os.path.join(APPS, "brisk_count")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    for e in load_golden("multisets.json"):
        if e["input"] != "test.fa" or (e["k"], e["m"], e["b"]) not in ((31, 11, 4), (63, 21, 14)):
            continue
        dump = str(tmp_path / "dump.txt")
        fa = os.path.join(GOLDEN, "test.fa")
        if mode == "--bulk":  # the streaming front-end: gz input, batches far smaller than the file
            import gzip
            fa = str(tmp_path / "test.fa.gz")
            with gzip.open(fa, "wt") as f:
                f.write(open(os.path.join(GOLDEN, "test.fa")).read())
        out = subprocess.run([exe, mode, fa, str(e["k"]), str(e["m"]), str(e["b"]), dump],
                             capture_output=True, text=True, timeout=300, env=dict(os.environ, BRISK_BATCH_BASES="1500"))
        assert out.returncode == 0, out.stderr
        assert out.stdout.split() == ["nb_kmers", str(e["nb_kmers"]), "nb_buckets", str(e["nb_buckets"]), "sum_counts", str(e["sum_counts"])]
        assert _md5_of_dump(dump)[0] == e["md5"]


@pytest.mark.gpu
def test_reference_counter_cpp_runs_on_the_gpu_index():
    """BASELINE config #1: apps/counter on data/test.fa, k=31 m=11 b=4, with its own --mode 2 self-check.
    The binary is the reference's counter.cpp compiled unchanged against brisk_amd/include (built where
    the reference tree exists; it travels to the GPU box as a built artefact)."""
    exe = os.path.join(APPS, "counter_ref")
    if not os.path.exists(exe):
        pytest.skip("counter_ref not built (needs the reference tree at build time)")
    out = subprocess.run([exe, "-f", os.path.join(GOLDEN, "test.fa"), "-k", "31", "-m", "11", "-b", "4", "-t", "1", "--mode", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "All counts are correct !" in out.stdout, out.stdout[-2000:]
    assert re.search(r"nb kmers: 6,163", out.stdout) and re.search(r"^221 bucket used", out.stdout, re.M), out.stdout[-1500:]
    # threads: the facade serialises calls on the handle
    out = subprocess.run([exe, "-f", os.path.join(GOLDEN, "test.fa"), "-k", "63", "-m", "21", "-b", "14", "-t", "4", "--mode", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "All counts are correct !" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    assert re.search(r"nb kmers: 6,105", out.stdout) and re.search(r"^237 bucket used", out.stdout, re.M)
    # the per-call route's throughput, as the app itself prints it (--mode 1: count only; one launch per super-k-mer, entry ids and
    # DATA on the host: the plumbing-compatible route, the bulk calls are the fast one)
    import time
    t0 = time.perf_counter()
    out = subprocess.run([exe, "-f", os.path.join(GOLDEN, "debug_test.fa"), "-k", "31", "-m", "11", "-b", "4", "-t", "1", "--mode", "1"],
                         capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    assert re.search(r"nb kmers: 27,283", out.stdout), out.stdout[-1500:]
    m_ = re.search(r"Kmer counted elapsed time: ([0-9.]+)s", out.stdout)
    line = "counter_ref --mode 1 on debug_test.fa (k31 m11 b4, 27,283 entries, 2,285 super-k-mers): %s s counting (%s entries/s), %.2f s wall" % (
        m_.group(1) if m_ else "?", ("%.0f" % (27283 / float(m_.group(1)))) if m_ and float(m_.group(1)) > 0 else "?", wall)
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "facade_throughput.txt"), "w").write(line + "\n")


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py prints ONE JSON line with the keys the driver reads, a roofline and a cpu_baseline object, and its
    own size-independent verification of what it built."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "300000", "--steps", "1", "--warmup", "1", "--cpu-sample-reads", "20000"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["verify"]["every_kmer_counted_once"] is True


@pytest.mark.gpu
def test_two_rank_rehearsal_matches_single_rank():
    """The N>1 flow of bench.py (pieces, route, all-to-alls of records and histogram slices, one insert per owner) on
    two gloo ranks sharing the GPU builds the same index -- entry count and digest -- as one rank over the same reads."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "600000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--reads", "300000",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    v1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])["verify"]
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    v2 = d2["verify"]
    assert d2["n_gpus"] == 2 and v2["every_kmer_counted_once"]
    for key in ("entries", "nb_kmers", "nb_buckets", "sum_counts", "digest_mod_2_64"):
        assert v1[key] == v2[key], key


@pytest.mark.gpu
def test_strong_scaling_rehearsal_three_ranks_odd_reads():
    """bench.py --scaling strong: the job is fixed (--reads in all), rank r takes its contiguous share; three gloo ranks on
    one GPU with a read count that does not divide build the same index as one rank, and the line says "strong"."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    reads = "500001"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", reads, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--scaling", "strong"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port",
                            str(29950 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--backend", "gloo", "--share-gpu", "--reads", reads,
                            "--scaling", "strong", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert three.returncode == 0, three.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    d3 = json.loads([l for l in three.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["scaling"] == d3["scaling"] == "strong" and d3["n_gpus"] == 3 and d3["config"]["total_reads"] == int(reads)
    assert d3["verify"]["every_kmer_counted_once"]
    for key in ("entries", "nb_kmers", "nb_buckets", "sum_counts", "digest_mod_2_64"):
        assert d1["verify"][key] == d3["verify"][key], key


@pytest.mark.gpu
def test_sharded_count_with_unequal_and_empty_shares():
    """ranks holding different numbers of reads -- on different sides of the piece thresholds, one of them
    none at all -- must issue the same collectives.  Two gloo ranks on one GPU count two batches (the second one with
    an empty rank and a different piece count than the first); the shards' digests add up to the single-index digest."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(30300 + os.getpid() % 300), os.path.join(ROOT, "tests", "uneven_ranks_worker.py")], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["pieces"] == [1, 4], d  # (a small batch goes in one piece: two collectives)
    assert d["sharded"] == d["single"], d
