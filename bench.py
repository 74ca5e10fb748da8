#!/usr/bin/env python3
"""bench.py -- distinct k-mers indexed/sec (k=63, m=21) on N MI355X.

One "step" = one whole counting job: a fresh index, then the hot path (scan ->
bucket-radix scatter -> per-partition insert) over this rank's synthetic reads,
which are generated ON DEVICE, packed 2-bit, before the timed region starts.
    N = 1 : BASELINE.json configs[2] -- 50M x 150 bp reads, k=63 m=21 b=14.
    N > 1 : weak scaling toward configs[3] (8 x 50M reads): every rank scans its own
            50M-read shard of an N x 500 Mbp genome, routes super-k-mer records
            to the owner of their bucket range with ONE all-to-all (RCCL over
            xGMI) and inserts what it owns.  value = entries created by all
            ranks / max-over-ranks time.
    --scaling strong : the job is fixed instead -- --reads reads in all, rank r scans
            reads [r, r+1) * reads / N of the SAME genome -- so that value(N) / value(1)
            is the strong-scaling speed-up BASELINE.json's target is quoted in.
Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under
python -m torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from env).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per read = packed read in + each super-k-mer
# record written once and read once + one count read-modify-write per k-mer instance
B_ALG = {(63, 21, 14): 456.0, (31, 11, 11): 633.0}
# super-k-mers per 150 bp read of the bench's synthetic reads (measured: records of the scan / reads), for the other parameter sets
N_SKM = {(63, 21): 4.17, (31, 11): 11.1, (31, 15): 13.0}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def b_alg(k, m, b, L, n_skm_per_read):
    if (k, m, b) in B_ALG and L == 150:
        return B_ALG[(k, m, b)]
    alloc = (2 * k - m - b + 3) // 4
    return (L + 3) // 4 + 2 * n_skm_per_read * (6 + alloc) + 2 * (L - k + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU per step (weak scaling) or in the whole job (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak: --reads per GPU; strong: --reads in all, split over the GPUs")
    ap.add_argument("--k", type=int, default=63)
    ap.add_argument("--m", type=int, default=21)
    ap.add_argument("--b", type=int, default=14)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--coverage", type=float, default=15.0)
    ap.add_argument("--part-bits", type=int, default=0, help="log2(#partitions); 0: library default")
    ap.add_argument("--cpu-sample-reads", type=int, default=8_000_000, help="reads of the CPU baseline sample (10-15 s of the reference on the box's CPU share; the leg stops after ~30 s whatever the host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-reads", type=int, default=4_000_000, help="reads of the end-to-end leg (N=1): the same synthetic reads written as .fa.gz and as plain FASTA, counted by "
                                                                   "brisk_count from the file; 0: skip.  Reported under \"end_to_end\", never part of `value`")
    ap.add_argument("--get", action="store_true", help="also time the get path afterwards (N=1): per-read sums of counts over the same reads; reported under \"get\"")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import brisk_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must run under torch.distributed.run (one rank per GPU)")
        args.gpus = world
    N = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --backend gloo + --share-gpu: rehearsal of the N>1 flow with every rank on cuda:0
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if N > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    k, m, b, L = args.k, args.m, args.b, args.read_len
    if args.scaling == "strong":  # one fixed job: this rank's contiguous share of its reads
        total_reads = args.reads
        first_read = total_reads * rank // N
        n_reads = total_reads * (rank + 1) // N - first_read
    else:
        n_reads = args.reads
        total_reads = n_reads * N
        first_read = rank * n_reads
    genome_len = max(int(total_reads * L / args.coverage), L + 1)
    stream = torch.cuda.Stream(device=dev)
    sptr = stream.cuda_stream

    def barrier():
        torch.cuda.synchronize(dev)
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- inputs resident in HBM before the timed region --------------------------
    with torch.cuda.stream(stream):
        d_packed = torch.zeros((n_reads * L + 15) // 16 + 4, dtype=torch.int32, device=dev)
        d_starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    gen = brisk_amd.BriskHip(k, m, b, device=dev_index, stream=sptr, part_bits=2)
    stream.synchronize()
    gen.synth_reads(genome_len, first_read, n_reads, L, d_packed.data_ptr(), d_starts.data_ptr())
    gen.sync()
    gen.close()

    # The end-to-end leg (N = 1; reported, never `value`) runs FIRST, while this process holds nothing on the device but the reads:
    # its child processes map their arenas from memory nobody has used -- run after the timed steps, next to (or right behind)
    # this process' own 146 GB index, their first arena mappings alone took 0.4-1.4 s.
    e2e_leg = None
    if args.e2e_reads and N == 1:
        try:
            e2e_leg = end_to_end(k, m, b, L, min(args.e2e_reads, n_reads), d_packed)
        except Exception as e:  # noqa: BLE001  (the leg must never take the bench line with it)
            e2e_leg = {"error": repr(e)[:300]}

    # one handle for the whole run: every step starts from brisk_hip_clear(), i.e. an
    # empty index whose device memory is already reserved (the allocator, not the path)
    from brisk_amd.exchange import ShardedCounter, suggest_part_bits
    # partitions follow the batch (one batch per job here): at most 512 k-mer instances per partition (1024 with the two-word
    # records of k <= 32, where fewer than 2^24 pay: exchange.suggest_part_bits); 2^24 for the headline config
    part_bits = args.part_bits or suggest_part_bits(b, total_reads, L - k + 1, min_bits=22 if k <= 32 else 24, per_partition=1024 if k <= 32 else 512)
    sc = ShardedCounter(k, m, b, rank, N, dev_index, stream, part_bits=part_bits)
    ix = sc.ix
    # ownership (N > 1): equal partition ranges unless a scan of this rank's first reads shows the most loaded owner more than 1.3x
    # above the mean; then every rank installs the same histogram-balanced cut points.  Once per job, before the
    # warm-up: not part of a step.
    cuts = sc.balance(d_packed, d_starts, min(n_reads, 2_000_000)) if N > 1 else None

    def one_job(profile):
        """empty index + the whole hot path over this rank's reads"""
        ix.clear()
        ix.profile_enable(profile)
        sc.count_packed(d_packed, d_starts, n_reads)
        ix.sync()

    def run_steps(nsteps, profile):
        entries = 0
        prof = {}
        ix.profile_reset()
        for _ in range(nsteps):
            one_job(profile)
            entries += ix.stats()["nb_kmers"]
        if profile:
            prof = {n: v for n, v in ix.profile_read().items()}
        return entries, prof

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    entries, prof = run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    e = torch.tensor([entries], dtype=torch.float64, device=dev)
    if N > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    entries_all = float(e.item())

    # after the timed region: a size-independent parity property of what was just built -- every k-mer
    # instance counted exactly once (sum of counts == reads x (L-k+1); no count wraps at this coverage) and
    # an order-independent digest of the index (shard digests add up; equal digests <=> equal multisets)
    ent, sumc, dig = ix.checksum()
    limbs = [(dig >> (22 * i)) & ((1 << 22) - 1) for i in range(3)]  # int64 all-reduce cannot overflow on 22-bit limbs
    chk = torch.tensor([ent, sumc] + limbs, dtype=torch.int64, device=dev)
    if N > 1:
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    chk = [int(v) for v in chk.cpu().tolist()]
    whole = sc.stats()  # Brisk::stats of the sharded index: sums (and one max) over the owners
    verify = {"entries": chk[0], "nb_kmers": whole["nb_kmers"], "nb_buckets": whole["nb_buckets"], "sum_counts": chk[1], "sum_counts_expected": total_reads * max(L - k + 1, 0),
              "every_kmer_counted_once": chk[1] == total_reads * max(L - k + 1, 0),
              "digest_mod_2_64": (chk[2] + (chk[3] << 22) + (chk[4] << 44)) % (1 << 64)}

    get_leg = None
    if args.get and N == 1:
        # query_sequence over the same reads against the index the last step built (apps/counter.cpp:281-310); not part of `value`
        sums = torch.zeros(n_reads, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        ix.get_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, sums.data_ptr())  # warm-up (buffers)
        ix.profile_reset()
        ix.profile_enable(True)
        torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        for _ in range(args.steps):
            ix.get_packed(d_packed.data_ptr(), d_starts.data_ptr(), n_reads, sums.data_ptr())
        torch.cuda.synchronize(dev)
        tg = (time.perf_counter() - tg) / args.steps
        total = int(sums.sum().item())
        get_leg = {"ms_per_step": round(tg * 1e3, 3), "kmers_queried_per_s": round(n_reads * max(L - k + 1, 0) / tg, 1),
                   "sum_of_counts": total,
                   "kernels_ms_per_step": {n: round(v["ms"] / args.steps, 3) for n, v in ix.profile_read().items() if v["launches"]}}

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = entries_all / dt
        # dominant kernel: largest total device time among the hot-path kernels
        hot = {n: v for n, v in prof.items() if n in ("k_scan", "k_scatter", "k_insert") and v["launches"]}
        dom = max(hot, key=lambda n: hot[n]["ms"]) if hot else None
        roofline = None
        if dom:
            launches_per_step = hot[dom]["launches"] / args.steps
            reads_per_launch = n_reads / launches_per_step  # rank 0's share (all shares are equal to within one read)
            avg_ms = hot[dom]["ms"] / hot[dom]["launches"]
            n_skm = N_SKM.get((k, m), 2.0 * max(L - k + 1, 0) / (k - m + 2))
            bytes_per_launch = b_alg(k, m, b, L, n_skm) * reads_per_launch
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), **pmc_traffic(dom, k, m, b, n_reads),
                        "avg_launch_ms": round(avg_ms, 3), "alg_bytes_per_read": b_alg(k, m, b, L, n_skm),
                        "kernels_ms_per_step": {n: round(v["ms"] / args.steps, 3) for n, v in prof.items() if v["launches"]}}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(k, m, b, L, args.coverage, args.cpu_sample_reads)
        line = {
            "metric": "distinct k-mers indexed/sec (k=%d,m=%d)" % (k, m), "value": round(value, 1), "unit": "k-mers/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%dx MI355X: %s synthetic %d bp reads %s, k=%d m=%d b=%d, uint8 counts, %gx coverage"
                                   % (N, ("%dM" % (args.reads // 1_000_000)) if args.reads >= 1_000_000 else str(args.reads), L,
                                      "per GPU" if args.scaling == "weak" else "in all", k, m, b, args.coverage),
                       "reads_per_gpu": n_reads, "total_reads": total_reads, "part_bits": ix.layout["part_bits"],
                       "ownership": ({"cut_points": "histogram-balanced" if cuts else "equal partition ranges", **getattr(sc, "owner_load", {})} if N > 1 else None), "genome_len": genome_len, "entries_per_step": entries_all / args.steps,
                       "parallelism": "bucket-range shard x%d + all-to-all" % N if N > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line["verify"] = verify
        if get_leg:
            line["get"] = get_leg
        if args.e2e_reads and N == 1:
            line["end_to_end"] = e2e_leg
        print(json.dumps(line))
    if N > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel, k, m, b, n_reads):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE, separate passes, KiB), corrected as
    MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for 16-B-per-lane streaming reads, WRITE_SIZE as is).
    PMC counters cannot be collected from inside this process, so the number is only as good as the passes are current:
    it is reported only when the JSON was taken on this workload AND on kernel sources that hash to what is in the
    tree now (tools/src_hash.py); otherwise traffic is null and traffic_note says which of the two failed."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from src_hash import kernel_source_hash
        now = kernel_source_hash(ROOT)
    except Exception as e:  # noqa: BLE001
        return {"traffic": None, "traffic_note": "kernel sources not hashable: %s" % e}
    why = "no profiles/*_pmc_traffic.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        w = d.get("workload", {})
        if (w.get("k"), w.get("m"), w.get("b"), w.get("reads")) != (k, m, b, n_reads):
            why = "committed PMC passes are of another workload"
            continue
        if d.get("kernel_source_sha256") != now:
            why = "committed PMC passes (%s) were taken on other kernel sources (%s..., now %s...)" % (
                os.path.basename(path), str(d.get("kernel_source_sha256"))[:12], now[:12])
            continue
        kernels = d.get("kernels", {})
        # templates carry their instantiation in the profile name (k_scan2<...>, k_insert<...>)
        pref = {"k_scan": "k_scan2<", "k_insert": "k_insert"}.get(kernel, kernel)
        name = next((n for n in kernels if n.startswith(pref)), None)
        e = kernels.get(name) if name else None
        if not e or not e.get("launches"):
            why = "kernel %s not in %s" % (kernel, os.path.basename(path))
            continue
        corr = d.get("correction", {})
        raw = (e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024 / e["launches"]
        cor = (e.get("FETCH_SIZE", 0) * corr.get("FETCH_SIZE", 1.0) + e.get("WRITE_SIZE", 0) * corr.get("WRITE_SIZE", 1.0)) * 1024 / e["launches"]
        return {"traffic": round(cor), "traffic_raw": round(raw), "traffic_source": os.path.basename(path), "traffic_kernel": name}
    return {"traffic": None, "traffic_note": why}


def cpu_baseline(k, m, b, L, coverage, sample_reads):
    """The reference's CPU path (oracle/_ref, kind "reference") or, where that build is
    absent, this repo's C restatement (kind "port"), timed on a bounded sample of the
    same workload: `sample_reads` reads at the same coverage.  Checker code, timed
    beside the GPU number; never part of the product path."""
    import numpy as np
    import oracle

    try:
        oracle.build(ref=True)
    except Exception:
        pass
    O = oracle.Oracle()
    G = max(int(sample_reads * L / coverage), L + 1)
    reads = O.synth_reads(G, 0, sample_reads, L)
    flat = np.ascontiguousarray(reads.reshape(-1))
    offs = (np.arange(sample_reads + 1, dtype=np.uint64) * np.uint64(L))
    cores = host_cpu_share()
    # the sample goes in slices into one index, and stops early on a host that turns out slow: the leg is bounded in time
    # (~30 s), not only in reads
    slice_reads, budget_s, done = 500_000, 30.0, 0
    if oracle.have_ref():
        R = oracle.Ref()
        h = R.index_new(k, m, b)
        t0 = time.perf_counter()
        while done < sample_reads and (done == 0 or time.perf_counter() - t0 < budget_s):
            n = min(slice_reads, sample_reads - done)
            R.index_insert_reads(h, flat[done * L:(done + n) * L], offs[:n + 1], threads=cores)
            done += n
        dt = time.perf_counter() - t0
        nk, _ = R.index_stats(h)
        R.index_free(h)
        kind, used = "reference", cores
    else:
        h = O.index_new(k, m, b)
        t0 = time.perf_counter()
        while done < sample_reads and (done == 0 or time.perf_counter() - t0 < budget_s):
            n = min(slice_reads, sample_reads - done)
            O.index_insert_reads(h, flat[done * L:(done + n) * L], offs[:n + 1])
            done += n
        dt = time.perf_counter() - t0
        nk, _ = O.index_stats(h)
        O.index_free(h)
        kind, used = "port", 1
    what = ("the reference's own Kmers.cpp / hashing.cpp / Decycling.cpp / buckets.hpp / SuperKmerLight.hpp under this repo's directory driver "
            "(oracle/ref_harness.cpp, OpenMP over reads; NOT apps/counter --mode 1, which parses FASTA under one lock and runs 0.9 M entries/s on 8 vCPU, BASELINE.md): "
            "a stronger baseline than the app; run-to-run spread on a shared host ~40 %") if kind == "reference" else "this repo's plain-C restatement (oracle/brisk_oracle.c), one thread"
    return {"value": round(nk / dt, 1), "unit": "k-mers/s", "cores": used, "kind": kind,
            "sample": "%d of %d synthetic %d bp reads, %gx coverage (genome %d bp): %d entries in %.2f s; %s" % (done, sample_reads, L, coverage, G, nk, dt, what)}


def end_to_end(k, m, b, L, n_reads, d_packed):
    """From a file to the index, as the reference's app times it (apps/counter.cpp:375-381 includes the parsing): the first `n_reads`
    of the bench's synthetic reads (unpacked from the device's 2-bit stream) written as FASTA -- gzipped and plain -- and counted by
    brisk_amd/apps/brisk_count --bulk (C++: FastaBatcher -> brisk_hip_insert_reads -> index), which prints its stage split.  Stages
    overlap (the reader works on batch i + 1 while batch i is counted; big batches are uploaded while the previous piece is
    scanned), so they do not add up to the wall time.  Not `value`: the headline starts with the reads resident in HBM."""
    import subprocess
    import tempfile
    import zlib
    import numpy as np
    import torch
    import brisk_amd
    exe = os.path.join(ROOT, "brisk_amd", "apps", "brisk_count")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    dev = d_packed.device
    tmp = tempfile.mkdtemp(prefix="brisk_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    fa, gz = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "reads.fa.gz")
    t0 = time.perf_counter()
    comp = zlib.compressobj(1, zlib.DEFLATED, 31)
    lut = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dev)  # A0 C1 T2 G3 (Kmers.cpp:442-444)
    sh = torch.arange(30, -2, -2, device=dev, dtype=torch.int64)
    with open(fa, "wb") as f_fa, open(gz, "wb") as f_gz:
        step = 320_000  # reads per piece: a whole number of 16-nt words (320000 * 150 / 16)
        for first in range(0, n_reads, step):
            n = min(step, n_reads - first)
            w0, w1 = first * L // 16, ((first + n) * L + 15) // 16
            words = d_packed[w0:w1].to(torch.int64) & 0xffffffff
            codes = ((words[:, None] >> sh[None, :]) & 3).reshape(-1)[first * L - w0 * 16: first * L - w0 * 16 + n * L]
            reads = lut[codes].reshape(n, L).cpu().numpy()
            rec = np.empty((n, L + 4), dtype=np.uint8)  # ">r\n" + read + "\n"
            rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3:L + 3], rec[:, L + 3] = ord(">"), ord("r"), 10, reads, 10
            blob = rec.tobytes()
            f_fa.write(blob)
            f_gz.write(comp.compress(blob))
        f_gz.write(comp.flush())
    t_files = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    out = {"reads": n_reads, "k": k, "m": m, "b": b, "cores": host_cpu_share(), "files_written_in_s": round(t_files, 1),
           "bytes": {"fa": os.path.getsize(fa), "fa_gz": os.path.getsize(gz)},
           "reference_app": "counter --mode 1 (parsing included): 0.909 M entries/s with 8 threads, 0.381 M with 1, k63 m21 b14, 1 M reads, survey container (BASELINE.md section 2)"}
    for name, path in (("fa_gz", gz), ("fa", fa)):
        best = None
        for rep in range(2):  # (the first run pays the arena mapping and the page cache)
            r = subprocess.run([exe, "--bulk", path, str(k), str(m), str(b), "-"], capture_output=True, text=True, timeout=max(180, n_reads // 50_000), env=dict(os.environ, BRISK_E2E_JSON="1"))  # (the leg runs before the timed steps: it must not be able to hold them up)
            if r.returncode != 0:
                raise RuntimeError("brisk_count failed: " + r.stderr[-300:])
            d = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("E2E "))[4:])
            if best is None or d["wall_s"] < best["wall_s"]:
                best = d
        best["entries_per_s"] = round(best["entries"] / best["wall_s"], 1)
        best["input_MB_per_s"] = round(os.path.getsize(path) / best["wall_s"] / 1e6, 1)
        out[name] = best
    for f in (fa, gz):
        os.remove(f)
    os.rmdir(tmp)
    return out


def host_cpu_share():
    """Threads for the CPU baseline: the cores this process may actually use.  The affinity mask of a GPU box lists every
    core of the host (256) while its cgroup grants a share of them (16 for one GPU): the reference's OpenMP path with
    its lock stripes collapses when it is given more threads than cores (measured on one box, 1 M reads: 16 threads 15.5 M
    entries/s, 32 threads 1.65 M, 256 threads 0.51 M), so the strongest -- the honest -- baseline runs with the share.
    BRISK_CPU_THREADS overrides."""
    if os.environ.get("BRISK_CPU_THREADS"):
        return max(1, int(os.environ["BRISK_CPU_THREADS"]))
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:  # noqa: BLE001
        pass
    if quota is None:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except Exception:  # noqa: BLE001
            pass
    if quota is None:
        quota = 16 if aff > 32 else aff  # no quota visible: a GPU box's share for one GPU
    return max(1, min(aff, quota))


if __name__ == "__main__":
    main()
