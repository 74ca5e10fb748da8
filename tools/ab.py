"""A/B timing of library variants: the same sources compiled with different -D flags, timed on the same workload.
    python tools/ab.py --build NAME[:-DFLAG[,-DFLAG...]] ...     (here: hipcc cross-compiles, variants in parallel)
    python tools/ab.py --run K M B READS [NAME ...]                (GPU box: every built variant, or the named ones; BRISK_AB_PART_BITS=N: brisk_hip_options.part_bits)
Variants live in tests/_v/libbrisk_ab_<NAME>.so (git-ignored, travel with gpurun).  Each is timed in its own process
(the library is loaded once per process): synthetic reads resident on the device, one warm-up job, then two timed jobs;
prints the per-kernel HIP-event times of the last job, the index digest (equal digests <=> equal multisets: a variant
that changes results shows here) and the job's wall time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vdir = os.path.join(ROOT, "tests", "_v")


def lib(name):
    return os.path.join(vdir, f"libbrisk_ab_{name}.so")


if "--build" in sys.argv:
    os.makedirs(vdir, exist_ok=True)
    procs = []
    for spec in sys.argv[sys.argv.index("--build") + 1:]:
        name, _, fl = spec.partition(":")
        flags = [f for f in fl.split(",") if f]
        procs.append((name, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wno-unused-value",
                                              *flags, "-o", lib(name), os.path.join(ROOT, "brisk_amd", "csrc", "brisk_capi.hip")])))
    rc = 0
    for name, p in procs:
        r = p.wait()
        print("built" if r == 0 else "FAILED", name)
        rc = max(rc, r)
    sys.exit(rc)

if "--one" in sys.argv:
    at = sys.argv.index("--one")
    name, k, m, b, reads = sys.argv[at + 1], *[int(x) for x in sys.argv[at + 2:at + 6]]
    os.environ["BRISK_HIP_LIB"] = os.path.relpath(lib(name), os.path.join(ROOT, "brisk_amd"))
    sys.path.insert(0, ROOT)
    import time
    import torch
    import brisk_amd
    dev = torch.device("cuda", 0)
    d_packed = torch.zeros((reads * 150 + 15) // 16 + 4, dtype=torch.int32, device=dev)
    d_starts = torch.zeros(reads + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # torch fills on its own stream, the library works on another: the fill must have landed
    ix = brisk_amd.BriskHip(k, m, b, part_bits=int(os.environ.get("BRISK_AB_PART_BITS", "0")))
    ix.synth_reads(max(reads * 10, 151), 0, reads, 150, d_packed.data_ptr(), d_starts.data_ptr())
    ix.sync()
    wall = 0.0
    for rep in range(3):
        ix.clear()
        ix.profile_reset(); ix.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), reads)
        ix.sync()
        wall = (time.perf_counter() - t0) * 1e3
        prof = ix.profile_read()
    ks = " ".join(f"{n} {v['ms']:.3f}" for n, v in prof.items() if v["launches"] and v["ms"] > 0.05)
    print(f"{name:14s} wall {wall:8.3f} ms | {ks} | digest {ix.checksum()}", flush=True)
    sys.exit(0)

if "--run" in sys.argv:
    at = sys.argv.index("--run")
    k, m, b, reads = sys.argv[at + 1:at + 5]
    names = sys.argv[at + 5:] or sorted(f[len("libbrisk_ab_"):-3] for f in os.listdir(vdir) if f.startswith("libbrisk_ab_") and f.endswith(".so"))
    for name in names:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", name, k, m, b, reads])
