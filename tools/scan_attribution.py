"""Where k_scan2's time goes, by leaving pieces out: builds tests/_v/libbrisk_scan_<variant>.so with one -DSCAN_ATTR_* each
(debug only: those builds give WRONG records; the product library never carries them) and times the scan of the bench
workload with each.  The difference to the full build is what the piece costs (re-scans, record emission, class tables,
mixer).    python tools/scan_attribution.py --build     (here)      python tools/scan_attribution.py [reads]   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"full": [], "norescan": ["-DSCAN_ATTR_NORESCAN"], "noemit": ["-DSCAN_ATTR_NOEMIT"], "noclass": ["-DSCAN_ATTR_NOCLASS"], "nomix": ["-DSCAN_ATTR_NOMIX"],
            "nokey": ["-DSCAN_ATTR_NOCLASS", "-DSCAN_ATTR_NOMIX"], "norescan_noemit": ["-DSCAN_ATTR_NORESCAN", "-DSCAN_ATTR_NOEMIT"],
            # the (k-1)-mer's windows run twice (same records: the pass' cost); the re-scan's scalar tie resolution left out (wrong records)
            "prologue_twice": ["-DSCAN_ATTR_PROLOGUE_TWICE"], "rescan_noepilogue_noemit": ["-DSCAN_ATTR_RS_NOEPI", "-DSCAN_ATTR_NOEMIT"]}
vdir = os.path.join(ROOT, "tests", "_v")
if "--build" in sys.argv:
    os.makedirs(vdir, exist_ok=True)
    procs = [subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wno-unused-value", *fl,
                               "-o", os.path.join(vdir, f"libbrisk_scan_{v}.so"), os.path.join(ROOT, "brisk_amd", "csrc", "brisk_capi.hip")]) for v, fl in VARIANTS.items()]
    sys.exit(max(p.wait() for p in procs))
if "--one" in sys.argv:
    v = sys.argv[sys.argv.index("--one") + 1]
    reads = int(sys.argv[-1])
    os.environ["BRISK_HIP_LIB"] = os.path.relpath(os.path.join(vdir, f"libbrisk_scan_{v}.so"), os.path.join(ROOT, "brisk_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import brisk_amd
    dev = torch.device("cuda", 0)
    d_packed = torch.zeros((reads * 150 + 15) // 16 + 4, dtype=torch.int32, device=dev)
    d_starts = torch.zeros(reads + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # torch fills on its own stream, the library works on another: the fill must have landed
    ix = brisk_amd.BriskHip(63, 21, 14)
    ix.synth_reads(max(reads * 10, 151), 0, reads, 150, d_packed.data_ptr(), d_starts.data_ptr())
    ix.sync()
    for rep in range(2):
        ix.clear()
        ix.profile_reset(); ix.profile_enable(True)
        try:
            ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), reads)
            ix.sync()
        except Exception as e:  # a variant's records may be nonsense to the insert
            print(v, "insert failed:", e)
        prof = ix.profile_read()
    print(f"{v:18s} k_scan {prof['k_scan']['ms']:8.3f} ms   k_insert {prof['k_insert']['ms']:8.3f} ms")
    sys.exit(0)
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
for v in VARIANTS:  # one process per variant: the library is loaded once per process
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", v, str(reads)])
