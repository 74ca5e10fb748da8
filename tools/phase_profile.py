"""k_insert phase timers on the GPU box: builds tests/_v/libbrisk_phase.so with -DBRISK_PHASE_PROF (debug only; the product
library never carries the timers), runs one counting job and prints each phase's share of the waves' cycles.
    python tools/phase_profile.py [reads] [k m b]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tests", "_v", "libbrisk_phase.so")
if "--build" in sys.argv or not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wno-unused-value",
                           "-DBRISK_PHASE_PROF", "-o", out, os.path.join(ROOT, "brisk_amd", "csrc", "brisk_capi.hip")])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["BRISK_HIP_LIB"] = os.path.relpath(out, os.path.join(ROOT, "brisk_amd"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reads = int(args[0]) if args else 10_000_000
k, m, b = (int(a) for a in args[1:4]) if len(args) >= 4 else (63, 21, 14)
import torch
import brisk_amd
from brisk_amd import hipapi
L = hipapi.load()
L.brisk_hip_debug_phases.argtypes = [C.POINTER(C.c_uint64), C.c_int]
dev = torch.device("cuda", 0)
G = max(int(reads * 150 / 15), 151)
d_packed = torch.zeros((reads * 150 + 15) // 16 + 4, dtype=torch.int32, device=dev)
d_starts = torch.zeros(reads + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()  # torch fills on its own stream, the library works on another: the fill must have landed
ix = brisk_amd.BriskHip(k, m, b)
ix.synth_reads(G, 0, reads, 150, d_packed.data_ptr(), d_starts.data_ptr())
ix.sync()
buf = (C.c_uint64 * 32)()
sbuf = (C.c_uint64 * 8)()
L.brisk_hip_debug_scan_counts.argtypes = [C.POINTER(C.c_uint64), C.c_int]
for rep in range(2):
    ix.clear()
    L.brisk_hip_debug_phases(buf, 1)
    L.brisk_hip_debug_scan_counts(sbuf, 1)
    ix.profile_reset(); ix.profile_enable(True)
    ix.insert_packed(d_packed.data_ptr(), d_starts.data_ptr(), reads)
    ix.sync()
    prof = ix.profile_read()
    L.brisk_hip_debug_phases(buf, 0)
names = ["0 loop/desc/wait", "1 rec store+scan+rec dedupe", "2 pref/table init/inst map", "3 expand+CAS dedupe", "4 -", "5 stream existing",
         "6 compact new", "7 alloc/move", "8 append", "9 epilogue", "10 tail"]
tot = sum(buf[i] for i in range(11)) or 1
print("k_insert ms:", {n: round(v["ms"], 3) for n, v in prof.items() if v["launches"]})
for i, n in enumerate(names):
    print(f"{n:32s} {buf[i]:16d} {100.0 * buf[i] / tot:6.2f} %")
cn = ["partitions", "chunks", "record-dedupe attempts", "expand its (x64 lanes)", "instances", "records", "append passes", "new entries", "CAS rounds x its"]
for i, n in enumerate(cn):
    print(f"{n:28s} {buf[16 + i]:14d}  per partition {buf[16 + i] / max(buf[16], 1):8.3f}")
L.brisk_hip_debug_scan_counts(sbuf, 0)
sn = ["wave-steps", "expiries re-scanned (lanes)", "re-scan rounds", "  of them with two k-mers", "super-k-mers queued", "-", "expiries served by the queue"]
for i, n in enumerate(sn):
    print(f"k_scan2 {n:28s} {sbuf[i]:14d}  per wave-step {sbuf[i] / max(sbuf[0], 1):8.3f}  per read {sbuf[i] / reads:8.3f}")
print(ix.stats())
