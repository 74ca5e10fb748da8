"""ctypes binding of include/brisk_hip.h (one-to-one; no logic)."""
from __future__ import annotations

import atexit
import ctypes as C
import math
import os
import subprocess
import sys
import weakref
from typing import Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_LIB_NAME = os.environ.get("BRISK_HIP_LIB", "libbrisk_hip.so")

STATUS = {0: "OK", 1: "EINVAL", 2: "EUNSUPPORTED", 3: "EHIP", 4: "ENOMEM", 5: "ECAPACITY", 6: "ENODEVICE"}
ECAPACITY = 5


class BriskHipError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        self.code = code
        super().__init__(f"brisk_hip: {STATUS.get(code, code)} {msg}".strip())


def library_path() -> str:
    return os.path.join(HERE, _LIB_NAME)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU). In-tree output."""
    out = library_path()
    srcs = [os.path.join(HERE, "csrc", f) for f in ("brisk_capi.hip", "brisk_kernels.hip", "brisk_scan.hip", "brisk_partition.hip", "brisk_insert.hip",
                                                    "brisk_readout.hip", "brisk_device.h")]
    srcs.append(os.path.join(ROOT, "include", "brisk_hip.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden",
           "-Wno-unused-value", "-o", out, srcs[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_apps(verbose: bool = False) -> dict:
    """C++ host side: this repo's brisk_count over the facade headers, and -- where the reference
    tree exists -- the reference's own apps/counter.cpp compiled UNCHANGED against
    brisk_amd/include (its CLI11/zstr third-party headers are taken from the reference tree)."""
    apps = os.path.join(HERE, "apps")
    inc = ["-I" + os.path.join(HERE, "include"), "-I" + os.path.join(ROOT, "include")]
    link = ["-L" + HERE, "-lbrisk_hip", "-Wl,-rpath,$ORIGIN/.."]
    out = {}
    cmd = ["g++", "-std=gnu++17", "-O2", "-pthread"] + inc + [os.path.join(apps, "brisk_count.cpp")] + link + ["-lz", "-o", os.path.join(apps, "brisk_count")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    out["brisk_count"] = os.path.join(apps, "brisk_count")
    # the multi-GPU job from C++: one process per GPU, RCCL grouped send/recv (apps/brisk_shard.cpp)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=gnu++17", "-O2", "-pthread", "-D__HIP_PLATFORM_AMD__"] + inc + ["-I" + os.path.join(rocm, "include"), os.path.join(apps, "brisk_shard.cpp")] + link + \
          ["-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", os.path.join(apps, "brisk_shard")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    out["brisk_shard"] = os.path.join(apps, "brisk_shard")
    ref_counter = "/root/reference/apps/counter.cpp"
    if os.path.exists(ref_counter):
        cmd = ["g++", "-std=gnu++17", "-O2", "-w", "-include", "cstdint", "-fopenmp"] + inc + ["-I/root/reference/apps", ref_counter] + link + \
              ["-lz", "-o", os.path.join(apps, "counter_ref")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        out["counter_ref"] = os.path.join(apps, "counter_ref")
    return out


def coef_table(m: int) -> np.ndarray:
    """DecyclingSet(m) coefficients (reference brisk/Decycling.cpp:7-13), computed on
    the HOST with libm; the device only ever sees these bits."""
    unit = 2 * math.pi / m
    coef = np.zeros(4 * m, dtype=np.float64)
    for j in range(1, m):
        s = math.sin(unit * float(j))
        coef[4 * j + 1] = s
        coef[4 * j + 2] = 2 * s
        coef[4 * j + 3] = 3 * s
    return coef


class _Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p), ("part_bits", C.c_uint32),
                ("owner_rank", C.c_uint32), ("n_owners", C.c_uint32), ("arena_entries", C.c_uint64),
                ("max_batch_reads", C.c_uint64), ("entry_ids", C.c_uint32), ("immediate_inserts", C.c_uint32)]


class _Layout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("k", "m", "b", "m_reduc", "compacted_size", "allocated_bytes", "record_words",
                                          "part_bits", "n_owners", "owner_rank", "ext_bits", "cls_bits", "cls_width")]


_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

# every symbol include/brisk_hip.h declares
SYMBOLS = [
    "brisk_hip_abi_version", "brisk_hip_create", "brisk_hip_destroy", "brisk_hip_clear", "brisk_hip_last_error", "brisk_hip_sync",
    "brisk_hip_get_layout", "brisk_hip_insert_reads", "brisk_hip_insert_packed", "brisk_hip_get_reads", "brisk_hip_lookup",
    "brisk_hip_enumerate", "brisk_hip_stats", "brisk_hip_memory_info", "brisk_hip_insert_slack", "brisk_hip_reallocate", "brisk_hip_checksum", "brisk_hip_scan_packed", "brisk_hip_scan_bound", "brisk_hip_route_records",
    "brisk_hip_get_packed", "brisk_hip_insert_records", "brisk_hip_set_owner_cuts", "brisk_hip_export_hist", "brisk_hip_export_hist_add", "brisk_hip_insert_records_hist", "brisk_hip_scan_query", "brisk_hip_route_tagged", "brisk_hip_query_records", "brisk_hip_pack_ascii", "brisk_hip_synth_reads", "brisk_hip_debug_order_keys", "brisk_hip_scan_sequence", "brisk_hip_upsert_kmers", "brisk_hip_find_kmers",
    "brisk_hip_enumerate_ids", "brisk_hip_profile_enable",
    "brisk_hip_profile_read", "brisk_hip_profile_reset",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise BriskHipError(6, f"{path} is missing: run __graft_entry__.build() (no CPU fallback exists)")
    L = C.CDLL(path)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.brisk_hip_abi_version.restype = u32
    L.brisk_hip_create.argtypes = [C.POINTER(vp), C.c_uint8, C.c_uint8, C.c_uint8, u32, C.POINTER(C.c_double), C.POINTER(_Options)]
    L.brisk_hip_destroy.argtypes = [vp]
    L.brisk_hip_clear.argtypes = [vp]
    L.brisk_hip_last_error.argtypes = [vp]
    L.brisk_hip_last_error.restype = C.c_char_p
    L.brisk_hip_sync.argtypes = [vp]
    L.brisk_hip_get_layout.argtypes = [vp, C.POINTER(_Layout)]
    L.brisk_hip_insert_reads.argtypes = [vp, _u8p, _u64p, u64]
    L.brisk_hip_insert_packed.argtypes = [vp, vp, vp, u64]
    L.brisk_hip_get_packed.argtypes = [vp, vp, vp, u64, vp]
    L.brisk_hip_get_reads.argtypes = [vp, _u8p, _u64p, u64, _u64p]
    L.brisk_hip_lookup.argtypes = [vp, _u64p, _u64p, _u8p, u64, _u8p, _u8p]
    L.brisk_hip_enumerate.argtypes = [vp, C.POINTER(u64), _u64p, _u64p, _u8p, _u8p, u64, C.POINTER(u64)]
    L.brisk_hip_stats.argtypes = [vp] + [C.POINTER(u64)] * 5
    L.brisk_hip_checksum.argtypes = [vp, _u64p]
    L.brisk_hip_memory_info.argtypes = [vp, _u64p]
    L.brisk_hip_insert_slack.argtypes = [vp, C.POINTER(u64)]
    L.brisk_hip_reallocate.argtypes = [vp, vp]
    L.brisk_hip_scan_packed.argtypes = [vp, vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.brisk_hip_scan_bound.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.brisk_hip_route_records.argtypes = [vp, vp, u64, vp, _u64p]
    L.brisk_hip_insert_records.argtypes = [vp, vp, u64]
    L.brisk_hip_set_owner_cuts.argtypes = [vp, _u64p]
    L.brisk_hip_export_hist.argtypes = [vp, vp, _u64p]
    L.brisk_hip_export_hist_add.argtypes = [vp, vp, _u64p]
    L.brisk_hip_insert_records_hist.argtypes = [vp, vp, u64, vp, u32]
    L.brisk_hip_scan_query.argtypes = [vp, vp, vp, u64, vp, vp, u64, C.POINTER(u64)]
    L.brisk_hip_route_tagged.argtypes = [vp, vp, vp, u64, vp, vp, _u64p]
    L.brisk_hip_query_records.argtypes = [vp, vp, u64, vp]
    L.brisk_hip_pack_ascii.argtypes = [vp, vp, u64, vp]
    L.brisk_hip_synth_reads.argtypes = [vp, u64, u64, u64, u32, u64, u64, vp, vp]
    _u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    L.brisk_hip_scan_sequence.argtypes = [vp, C.c_char_p, u64, u64, _u64p, _u32p, _u64p, _u64p, _u8p, C.POINTER(u64)]
    L.brisk_hip_upsert_kmers.argtypes = [vp, _u64p, _u64p, _u8p, u64, _u32p, _u8p]
    L.brisk_hip_find_kmers.argtypes = [vp, _u64p, _u64p, _u8p, u64, _u32p]
    L.brisk_hip_enumerate_ids.argtypes = [vp, C.POINTER(u64), _u64p, _u64p, _u8p, _u32p, u64, C.POINTER(u64)]
    L.brisk_hip_debug_order_keys.argtypes = [vp, _u64p, u64, i32, _u64p]
    L.brisk_hip_profile_enable.argtypes = [vp, i32]
    L.brisk_hip_profile_read.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double)]
    L.brisk_hip_profile_reset.argtypes = [vp]
    for s in SYMBOLS:
        if getattr(L, s).restype is C.c_int:
            pass
    _lib = L
    return L


def _pack_reads(seqs) -> Tuple[np.ndarray, np.ndarray]:
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    flat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(flat), offs


_live = weakref.WeakSet()


@atexit.register
def _close_all_handles():
    # destroy device state while the HIP runtime is still up: finalisers that run during
    # interpreter teardown would call into a runtime that may already be gone
    for ix in list(_live):
        try:
            ix.close()
        except Exception:
            pass


class BriskHip:
    """One index handle.  Methods map one-to-one onto the C-ABI."""

    def __init__(self, k: int, m: int, b: int, device: int = 0, stream: Optional[int] = None, part_bits: int = 0,
                 owner_rank: int = 0, n_owners: int = 1, arena_entries: int = 0, max_batch_reads: int = 0,
                 entry_ids: bool = False, immediate_inserts: bool = False):
        self.L = load()
        self.h = C.c_void_p()
        self.k, self.m, self.b = k, m, b
        opt = _Options(C.sizeof(_Options), device, stream, part_bits, owner_rank, n_owners, arena_entries, max_batch_reads,
                       1 if entry_ids else 0, 1 if immediate_inserts else 0)
        coef = coef_table(m) if 1 <= m <= 31 else np.zeros(4, np.float64)
        rc = self.L.brisk_hip_create(C.byref(self.h), k, m, b, 1, coef.ctypes.data_as(C.POINTER(C.c_double)), C.byref(opt))
        if rc:
            self.h = C.c_void_p()
            raise BriskHipError(rc, f"create(k={k},m={m},b={b})")
        _live.add(self)
        lay = _Layout()
        self._chk(self.L.brisk_hip_get_layout(self.h, C.byref(lay)))
        self.layout = {n: getattr(lay, n) for n, _ in _Layout._fields_}
        self.record_words = lay.record_words

    def _chk(self, rc: int):
        if rc:
            raise BriskHipError(rc, (self.L.brisk_hip_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.brisk_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():  # module globals are already gone late in interpreter shutdown
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- bulk host paths
    def insert_reads(self, seqs: Sequence) -> None:
        flat, offs = _pack_reads(seqs)
        self.insert_flat(flat, offs)

    def insert_flat(self, flat: np.ndarray, offs: np.ndarray) -> None:
        if len(flat) == 0:
            flat = np.zeros(1, np.uint8)
        self._chk(self.L.brisk_hip_insert_reads(self.h, flat, offs, len(offs) - 1))

    def get_reads(self, seqs: Sequence) -> np.ndarray:
        flat, offs = _pack_reads(seqs)
        out = np.zeros(len(offs) - 1, np.uint64)
        if len(flat) == 0:
            flat = np.zeros(1, np.uint8)
        self._chk(self.L.brisk_hip_get_reads(self.h, flat, offs, len(offs) - 1, out))
        return out

    def lookup(self, lo, hi, idx) -> Tuple[np.ndarray, np.ndarray]:
        lo = np.ascontiguousarray(lo, np.uint64)
        hi = np.ascontiguousarray(hi, np.uint64)
        idx = np.ascontiguousarray(idx, np.uint8)
        n = len(lo)
        data = np.zeros(max(n, 1), np.uint8)
        found = np.zeros(max(n, 1), np.uint8)
        self._chk(self.L.brisk_hip_lookup(self.h, lo, hi, idx, n, data, found))
        return data[:n], found[:n]

    def enumerate(self, chunk: int = 1 << 20):
        """All entries: (lo, hi, minimizer_idx, count) arrays, k-mers unhashed."""
        cur = C.c_uint64(0)
        n = C.c_uint64(0)
        los, his, idxs, cnts = [], [], [], []
        cap = chunk
        while True:
            lo = np.zeros(cap, np.uint64)
            hi = np.zeros(cap, np.uint64)
            idx = np.zeros(cap, np.uint8)
            cnt = np.zeros(cap, np.uint8)
            rc = self.L.brisk_hip_enumerate(self.h, C.byref(cur), lo, hi, idx, cnt, cap, C.byref(n))
            if rc == ECAPACITY:
                cap *= 4
                continue
            self._chk(rc)
            if n.value == 0:
                break
            los.append(lo[: n.value]); his.append(hi[: n.value]); idxs.append(idx[: n.value]); cnts.append(cnt[: n.value])
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        return cat(los, np.uint64), cat(his, np.uint64), cat(idxs, np.uint8), cat(cnts, np.uint8)

    def stats(self) -> dict:
        v = [C.c_uint64() for _ in range(5)]
        self._chk(self.L.brisk_hip_stats(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("nb_buckets", "nb_skmers", "nb_kmers", "memory_bytes", "largest_bucket"), (x.value for x in v)))

    def reallocate_into(self, fresh: "BriskHip") -> None:
        """Brisk::reallocate: move every entry into `fresh`, an empty index over the same k with its own (m, b)."""
        rc = self.L.brisk_hip_reallocate(self.h, fresh.h)
        if rc:
            raise BriskHipError(rc, self.L.brisk_hip_last_error(fresh.h).decode())

    def memory_info(self) -> dict:
        out = np.zeros(4, np.uint64)
        self._chk(self.L.brisk_hip_memory_info(self.h, out))
        return dict(zip(("arena_mapped", "arena_reserved", "pooled", "retired_va"), (int(v) for v in out)))

    def insert_slack(self) -> int:
        """arena entries the single-pass insert reserves beyond a batch's own need (resident insert waves x chunk entries)"""
        v = C.c_uint64()
        self._chk(self.L.brisk_hip_insert_slack(self.h, C.byref(v)))
        return v.value

    def checksum(self) -> tuple:
        """(entries, sum of counts, order-independent digest) of the whole index"""
        out = np.zeros(3, np.uint64)
        self._chk(self.L.brisk_hip_checksum(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def clear(self):
        self._chk(self.L.brisk_hip_clear(self.h))

    def sync(self):
        self._chk(self.L.brisk_hip_sync(self.h))

    # ---- device-buffer paths (pointers are ints: tensor.data_ptr())
    def insert_packed(self, d_packed: int, d_starts: int, n_reads: int):
        self._chk(self.L.brisk_hip_insert_packed(self.h, d_packed, d_starts, n_reads))

    def get_packed(self, d_packed: int, d_starts: int, n_reads: int, d_sums: int):
        """per-read sums of counts (query_sequence), reads and sums on the device"""
        self._chk(self.L.brisk_hip_get_packed(self.h, d_packed, d_starts, n_reads, d_sums))

    def scan_bound(self, d_starts: int, n_reads: int) -> int:
        out = C.c_uint64()
        self._chk(self.L.brisk_hip_scan_bound(self.h, d_starts, n_reads, C.byref(out)))
        return out.value

    def scan_packed(self, d_packed: int, d_starts: int, n_reads: int, d_records: int, cap: int) -> int:
        out = C.c_uint64()
        rc = self.L.brisk_hip_scan_packed(self.h, d_packed, d_starts, n_reads, d_records, cap, C.byref(out))
        if rc == ECAPACITY:
            raise BriskHipError(rc, f"scan needs {out.value} records, cap {cap}")
        self._chk(rc)
        return out.value

    def route_records(self, d_records: int, n: int, d_out: int) -> np.ndarray:
        counts = np.zeros(max(self.layout["n_owners"], 1), np.uint64)
        self._chk(self.L.brisk_hip_route_records(self.h, d_records, n, d_out, counts))
        return counts

    def insert_records(self, d_records: int, n: int):
        self._chk(self.L.brisk_hip_insert_records(self.h, d_records, n))

    def set_owner_cuts(self, first_partition) -> None:
        """owner o holds partitions [first_partition[o], first_partition[o + 1]); the same array on every rank, before anything is routed"""
        cuts = np.ascontiguousarray(first_partition, np.uint64)
        assert len(cuts) == self.layout["n_owners"] + 1
        self._chk(self.L.brisk_hip_set_owner_cuts(self.h, cuts))

    def export_hist(self, d_hist_out: int) -> np.ndarray:
        """copy the last scan's per-partition histogram (2^part_bits u64) to d_hist_out; returns the slice length per owner"""
        lens = np.zeros(max(self.layout["n_owners"], 1), np.uint64)
        self._chk(self.L.brisk_hip_export_hist(self.h, d_hist_out, lens))
        return lens

    def export_hist_add(self, d_hist_acc: int) -> np.ndarray:
        """add the last scan's per-partition histogram to d_hist_acc (2^part_bits u64); returns the slice length per owner"""
        lens = np.zeros(max(self.layout["n_owners"], 1), np.uint64)
        self._chk(self.L.brisk_hip_export_hist_add(self.h, d_hist_acc, lens))
        return lens

    def insert_records_hist(self, d_records: int, n: int, d_hist_slices: int, n_slices: int):
        self._chk(self.L.brisk_hip_insert_records_hist(self.h, d_records, n, d_hist_slices, n_slices))

    def scan_query(self, d_packed: int, d_starts: int, n_reads: int, d_records: int, d_tags: int, cap: int) -> int:
        """query-mode scan: records + the index of the read each came from (u32)"""
        out = C.c_uint64()
        rc = self.L.brisk_hip_scan_query(self.h, d_packed, d_starts, n_reads, d_records, d_tags, cap, C.byref(out))
        if rc == ECAPACITY:
            raise BriskHipError(rc, f"scan needs {out.value} records, cap {cap}")
        self._chk(rc)
        return out.value

    def route_tagged(self, d_records: int, d_tags: int, n: int, d_out: int, d_tags_out: int) -> np.ndarray:
        counts = np.zeros(max(self.layout["n_owners"], 1), np.uint64)
        self._chk(self.L.brisk_hip_route_tagged(self.h, d_records, d_tags, n, d_out, d_tags_out, counts))
        return counts

    def query_records(self, d_records: int, n: int, d_sums: int):
        """d_sums[i] (u64) = sum of the counts of record i's k-mers present in this index"""
        self._chk(self.L.brisk_hip_query_records(self.h, d_records, n, d_sums))

    def pack_ascii(self, d_bases: int, n_bases: int, d_packed: int):
        self._chk(self.L.brisk_hip_pack_ascii(self.h, d_bases, n_bases, d_packed))

    def synth_reads(self, genome_len: int, first_read: int, n_reads: int, read_len: int, d_packed: int, d_starts: int,
                    seed_g: int = 1, seed_r: int = 2):
        self._chk(self.L.brisk_hip_synth_reads(self.h, genome_len, first_read, n_reads, read_len, seed_g, seed_r, d_packed, d_starts))

    # ---- per-call API (entry-id mode), as the C++ facade uses it
    def scan_sequence(self, seq):
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nk = max(len(s) - self.k + 1, 1)
        ret = np.zeros(nk, np.uint64); cnt = np.zeros(nk, np.uint32)
        lo = np.zeros(nk, np.uint64); hi = np.zeros(nk, np.uint64); idx = np.zeros(nk, np.uint8)
        n = C.c_uint64()
        self._chk(self.L.brisk_hip_scan_sequence(self.h, s, len(s), nk, ret, cnt, lo, hi, idx, C.byref(n)))
        t = int(cnt[: n.value].sum())
        return ret[: n.value], cnt[: n.value], lo[:t], hi[:t], idx[:t]

    def upsert_kmers(self, lo, hi, idx):
        lo = np.ascontiguousarray(lo, np.uint64); hi = np.ascontiguousarray(hi, np.uint64); idx = np.ascontiguousarray(idx, np.uint8)
        ids = np.zeros(max(len(lo), 1), np.uint32); new = np.zeros(max(len(lo), 1), np.uint8)
        self._chk(self.L.brisk_hip_upsert_kmers(self.h, lo, hi, idx, len(lo), ids, new))
        return ids[: len(lo)], new[: len(lo)]

    def find_kmers(self, lo, hi, idx):
        lo = np.ascontiguousarray(lo, np.uint64); hi = np.ascontiguousarray(hi, np.uint64); idx = np.ascontiguousarray(idx, np.uint8)
        ids = np.zeros(max(len(lo), 1), np.uint32)
        self._chk(self.L.brisk_hip_find_kmers(self.h, lo, hi, idx, len(lo), ids))
        return ids[: len(lo)]

    def enumerate_ids(self, chunk: int = 1 << 20):
        cur = C.c_uint64(0); n = C.c_uint64(0)
        out = [[], [], [], []]
        cap = chunk
        while True:
            lo = np.zeros(cap, np.uint64); hi = np.zeros(cap, np.uint64); idx = np.zeros(cap, np.uint8); ids = np.zeros(cap, np.uint32)
            rc = self.L.brisk_hip_enumerate_ids(self.h, C.byref(cur), lo, hi, idx, ids, cap, C.byref(n))
            if rc == ECAPACITY:
                cap *= 4
                continue
            self._chk(rc)
            if n.value == 0:
                break
            for a, v in zip(out, (lo, hi, idx, ids)):
                a.append(v[: n.value])
        dts = (np.uint64, np.uint64, np.uint8, np.uint32)
        return tuple(np.concatenate(a) if a else np.zeros(0, dt) for a, dt in zip(out, dts))

    def debug_order_keys(self, mmers, exact: bool = False) -> np.ndarray:
        x = np.ascontiguousarray(mmers, np.uint64)
        out = np.zeros(max(len(x), 1), np.uint64)
        self._chk(self.L.brisk_hip_debug_order_keys(self.h, x, len(x), 1 if exact else 0, out))
        return out[: len(x)]

    # ---- measurement
    def profile_enable(self, on: bool = True):
        self._chk(self.L.brisk_hip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.L.brisk_hip_profile_reset(self.h))

    def profile_read(self) -> dict:
        n = C.c_uint32()
        names = (C.c_char_p * 16)()
        launches = (C.c_uint64 * 16)()
        ms = (C.c_double * 16)()
        self._chk(self.L.brisk_hip_profile_read(self.h, C.byref(n), names, launches, ms))
        return {names[i].decode(): {"launches": launches[i], "ms": ms[i]} for i in range(n.value)}
