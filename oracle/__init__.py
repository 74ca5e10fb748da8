"""CPU oracle for the Brisk hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  Nothing under brisk_amd/ imports it.

Two libraries, same calling conventions:
  * ``Oracle``  -- oracle/libbrisk_oracle.so, this repo's plain-C restatement
                   (brisk_oracle.c), built anywhere with gcc.
  * ``Ref``     -- oracle/_ref/libbrisk_ref.so, the REAL reference sources
                   compiled where they lie (only buildable where /root/reference
                   exists; the built .so travels to the GPU box, sources never do).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libbrisk_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libbrisk_ref.so")
REFERENCE_ROOT = "/root/reference"

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile the restatement and, where /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref() -> bool:
    return os.path.exists(REF_SO)


# ---------------------------------------------------------------------------
# helpers shared by tests: sequences <-> flat buffers, k-mer strings
def pack_reads(seqs: Sequence[bytes | str]) -> Tuple[np.ndarray, np.ndarray]:
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    flat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return flat, offs


_NT = "ACTG"  # code -> letter (A0 C1 T2 G3; Kmers.cpp:442-444)


def kmer2str(lo: int, hi: int, k: int) -> str:
    v = (int(hi) << 64) | int(lo)
    return "".join(_NT[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def str2kmer(s: str) -> Tuple[int, int]:
    v = 0
    for ch in s:
        v = (v << 2) | ((ord(ch) >> 1) & 3)
    return v & ((1 << 64) - 1), v >> 64


def multiset_lines(lo, hi, idx, cnt, k: int) -> List[str]:
    """Sorted ``KMER idx count`` lines: the parity object."""
    return sorted(f"{kmer2str(l, h, k)} {int(i)} {int(c)}" for l, h, i, c in zip(lo, hi, idx, cnt))


def _fmix(z: int) -> int:
    M = (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def digest(lo, hi, idx, cnt) -> tuple:
    """The order-independent digest brisk_hip_checksum defines (include/brisk_hip.h)."""
    d = 0
    for a, b, c, e in zip(lo, hi, idx, cnt):
        d = (d + _fmix(int(a) ^ _fmix(int(b) ^ _fmix((int(c) << 8) | int(e))))) & ((1 << 64) - 1)
    return len(lo), int(sum(int(x) for x in cnt)), d


def fasta_sequences(text: str) -> List[str]:
    """The harness's record/segment rules (counter.cpp:130-190):
    a record is every line up to the next '>'; it is cut at the first character
    outside [ACGTacgt], the remainder (from the next valid base) is handled as a
    separate sequence; upper-cased."""
    out: List[str] = []
    rec: List[str] = []

    def flush():
        s = "".join(rec)
        rec.clear()
        while s:
            i = 0
            while i < len(s) and s[i] in "ACGTacgt":
                i += 1
            out.append(s[:i].upper())
            if i >= len(s):
                break
            j = i
            while j < len(s) and s[j] not in "ACGTacgt":
                j += 1
            s = s[j:]

    for line in text.splitlines():
        if line.startswith(">"):
            flush()
        else:
            rec.append(line.strip())
    flush()
    return [s for s in out if s]


# ---------------------------------------------------------------------------
class _Lib:
    prefix = ""
    path = ""

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(f"{self.path} missing: run oracle.build()")
        self.lib = C.CDLL(self.path)
        p = self.prefix
        L = self.lib
        self._enumerate = getattr(L, p + "enumerate")
        self._enumerate.restype = C.c_int64
        self._enumerate.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_uint, u64p, u32p, C.c_uint64,
                                    u64p, u64p, u8p, u64p, C.c_uint64, C.POINTER(C.c_uint64)]
        getattr(L, p + "coef_table").argtypes = [C.c_uint, f64p]
        getattr(L, p + "coef_table").restype = None
        getattr(L, p + "rcbc").restype = C.c_uint64
        getattr(L, p + "rcbc").argtypes = [C.c_uint64, C.c_uint]
        getattr(L, p + "rcb").restype = None
        getattr(L, p + "rcb").argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        getattr(L, p + "get_minimizer").restype = C.c_uint64
        getattr(L, p + "get_minimizer").argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint,
                                                    C.POINTER(C.c_uint8), C.POINTER(C.c_int)]
        getattr(L, p + "compacted").restype = None
        getattr(L, p + "compacted").argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        for name, res, args in [
            ("index_new", C.c_void_p, [C.c_uint, C.c_uint, C.c_uint]),
            ("index_free", None, [C.c_void_p]),
            ("index_nb_kmers", C.c_uint64, [C.c_void_p]),
            ("index_nb_buckets", C.c_uint64, [C.c_void_p]),
            ("index_dump", C.c_uint64, [C.c_void_p, u64p, u64p, u8p, u8p, C.c_uint64]),
            ("index_get", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint]),
            ("index_query_reads", C.c_int, [C.c_void_p, u8p, u64p, C.c_uint64, u64p]),
        ]:
            f = getattr(L, p + name)
            f.restype = res
            f.argtypes = args

    # -- units
    def coef_table(self, m: int) -> np.ndarray:
        out = np.zeros(4 * m, dtype=np.float64)
        getattr(self.lib, self.prefix + "coef_table")(m, out)
        return out

    def rcbc(self, x: int, n: int) -> int:
        return int(getattr(self.lib, self.prefix + "rcbc")(x, n))

    def rcb(self, lo: int, hi: int, n: int) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        getattr(self.lib, self.prefix + "rcb")(lo, hi, n, C.byref(a), C.byref(b))
        return a.value, b.value

    def get_minimizer(self, lo: int, hi: int, K: int, m: int) -> Tuple[int, int, int]:
        pos, rev = C.c_uint8(), C.c_int()
        mini = getattr(self.lib, self.prefix + "get_minimizer")(lo, hi, K, m, C.byref(pos), C.byref(rev))
        return int(mini), pos.value, rev.value

    def compacted(self, lo: int, hi: int, b: int, idxp: int) -> Tuple[int, int]:
        a, c = C.c_uint64(), C.c_uint64()
        getattr(self.lib, self.prefix + "compacted")(lo, hi, b, idxp, C.byref(a), C.byref(c))
        return a.value, c.value

    # -- enumerator stream
    def enumerate(self, seq: str | bytes, k: int, m: int):
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nk = max(len(s) - k + 1, 0)
        skm_ret = np.zeros(nk + 1, np.uint64)
        skm_n = np.zeros(nk + 1, np.uint32)
        lo = np.zeros(nk + 1, np.uint64)
        hi = np.zeros(nk + 1, np.uint64)
        idx = np.zeros(nk + 1, np.uint8)
        mini = np.zeros(nk + 1, np.uint64)
        nkm = C.c_uint64()
        n = self._enumerate(s, len(s), k, m, skm_ret, skm_n, nk + 1, lo, hi, idx, mini, nk + 1, C.byref(nkm))
        if n < 0:
            raise ValueError("enumerate failed")
        t = nkm.value
        return skm_ret[:n], skm_n[:n], lo[:t], hi[:t], idx[:t], mini[:t]

    # -- index
    def index_new(self, k: int, m: int, b: int):
        h = getattr(self.lib, self.prefix + "index_new")(k, m, b)
        if not h:
            raise ValueError(f"invalid parameters k={k} m={m} b={b}")
        return h

    def index_free(self, h) -> None:
        getattr(self.lib, self.prefix + "index_free")(h)

    def index_stats(self, h) -> Tuple[int, int]:
        return (int(getattr(self.lib, self.prefix + "index_nb_kmers")(h)),
                int(getattr(self.lib, self.prefix + "index_nb_buckets")(h)))

    def index_dump(self, h):
        n = int(getattr(self.lib, self.prefix + "index_nb_kmers")(h))
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        idx = np.zeros(n, np.uint8)
        cnt = np.zeros(n, np.uint8)
        got = getattr(self.lib, self.prefix + "index_dump")(h, lo, hi, idx, cnt, n)
        assert got == n, (got, n)
        return lo, hi, idx, cnt

    def index_get(self, h, lo: int, hi: int, idx: int) -> int:
        return int(getattr(self.lib, self.prefix + "index_get")(h, lo, hi, idx))

    def index_query_reads(self, h, flat: np.ndarray, offs: np.ndarray) -> np.ndarray:
        out = np.zeros(len(offs) - 1, np.uint64)
        getattr(self.lib, self.prefix + "index_query_reads")(h, flat, offs, len(offs) - 1, out)
        return out

    def count(self, seqs: Sequence[str | bytes], k: int, m: int, b: int, **kw):
        """Count ``seqs``; returns (sorted multiset lines, nb_kmers, nb_buckets)."""
        flat, offs = pack_reads(seqs)
        h = self.index_new(k, m, b)
        try:
            self.index_insert_reads(h, flat, offs, **kw)
            nb_kmers, nb_buckets = self.index_stats(h)
            lines = multiset_lines(*self.index_dump(h), k)
        finally:
            self.index_free(h)
        return lines, nb_kmers, nb_buckets


class Oracle(_Lib):
    prefix = "bo_"
    path = ORACLE_SO

    def __init__(self):
        super().__init__()
        L = self.lib
        L.bo_index_insert_reads.restype = C.c_int
        L.bo_index_insert_reads.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64]
        L.bo_class.restype = C.c_uint
        L.bo_class.argtypes = [C.c_uint64, C.c_uint, f64p]
        L.bo_key.restype = C.c_uint64
        L.bo_key.argtypes = [C.c_uint64, C.c_uint, f64p]
        L.bo_mix_inv.restype = C.c_uint64
        L.bo_mix_inv.argtypes = [C.c_uint64, C.c_uint]
        L.bo_record_words.restype = C.c_uint
        L.bo_record_words.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        L.bo_records.restype = C.c_int64
        L.bo_records.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, u64p, u32p, u8p, u8p, C.c_uint64]
        L.bo_synth_reads.restype = None
        L.bo_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64, C.c_uint64, u8p]

    def index_insert_reads(self, h, flat, offs, threads: int = 1) -> None:
        self.lib.bo_index_insert_reads(h, flat, offs, len(offs) - 1)

    def class_many(self, xs: Iterable[int], m: int) -> np.ndarray:
        coef = self.coef_table(m)
        return np.array([self.lib.bo_class(int(x), m, coef) for x in xs], dtype=np.uint8)

    def key_many(self, xs: Iterable[int], m: int) -> np.ndarray:
        coef = self.coef_table(m)
        return np.array([self.lib.bo_key(int(x), m, coef) for x in xs], dtype=np.uint64)

    def mix_inv_many(self, xs: Iterable[int], m: int) -> np.ndarray:
        return np.array([self.lib.bo_mix_inv(int(x), m) for x in xs], dtype=np.uint64)

    def records(self, h, seq: str | bytes, k: int, m: int, b: int):
        """Super-k-mer records of one read in the GPU exchange format."""
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nw = int(self.lib.bo_record_words(k, m, b))
        cap = max(len(s) - k + 1, 1)
        c = np.zeros(cap * nw, np.uint64)
        bucket = np.zeros(cap, np.uint32)
        n = np.zeros(cap, np.uint8)
        idx0 = np.zeros(cap, np.uint8)
        got = self.lib.bo_records(h, s, len(s), c, bucket, n, idx0, cap)
        assert got >= 0
        return c[: got * nw].reshape(got, nw), bucket[:got], n[:got], idx0[:got]

    def synth_reads(self, genome_len: int, first: int, n: int, L: int = 150, seed_g: int = 1, seed_r: int = 2) -> np.ndarray:
        out = np.zeros(n * L, np.uint8)
        self.lib.bo_synth_reads(genome_len, first, n, L, seed_g, seed_r, out)
        return out.reshape(n, L)


class Ref(_Lib):
    prefix = "ref_"
    path = REF_SO

    def __init__(self):
        super().__init__()
        L = self.lib
        L.ref_index_insert_reads.restype = C.c_int
        L.ref_index_insert_reads.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_int]
        L.ref_class_many.argtypes = [u64p, C.c_uint64, C.c_uint, u8p]
        L.ref_key_many.argtypes = [u64p, C.c_uint64, C.c_uint, u64p]
        L.ref_mix_inv_many.argtypes = [u64p, C.c_uint64, C.c_uint, u64p]
        L.ref_canonized.restype = C.c_int
        L.ref_canonized.argtypes = [C.c_uint64, C.c_uint64, C.c_uint]
        L.ref_index_nb_skmers.restype = C.c_uint64
        L.ref_index_nb_skmers.argtypes = [C.c_void_p]

    def index_insert_reads(self, h, flat, offs, threads: int = 1) -> None:
        self.lib.ref_index_insert_reads(h, flat, offs, len(offs) - 1, threads)

    def class_many(self, xs, m: int) -> np.ndarray:
        x = np.ascontiguousarray(np.array(list(xs), dtype=np.uint64))
        out = np.zeros(len(x), np.uint8)
        self.lib.ref_class_many(x, len(x), m, out)
        return out

    def key_many(self, xs, m: int) -> np.ndarray:
        x = np.ascontiguousarray(np.array(list(xs), dtype=np.uint64))
        out = np.zeros(len(x), np.uint64)
        self.lib.ref_key_many(x, len(x), m, out)
        return out

    def mix_inv_many(self, xs, m: int) -> np.ndarray:
        x = np.ascontiguousarray(np.array(list(xs), dtype=np.uint64))
        out = np.zeros(len(x), np.uint64)
        self.lib.ref_mix_inv_many(x, len(x), m, out)
        return out
