"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/brisk_hip.h declares, and fails loudly (no CPU fallback) when no
gfx950 device is present.  No compute calls are made here."""
import os
import re

import pytest

import brisk_amd
from brisk_amd import hipapi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    brisk_amd.build_library()
    return hipapi.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "brisk_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(brisk_hip_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(hipapi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.brisk_hip_abi_version() == 4


def test_no_cpu_fallback_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(brisk_amd.BriskHipError) as e:
        brisk_amd.BriskHip(31, 11, 4)
    assert e.value.code == 6  # ENODEVICE


def test_product_never_touches_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "brisk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "brisk_oracle" not in txt and "libbrisk_ref" not in txt, f


def test_host_coef_table_matches_golden():
    from conftest import load_golden
    g = load_golden("units.json.gz")
    for m in (11, 21, 31):
        assert [float(c).hex() for c in brisk_amd.coef_table(m)] == g[str(m)]["coef_hex"]


def test_host_packer_is_nuc2int_in_bulk(lib):
    """The upload threads' ASCII -> 2-bit packer (host_pack, brisk_capi.hip: AVX2 with a scalar tail) against nuc2int
    (Kmers.cpp:442-444: (c >> 1) & 3) restated in numpy: every byte value, every length modulo 32 and 16, first
    nucleotide of a word in its top bits, the last word zero padded -- the layout k_pack_ascii writes on the device."""
    import ctypes as C
    import numpy as np
    f = lib.brisk_hip_debug_host_pack
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int]
    f.restype = C.c_int
    rng = np.random.default_rng(5)

    def want(b):
        v = (np.frombuffer(b, np.uint8) >> 1) & 3
        v = np.concatenate([v, np.zeros((-len(v)) % 16, np.uint8)]).reshape(-1, 16).astype(np.uint32)
        return (v << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)

    cases = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in list(range(0, 100)) + [1000, 4095, 4096, 4097, 65536 + 17]]
    cases += [bytes(rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), n)) for n in (31, 32, 33, 150, 100003)]
    for b in cases:
        for scalar in (0, 1):
            out = np.full((len(b) + 15) // 16 + 1, 0xDEADBEEF, np.uint32)
            assert f(b, len(b), out.ctypes.data, scalar) == 0
            assert out[-1] == 0xDEADBEEF                      # nothing written beyond the last word
            assert np.array_equal(out[:-1], want(b)), (len(b), scalar)
