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
