"""KFF output (reference brisk/writer.hpp:75-179).  The reference writes through the un-vendored
kff-cpp-api and ships no KFF fixture, so the bytes cannot be compared with the reference's: PARITY UNPINNED.  What is
checked: the file follows the KFF 1.0 layout (tests/kff_reader.py, an independent reader), carries the sections
BriskWriter::write emits (two global-variable sections with the reference's variables, minimizer sections, the
reference's encoding byte and metadata), and round-trips to exactly the (k-mer, minimizer_idx, count) multiset of the
reference-generated goldens."""
import gzip
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kff_reader import read_kff  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _golden_lines(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return [l.split() for l in f.read().splitlines() if l.strip()]


def _triples(rows):
    out = []
    for w in rows:  # "KMER idx=N count" or "KMER N count"
        idx = int(w[1].split("=")[-1])
        out.append((w[0], idx, int(w[2])))
    return sorted(out)


def _check_file(path, k, m, want):
    hdr, ents = read_kff(path)
    assert hdr["encoding"] == "ACTG"  # code 0 A, 1 C, 2 T, 3 G: write_encoding(0, 1, 3, 2), brisk/writer.hpp:26
    assert hdr["metadata"].startswith("File generated with Brisk v1")
    assert hdr["sections"][:2] == ["v", "v"] and hdr["sections"][-1] == "v" and set(hdr["sections"][2:-1]) <= {"m"}
    assert hdr["vars"][0] == {"k": k, "data_size": 1, "max": 1}
    assert {key: hdr["vars"][1][key] for key in ("k", "m", "data_size", "max")} == {"k": k, "m": m, "data_size": 1, "max": 2 * (k - m)}
    got = sorted((km, idx, data[0]) for km, idx, data in ents)
    assert got == want
    return hdr


@pytest.mark.parametrize("name,k,m", [("multiset_test_k31m11b4.txt.gz", 31, 11), ("multiset_test_k63m21b14.txt.gz", 63, 21)])
def test_kff_emitter_round_trips_the_golden_multiset(tmp_path, name, k, m):
    """the emitter alone (no GPU): golden entries in, KFF out, independent reader back"""
    exe = str(tmp_path / "kff_unit")
    subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-I" + os.path.join(ROOT, "brisk_amd", "include"), os.path.join(ROOT, "tests", "cpp", "kff_unit.cpp"), "-o", exe])
    rows = _golden_lines(name)
    want = _triples(rows)
    text = "".join("%s %d %d\n" % t for t in want)
    out = str(tmp_path / "x.kff")
    subprocess.run([exe, out, str(k), str(m)], input=text, text=True, check=True)
    hdr = _check_file(out, k, m, want)
    # sorted input: one minimizer section per run of equal minimizers, so at least one and at most one per entry
    assert 1 <= hdr["sections"].count("m") <= len(want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--facade", "--bulk"])
def test_kff_of_the_gpu_index_equals_the_reference_multiset(tmp_path, mode):
    """BriskWriter (facade, entry-id DATA on the host) and brisk_write_kff (bulk counts on the device) on data/test.fa:
    the file's (k-mer, minimizer_idx, count) multiset == the reference's (BASELINE config #1, 6,163 entries)"""
    import brisk_amd
    exe = os.path.join(ROOT, "brisk_amd", "apps", "brisk_count")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    out = str(tmp_path / "index.kff")
    run = subprocess.run([exe, mode, os.path.join(GOLD, "test.fa"), "31", "11", "4", "-", out], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    want = _triples(_golden_lines("multiset_test_k31m11b4.txt.gz"))
    assert len(want) == 6163
    hdr = _check_file(out, 31, 11, want)
    assert hdr["sections"].count("m") >= 221  # at least one section per non-empty bucket (221 of 256)
