"""Brisk::reallocate (reference brisk/Brisk.hpp:202-224: the index re-bucketed to (m + 2, b + 2)).  The
reference never runs it (call site commented out, :124-129) and its body calls an update_kmer overload that no file
defines, so there is nothing of the reference to compare bytes with: PARITY UNPINNED.  The semantics built here are the
path's own -- every entry's k-mer, as a sequence of k nts, goes through SuperKmerEnumerator at the new m and is inserted
with its count -- and THAT is checked bit-exactly against the oracle's enumerator, plus the size-independent property
the verdict asks for: counts summed per canonical k-mer are unchanged."""
import os
import random
import subprocess
from collections import Counter

import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _canon(s):
    rc = "".join(COMP[c] for c in reversed(s))
    return min(s, rc)


def _one_kmer(O, km, k, m2):
    """what SuperKmerEnumerator yields for a sequence of exactly k nts at minimizer size m2: [(kmer_s string, minimizer_idx)]"""
    _, _, lo, hi, idx, _ = O.enumerate(km, k, m2)
    return [(oracle.kmer2str(int(l), int(h), k), int(i)) for l, h, i in zip(lo, hi, idx)]


def _expected(O, old_lines, k, m2):
    """old entries ("KMER idx=N count") -> the multiset the re-bucketed index must hold"""
    acc = Counter()
    for line in old_lines:
        w = line.split()
        km, cnt = w[0], int(w[-1])
        got = _one_kmer(O, km, k, m2)
        assert len(got) == 1, (km, got)
        acc[got[0]] = (acc[got[0]] + cnt) % 256
    return acc


def _lines_to_counter(lines):
    c = Counter()
    for line in lines:
        w = line.split()
        c[(w[0], int(w[1].split("=")[-1]))] = int(w[-1])
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("k,m,b,pb", [(31, 11, 4, 0), (63, 21, 14, 0), (41, 15, 10, 0),
                                      # few partitions in the target: more than 128 one-k-mer records per partition, each with its entry's
                                      # count in the header -- k_insert_big's in-place collapse must add those counts up, not count the records
                                      (31, 11, 4, 4), (41, 15, 10, 3)])
def test_bulk_reallocate_matches_the_enumerator_at_the_new_m(O, k, m, b, pb):
    import brisk_amd
    rng = random.Random(77)
    genome = "".join(rng.choice("ACGT") for _ in range(6000))
    reads = []
    for _ in range(1500):
        p = rng.randrange(0, len(genome) - 150)
        s = genome[p:p + 150]
        reads.append(s if rng.random() < 0.5 else "".join(COMP[c] for c in reversed(s)))
    reads += ["A" * 120, "ACGT" * 40, "AC" * 70]  # low complexity: minimizer ties, entries that merge
    with brisk_amd.BriskHip(k, m, b) as old, brisk_amd.BriskHip(k, m + 2, b + 2, part_bits=pb) as new:
        old.insert_reads(reads)
        before = oracle.multiset_lines(*old.enumerate(), k)
        assert max(int(l.split()[-1]) for l in before) > 20  # counts well above 1, even and odd
        old.reallocate_into(new)
        after = oracle.multiset_lines(*new.enumerate(), k)
        assert oracle.multiset_lines(*old.enumerate(), k) == before, "the source index must stay as it is"
        st = new.stats()
    want = _expected(O, before, k, m + 2)
    got = _lines_to_counter(after)
    assert got == want
    assert st["nb_kmers"] == len(want)
    # counts summed per canonical k-mer (mod 256) are what they were
    def per_canon(lines):
        c = Counter()
        for line in lines:
            w = line.split()
            c[_canon(w[0])] = (c[_canon(w[0])] + int(w[-1])) % 256
        return c
    assert per_canon(after) == per_canon(before)


@pytest.mark.gpu
def test_facade_reallocate_carries_the_data_over(tmp_path, O):
    """Brisk<uint8_t>::reallocate() behind the reference's API: params read (k, m + 2, b + 2), every entry's DATA (the
    count kept on the host) arrives under its new identity"""
    import brisk_amd
    exe = os.path.join(ROOT, "brisk_amd", "apps", "brisk_count")
    if not os.path.exists(exe):
        brisk_amd.build_apps()
    fa = os.path.join(ROOT, "tests", "golden", "test.fa")
    k, m, b = 31, 11, 4
    d0, d1 = str(tmp_path / "before.txt"), str(tmp_path / "after.txt")
    r0 = subprocess.run([exe, "--facade", fa, str(k), str(m), str(b), d0], capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr[-1500:]
    r1 = subprocess.run([exe, "--facade", fa, str(k), str(m), str(b), d1], capture_output=True, text=True, timeout=900, env=dict(os.environ, BRISK_REALLOCATE="1"))
    assert r1.returncode == 0, r1.stderr[-1500:]
    assert "reallocated k 31 m 13 b 6" in r1.stdout
    before = open(d0).read().splitlines()
    after = open(d1).read().splitlines()
    assert len(before) == 6163
    # entries that merge keep the DATA of the last one written (`*value = *old_value`, brisk/Brisk.hpp:217) and the walk's
    # order is the index's own: a merged entry may hold the count of any of the entries that merged into it
    want = {}
    for line in before:
        w = line.split()
        got = _one_kmer(O, w[0], k, m + 2)
        assert len(got) == 1
        want.setdefault(got[0], set()).add(int(w[-1]))
    have = _lines_to_counter(after)
    assert set(have) == set(want)
    assert all(have[key] in want[key] for key in have)
    assert sum(len(v) > 1 for v in want.values()) < 20 and len(have) > 6000
