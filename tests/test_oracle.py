"""The CPU oracle (oracle/brisk_oracle.c) against the golden vectors produced by
the reference's own code (tests/golden/make_golden.py), and -- where the real
reference build oracle/_ref is present -- against the reference itself on fresh
random inputs.  Bit-exact everywhere: integer/byte work."""
import hashlib
import random

import numpy as np
import pytest

import oracle
from conftest import load_golden

X = lambda s: int(s, 16)


def md5_lines(lines):
    return hashlib.md5("".join("{} idx={} {}\n".format(*l.split()) for l in lines).encode()).hexdigest()


# ---- a2/a3: hash, decycling class, coefficient tables (F5) -----------------
@pytest.mark.parametrize("m", [5, 11, 13, 15, 21, 31])
def test_units_against_golden(O, m):
    g = load_golden("units.json.gz")[str(m)]
    coef = O.coef_table(m)
    # the GPU box's libm must regenerate the committed table bit for bit
    assert [float(c).hex() for c in coef] == g["coef_hex"]
    xs = [X(x) for x in g["x"]]
    assert list(O.class_many(xs, m)) == g["class"]
    keys = O.key_many(xs, m)
    assert [f"{int(v):x}" for v in keys] == g["key"]
    M = (1 << (2 * m)) - 1
    assert [f"{int(v):x}" for v in O.mix_inv_many([int(v) & M for v in keys], m)] == g["mix_inv_of_keylow"]
    # the mixer is a bijection on 2m bits: inverse(key & M) == x
    assert [int(v) for v in O.mix_inv_many([int(v) & M for v in keys], m)] == xs


# ---- a4/a5: reverse complements, including the broken 128-bit one (F4) -----
def test_rc_against_golden(O):
    g = load_golden("rc.json.gz")
    for x, n, want in g["rcbc"]:
        assert O.rcbc(X(x), n) == X(want)
    for lo, hi, n, wlo, whi in g["rcb"]:
        assert O.rcb(X(lo), X(hi), n) == (X(wlo), X(whi))


def test_rcbc_is_a_true_reverse_complement(O):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rng = random.Random(1)
    for n in (1, 7, 21, 32):
        s = "".join(rng.choice("ACGT") for _ in range(n))
        rc = "".join(comp[c] for c in reversed(s))
        assert O.rcbc(oracle.str2kmer(s)[0], n) == oracle.str2kmer(rc)[0]


# ---- a6: get_minimizer with its 64-bit truncation and tie rules (F2) -------
def test_get_minimizer_against_golden(O):
    for lo, hi, K, m, mini, pos, rev in load_golden("get_minimizer.json.gz"):
        assert O.get_minimizer(X(lo), X(hi), K, m) == (X(mini), pos, rev), (lo, hi, K, m)


# ---- a7: the enumerator stream ----------------------------------------------
def test_enumerator_against_golden(O):
    cases = load_golden("enumerator.json.gz")
    assert len(cases) > 200
    for c in cases:
        ret, n, lo, hi, idx, mini = O.enumerate(c["seq"], c["k"], c["m"])
        assert [f"{int(v):x}" for v in ret] == c["skm_ret"], c["seq"]
        assert [int(v) for v in n] == c["skm_n"]
        assert [f"{int(v):x}" for v in lo] == c["lo"]
        assert [f"{int(v):x}" for v in hi] == c["hi"]
        assert [int(v) for v in idx] == c["idx"]
        assert [f"{int(v):x}" for v in mini] == c["mini"]


def test_enumerator_invariant_idx_increases_and_kmers_extend(O):
    """SuperKmerLight.hpp:99-104 relies on this inside every returned vector."""
    rng = random.Random(7)
    for k, m in ((31, 11), (63, 21)):
        s = "".join(rng.choice("ACGT") for _ in range(300))
        ret, n, lo, hi, idx, _ = O.enumerate(s, k, m)
        assert int(n.sum()) == len(s) - k + 1
        p = 0
        for cnt in n:
            km = [(int(hi[p + i]) << 64) | int(lo[p + i]) for i in range(cnt)]
            for i in range(1, cnt):
                assert idx[p + i] == idx[p + i - 1] + 1
                assert km[i] >> 2 == km[i - 1] & ((1 << (2 * (k - 1))) - 1)
            p += cnt


# ---- a8-a12: index semantics, multisets --------------------------------------
def _seqs(name):
    return oracle.fasta_sequences(load_golden(name))


def test_fasta_segmentation_of_the_reference_fixture():
    # data/test.fa has one N (line 48): two segments
    assert [len(s) for s in _seqs("test.fa")] == [3285, 2944]
    assert [len(s) for s in _seqs("debug_test.fa")] == [27313]


def test_multisets_against_golden(O):
    for e in load_golden("multisets.json"):
        if e["input"].startswith("synth:"):
            kv = dict(p.split("=") for p in e["input"][6:].split(","))
            reads = O.synth_reads(int(kv["G"]), 0, int(kv["n"]))
            assert bytes(reads[0]).decode() == e["first_read"]
            assert bytes(reads[399]).decode() == e["read_399"]
            seqs = [bytes(r) for r in reads]
        elif e["input"].startswith("literal:"):
            seqs = [e["input"][8:]]
        else:
            seqs = _seqs(e["input"])
        lines, nk, nb = O.count(seqs, e["k"], e["m"], e["b"])
        assert (nk, nb) == (e["nb_kmers"], e["nb_buckets"]), e["input"]
        assert sum(int(l.split()[2]) for l in lines) == e["sum_counts"]
        assert md5_lines(lines) == e["md5"], (e["input"], e["k"], e["m"], e["b"])
        if "lines" in e:
            assert lines == e["lines"]
        if "query_sums_first50" in e:
            flat, offs = oracle.pack_reads(seqs)
            h = O.index_new(e["k"], e["m"], e["b"])
            O.index_insert_reads(h, flat, offs)
            sums = O.index_query_reads(h, flat[: int(offs[50])], offs[:51])
            O.index_free(h)
            assert [int(v) for v in sums] == e["query_sums_first50"]


def test_survey_appendix_c_numbers(O):
    """The survey's own capture from the full reference binary."""
    want = {("test.fa", 31, 11, 4): (6163, 6169, 221, "413dd23230e1"),
            ("test.fa", 63, 21, 14): (6105, 6105, 237, "b08ef37a73db"),
            ("debug_test.fa", 31, 11, 4): (27283, 27283, 256, "5a21dc9f0063"),
            ("debug_test.fa", 63, 21, 14): (27251, 27251, 1064, "866edbe7486f")}
    for (name, k, m, b), (nk, sc, nb, md5) in want.items():
        lines, got_nk, got_nb = O.count(_seqs(name), k, m, b)
        assert (got_nk, got_nb, sum(int(l.split()[2]) for l in lines)) == (nk, nb, sc)
        assert md5_lines(lines).startswith(md5)


def test_full_multiset_fixture(O):
    for k, m, b in ((31, 11, 4), (63, 21, 14)):
        want = load_golden(f"multiset_test_k{k}m{m}b{b}.txt.gz").split("\n")[:-1]
        lines, _, _ = O.count(_seqs("test.fa"), k, m, b)
        assert lines == want


def test_poly_a_is_stored_three_times(O):
    # SURVEY.md F3 / Appendix C
    # same k-mer, three entries: identity is (kmer_s, minimizer_idx), not the k-mer
    lines, nk, nb = O.count(["A" * 33], 31, 11, 4)
    want = [e for e in load_golden("multisets.json") if e["input"] == "literal:" + "A" * 33][0]["lines"]
    assert lines == want and nk == 3 and nb == 1
    assert {l.split()[0] for l in lines} == {"A" * 31} and len({l.split()[1] for l in lines}) == 3


def test_counts_wrap_mod_256(O):
    s = "ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGGCTAGCTAGCTAGGCTAG"
    lines, _, _ = O.count([s] * 300, 31, 11, 4)
    assert {int(l.split()[2]) for l in lines} == {300 % 256}


def test_parameter_contract(O):
    # F1: b > m is invalid in the reference (parameters.hpp:22,28 underflows); reject
    for k, m, b in ((31, 11, 14), (31, 12, 4), (31, 31, 4), (64, 21, 14), (31, 11, 0), (63, 33, 4)):
        with pytest.raises(ValueError):
            O.index_new(k, m, b)


def test_record_format_matches_kmer_semantics(O):
    """A super-k-mer record (what the GPU scan emits) expands back to exactly the
    compacted k-mers the reference stores (Kmers.cpp:138-145, SKL :301-312)."""
    rng = random.Random(3)
    for k, m, b in ((31, 11, 4), (31, 11, 11), (63, 21, 14), (63, 21, 9), (41, 13, 6)):
        h = O.index_new(k, m, b)
        s = "".join(rng.choice("ACGT") for _ in range(200))
        c, bucket, n, idx0 = O.records(h, s, k, m, b)
        ret, skm_n, lo, hi, idx, _ = O.enumerate(s, k, m)
        assert list(n) == list(skm_n)
        suff = (m - b + 1) // 2
        M = (1 << (2 * m)) - 1
        coef = O.coef_table(m)
        p = 0
        for r in range(len(n)):
            big = sum(int(w) << (64 * i) for i, w in enumerate(c[r]))
            assert idx0[r] == idx[p] + suff
            for j in range(n[r]):
                kmer = (int(hi[p + j]) << 64) | int(lo[p + j])
                i = int(idx[p + j])
                key = int(O.lib.bo_key((kmer >> (2 * i)) & M, m, coef))
                hk = (kmer & ~(M << (2 * i))) | ((key & M) << (2 * i))
                if j == 0:
                    assert bucket[r] == (key >> (2 * suff)) & ((1 << (2 * b)) - 1)
                clo, chi = O.compacted(hk & (2**64 - 1), hk >> 64, b, i + suff)
                want = (chi << 64) | clo
                got = (big >> (2 * (int(n[r]) - 1 - j))) & ((1 << (2 * (k - b))) - 1)
                assert got == want
            p += int(n[r])
        O.index_free(h)


# ---- oracle vs the real reference on fresh random inputs ---------------------
def test_against_reference_build_random_reads(O, R):
    rng = random.Random(99)
    for k, m, b in ((31, 11, 4), (63, 21, 9), (33, 11, 7), (47, 15, 10), (21, 7, 3), (63, 31, 12)):
        genome = "".join(rng.choice("ACGT") for _ in range(3000))
        reads = []
        for _ in range(300):
            p = rng.randrange(0, len(genome) - 150)
            s = genome[p:p + 150]
            if rng.random() < 0.5:
                s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
            reads.append(s)
        reads += ["A" * 150, "AC" * 75, "ACG" * 50, "T" * 149 + "A"]
        a = O.count(reads, k, m, b)
        r = R.count(reads, k, m, b, threads=2)
        assert a == r, (k, m, b)
        for s in reads[:40]:
            ea, er = O.enumerate(s, k, m), R.enumerate(s, k, m)
            for x, y in zip(ea, er):
                assert np.array_equal(x, y)


def test_against_reference_build_units(O, R):
    rng = random.Random(5)
    for m in (7, 11, 21, 31):
        xs = [rng.getrandbits(2 * m) for _ in range(3000)]
        assert np.array_equal(O.class_many(xs, m), R.class_many(xs, m))
        assert np.array_equal(O.key_many(xs, m), R.key_many(xs, m))
        assert np.array_equal(O.coef_table(m), R.coef_table(m))
