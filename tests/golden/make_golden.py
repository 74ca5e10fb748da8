#!/usr/bin/env python3
"""Generate tests/golden/*.json[.gz] from the REAL reference (oracle/_ref).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

Every expected value written here is an output of the reference's own code
(Kmers.cpp, hashing.cpp, Decycling.cpp, buckets.hpp, SuperKmerLight.hpp)
driven through oracle/ref_harness.cpp.  The two FASTA files are the
reference's own data fixtures (data/test.fa, data/debug/test.fa), copied as
data.  Nothing here is reference source text.
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import oracle  # noqa: E402

REFROOT = "/root/reference"
hx = lambda v: f"{int(v):x}"


def special_reads(rng, k):
    """Reads that force ties, fake minimizers, expiry chains and strand flips."""
    L = 150
    acgt = "ACGT"
    rnd = lambda n: "".join(rng.choice(acgt) for _ in range(n))
    out = [
        "A" * L, "C" * L, "G" * L, "T" * L,
        "AC" * (L // 2), "ACG" * (L // 3), "ACGT" * 40,
        "A" * 70 + rnd(80), rnd(80) + "T" * 70, rnd(40) + "A" * 70 + rnd(40),
        (rnd(17) * 10)[:L], (rnd(33) * 6)[:L], rnd(k), rnd(k + 1), rnd(k + 2), rnd(2 * k),
        rnd(L).lower(),
    ]
    s = rnd(75)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    out.append(s + "".join(comp[c] for c in reversed(s)))  # reverse-palindrome
    out += [rnd(L) for _ in range(12)]
    out += [rnd(rng.randint(k, 400)) for _ in range(6)]
    return out


def units(R, rng):
    res = {}
    for m in (5, 11, 13, 15, 21, 31):
        M = (1 << (2 * m)) - 1
        xs = [0, 1, 2, 3, M, M - 1, M >> 1, 0xAAAAAAAAAAAAAAAA & M, 0x5555555555555555 & M]
        xs += [rng.getrandbits(2 * m) for _ in range(400)]
        # low-complexity m-mers: R(x) near 0, where eps matters
        for unit in ("A", "C", "G", "T", "AC", "AG", "AT", "CG", "CT", "GT", "ACG", "ACT", "AAC", "ACGT"):
            s = (unit * m)[:m]
            xs.append(oracle.str2kmer(s)[0])
        res[str(m)] = {
            "coef_hex": [float(c).hex() for c in R.coef_table(m)],
            "x": [hx(x) for x in xs],
            "class": [int(c) for c in R.class_many(xs, m)],
            "key": [hx(v) for v in R.key_many(xs, m)],
            "mix_inv_of_keylow": [hx(v) for v in R.mix_inv_many([int(v) & M for v in R.key_many(xs, m)], m)],
        }
    return res


def rc_vectors(R, rng):
    out = {"rcbc": [], "rcb": [], "canonized": []}
    for n in (1, 5, 11, 21, 31, 32):
        for _ in range(20):
            x = rng.getrandbits(2 * n)
            out["rcbc"].append([hx(x), n, hx(R.rcbc(x, n))])
    for n in (5, 21, 30, 31, 32, 33, 47, 62, 63):
        for _ in range(20):
            v = rng.getrandbits(2 * n)
            lo, hi = v & (2**64 - 1), v >> 64
            a, b = R.rcb(lo, hi, n)
            out["rcb"].append([hx(lo), hx(hi), n, hx(a), hx(b)])
            out["canonized"].append([hx(lo), hx(hi), n, int(R.lib.ref_canonized(lo, hi, n))])
    return out


def minimizer_vectors(R, rng):
    out = []
    for (K, m) in ((30, 11), (31, 11), (31, 15), (62, 21), (63, 21), (63, 31), (40, 13), (33, 11)):
        cases = [rng.getrandbits(2 * K) for _ in range(60)]
        cases += [0, (1 << (2 * K)) - 1]
        for unit in ("A", "AC", "ACG", "T", "GT", "CA"):
            cases.append((oracle.str2kmer((unit * K)[:K])[1] << 64) | oracle.str2kmer((unit * K)[:K])[0])
        # k-mers with a repeated m-mer (forces the `==` tie rules)
        for _ in range(20):
            mm = "".join(rng.choice("ACGT") for _ in range(m))
            s = list("".join(rng.choice("ACGT") for _ in range(K)))
            for st in rng.sample(range(0, K - m + 1), 2):
                s[st:st + m] = mm
            lo, hi = oracle.str2kmer("".join(s[:K]))
            cases.append((hi << 64) | lo)
        for v in cases:
            lo, hi = v & (2**64 - 1), v >> 64
            mini, pos, rev = R.get_minimizer(lo, hi, K, m)
            out.append([hx(lo), hx(hi), K, m, hx(mini), pos, rev])
    return out


def enum_vectors(R, rng):
    out = []
    for (k, m) in ((31, 11), (63, 21), (31, 15), (33, 11), (41, 21), (63, 31), (21, 7)):
        for s in special_reads(rng, k):
            if len(s) < k:
                continue
            ret, n, lo, hi, idx, mini = R.enumerate(s, k, m)
            out.append({
                "k": k, "m": m, "seq": s,
                "skm_ret": [hx(v) for v in ret], "skm_n": [int(v) for v in n],
                "lo": [hx(v) for v in lo], "hi": [hx(v) for v in hi],
                "idx": [int(v) for v in idx], "mini": [hx(v) for v in mini],
            })
    return out


def md5_lines(lines):
    # same text the survey hashed: "KMER idx=N count\n"
    txt = "".join("{} idx={} {}\n".format(*l.split()) for l in lines)
    return hashlib.md5(txt.encode()).hexdigest()


def multisets(R, O, rng):
    summary = []
    fastas = {"test.fa": os.path.join(REFROOT, "data", "test.fa"),
              "debug_test.fa": os.path.join(REFROOT, "data", "debug", "test.fa")}
    for name, path in fastas.items():
        shutil.copyfile(path, os.path.join(HERE, name))
        seqs = oracle.fasta_sequences(open(path).read())
        for (k, m, b) in ((31, 11, 4), (63, 21, 14), (31, 13, 12), (31, 15, 14), (63, 21, 9), (31, 11, 11), (31, 11, 10)):
            lines, nk, nb = R.count(seqs, k, m, b)
            summary.append({"input": name, "k": k, "m": m, "b": b, "nb_kmers": nk, "nb_buckets": nb,
                            "sum_counts": sum(int(l.split()[2]) for l in lines), "md5": md5_lines(lines)})
            if name == "test.fa" and (k, m, b) in ((31, 11, 4), (63, 21, 14)):
                with gzip.open(os.path.join(HERE, f"multiset_test_k{k}m{m}b{b}.txt.gz"), "wt") as f:
                    f.write("\n".join(lines) + "\n")
    # synthetic reads from the SURVEY 8(d) generator: 15x coverage, both strands
    for (n_reads, G, k, m, b) in ((400, 4000, 31, 11, 4), (400, 4000, 31, 11, 11), (400, 4000, 63, 21, 14), (2000, 20000, 63, 21, 9)):
        reads = O.synth_reads(G, 0, n_reads)
        seqs = [bytes(r) for r in reads]
        lines, nk, nb = R.count(seqs, k, m, b, threads=4)
        flat, offs = oracle.pack_reads(seqs)
        h = R.index_new(k, m, b)
        R.index_insert_reads(h, flat, offs)
        sums = R.index_query_reads(h, flat[: int(offs[50])], offs[:51])
        R.index_free(h)
        summary.append({"input": f"synth:G={G},n={n_reads},L=150,seed_g=1,seed_r=2", "k": k, "m": m, "b": b,
                        "nb_kmers": nk, "nb_buckets": nb, "sum_counts": sum(int(l.split()[2]) for l in lines),
                        "md5": md5_lines(lines), "query_sums_first50": [int(v) for v in sums],
                        "first_read": bytes(reads[0]).decode(), "read_399": bytes(reads[399]).decode()})
    # the k-mer of SURVEY F3: 31 x A is stored three times (idx 0,1,2)
    lines, nk, nb = R.count(["A" * 33], 31, 11, 4)
    summary.append({"input": "literal:" + "A" * 33, "k": 31, "m": 11, "b": 4, "nb_kmers": nk, "nb_buckets": nb,
                    "sum_counts": sum(int(l.split()[2]) for l in lines), "md5": md5_lines(lines), "lines": lines})
    return summary


def main():
    oracle.build()
    R, O = oracle.Ref(), oracle.Oracle()
    rng = random.Random(20250321)
    dump = lambda name, obj: json.dump(obj, gzip.open(os.path.join(HERE, name), "wt"), separators=(",", ":"))
    dump("units.json.gz", units(R, rng))
    dump("rc.json.gz", rc_vectors(R, rng))
    dump("get_minimizer.json.gz", minimizer_vectors(R, rng))
    dump("enumerator.json.gz", enum_vectors(R, rng))
    json.dump(multisets(R, O, rng), open(os.path.join(HERE, "multisets.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
