"""Randomised parity soak: random (k, m, b), read sets with repeats and low-complexity stretches, random batch splits, readers in
between; index and get against the oracle.  python tests/fuzz_parity.py SECONDS SEED   (test_randomised_parity_soak runs 25 s of it; 2,211 cases in 300 s on the MI355X: profiles/r02_fuzz.txt)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import brisk_amd, oracle
oracle.build(ref=False)
O = oracle.Oracle()
budget, seed = float(sys.argv[1]), int(sys.argv[2])
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # replay: run just this case of the seed's sequence
rng = random.Random(seed)
t_end = time.time() + budget
n_ok = 0

def rand_reads():
    glen = rng.choice([300, 2000, 20000])
    genome = "".join(rng.choice("ACGT") for _ in range(glen))
    if rng.random() < 0.5:  # plant repeats / homopolymers
        for _ in range(rng.randint(1, 6)):
            p = rng.randrange(0, glen - 60)
            unit = rng.choice(["A", "C", "AC", "ACG", "ACGT", "T", "AAAAC", "GT"])
            ln = rng.randint(20, min(200, glen - p))
            genome = genome[:p] + (unit * ln)[:ln] + genome[p + ln:]
    out = []
    for _ in range(rng.randint(1, 1200)):
        L = rng.choice([150, 150, 150, rng.randint(1, 400), rng.randint(60, 2000)])
        L = min(L, glen)
        p = rng.randrange(0, glen - L + 1)
        s = genome[p:p + L]
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        out.append(s)
    if rng.random() < 0.3:
        out += ["A" * rng.randint(1, 300), "T" * 200, "AC" * 100]
    return out

case = -1
while time.time() < t_end:
    case += 1
    k = rng.randint(12, 63)
    m = rng.choice([x for x in range(5, min(k - 1, 31) + 1, 2)])
    b = rng.randint(1, min(m, 14))
    pb = rng.choice([0, 0, 0, rng.randint(1, min(2 * b, 20))])
    if 2 * (k - b) + 6 > 128:
        continue  # outside the library's envelope (EUNSUPPORTED): the entry key [compacted k-mer | idx'] must fit 128 bits
    reads = rand_reads()
    immediate, splits, peeks = rng.random() < 0.3, [rng.choice([1, 7, 64, 300, 10**9]) for _ in range(4000)], [rng.random() < 0.1 for _ in range(4000)]
    qsel = sorted(rng.sample(range(len(reads)), min(len(reads), 60)))
    if only >= 0 and case != only:
        if case > only:
            break
        continue
    want = O.count(reads, k, m, b)
    q = [reads[i] for i in qsel] + ["A" * 150]
    flat, offs = oracle.pack_reads(reads)
    h = O.index_new(k, m, b); O.index_insert_reads(h, flat, offs)
    qf, qo = oracle.pack_reads(q); want_q = O.index_query_reads(h, qf, qo); O.index_free(h)
    try:
        with brisk_amd.BriskHip(k, m, b, part_bits=pb, immediate_inserts=immediate) as ix:
            i = step = 0
            while i < len(reads):
                n = splits[step % len(splits)]
                ix.insert_reads(reads[i:i + n]); i += n
                if peeks[step % len(peeks)]:
                    ix.stats()
                step += 1
            st = ix.stats()
            got = (sorted(oracle.multiset_lines(*ix.enumerate(), k)), st["nb_kmers"], st["nb_buckets"])
            if got != want:
                import collections
                ca, cb = collections.Counter(got[0]), collections.Counter(want[0])
                extra, missing = list((ca - cb).items()), list((cb - ca).items())
                wk = collections.defaultdict(list)
                for line in want[0]:
                    w_ = line.split()
                    wk[w_[0]].append((w_[1], w_[2]))
                comp = str.maketrans("ACGT", "TGCA")
                notes = []
                for line, _ in extra[:40]:
                    km, idx, cnt = line.split()
                    where = [ri for ri, r in enumerate(reads) if km in r.upper() or km[::-1].translate(comp) in r.upper()][:4]
                    notes.append((km[:12] + "..", idx, cnt, "oracle has this k-mer as", wk.get(km), "reads", where))
                again = sorted(oracle.multiset_lines(*ix.enumerate(), k))
                raise AssertionError("index differs: nb_kmers %d/%d nb_buckets %d/%d; %d extra, %d missing; enumerate stable: %s; immediate=%s splits=%s\n  extra: %s\n  missing: %s" % (
                    got[1], want[1], got[2], want[2], len(extra), len(missing), again == got[0], immediate, splits[:8], notes, missing[:40]))
            assert np.array_equal(ix.get_reads(q), want_q), "get differs"
            ix.sync()
    except Exception as e:
        print("FAIL", dict(case=case, k=k, m=m, b=b, pb=pb, n=len(reads), seed=seed, immediate=immediate), repr(e)[:6000], flush=True)
        sys.exit(1)
    n_ok += 1
    if n_ok % 20 == 0:
        print("ok", n_ok, dict(k=k, m=m, b=b, pb=pb, n=len(reads)), flush=True)
print("done", n_ok, "cases")
