"""Randomised parity soak: random (k, m, b), read sets with repeats and low-complexity stretches, random batch splits, readers in
between; index and get against the oracle.  python tests/fuzz_parity.py SECONDS SEED   (test_randomised_parity_soak runs 25 s of it; 2,211 cases in 300 s on the MI355X: profiles/r02_fuzz.txt)"""
import collections, json, os, random, sys, time
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


class IndexDiffers(Exception):
    def __init__(self, got):
        super().__init__("index differs")
        self.got = got


def run_case(reads, k, m, b, pb, immediate, splits, peeks, env):
    """the case's calls on a fresh handle under `env`; returns what the fuzz compares"""
    old = {n: os.environ.get(n) for n in env}
    os.environ.update(env)
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
            return (sorted(oracle.multiset_lines(*ix.enumerate(), k)), st["nb_kmers"], st["nb_buckets"])
    finally:
        for n, v in old.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v


def diagnose(got, want, reads, k, m, b, pb, immediate, splits, peeks):
    """Everything the next natural occurrence needs to name its stage, in full (round 2's record kept four lines of each list):
    the whole difference; whether the same calls on a fresh handle give the wrong index again, with every stage hand-over
    checked (BRISK_VERIFY) and the host path of every batch on stderr (BRISK_TRACE), deferred and immediate; and which
    single-nucleotide changes of the INPUT reproduce the wrong index exactly (tools/replay_fuzz_case.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay_fuzz_case
    ca, cb = collections.Counter(got[0]), collections.Counter(want[0])
    out = dict(nb_kmers=(got[1], want[1]), nb_buckets=(got[2], want[2]), extra=sorted((ca - cb).elements()), missing=sorted((cb - ca).elements()))
    out["n_extra"], out["n_missing"] = len(out["extra"]), len(out["missing"])
    for name, imm in (("rerun_same", immediate), ("rerun_immediate", True)):
        try:
            again = run_case(reads, k, m, b, pb, imm, splits, peeks, {"BRISK_VERIFY": "1", "BRISK_TRACE": "1"})
            out[name] = "right index" if again == want else ("the SAME wrong index" if again == got else "another wrong index")
        except Exception as e2:  # a stage check fired: its message names the stage
            out[name] = "stage check: " + repr(e2)[:1500]
    try:
        matches, tried = replay_fuzz_case.explain_by_one_input_nt(O, reads, k, m, b, got)
        out["one_input_nt"] = dict(candidates_tried=tried, reproduce_exactly=matches,
                                   reading="a match: the whole read was scanned with that nucleotide wrong (input stage: upload, ASCII staging, pack, packed stream); "
                                           "none: the fault is downstream of the scan's input")
    except Exception as e3:
        out["one_input_nt"] = "analysis failed: " + repr(e3)[:500]
    return out


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
vrng = random.Random(seed ^ 0x5eed)  # its own sequence: the cases stay those of the same seed without it
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
    verify = vrng.random() < 0.5  # half of the cases with the stage checks on (BRISK_VERIFY=1: include/brisk_hip.h)
    if only >= 0 and case != only:
        if case > only:
            break
        continue
    os.environ["BRISK_VERIFY"] = "1" if verify else "0"
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
                raise IndexDiffers(got)
            assert np.array_equal(ix.get_reads(q), want_q), "get differs"
            ix.sync()
    except Exception as e:
        rec = dict(case=case, k=k, m=m, b=b, pb=pb, n=len(reads), seed=seed, immediate=immediate, verify=verify, error=repr(e)[:2000],
                   calls=[(n_, pk) for n_, pk in zip(splits[:step + 1], peeks[:step + 1])] if "step" in dir() else None)
        if isinstance(e, IndexDiffers):
            rec.update(diagnose(e.got, want, reads, k, m, b, pb, immediate, splits, peeks))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "fuzz_fail_seed%d_case%d.json" % (seed, case))
        json.dump(rec, open(path, "w"), indent=1)
        print("FAIL", {k_: v for k_, v in rec.items() if k_ not in ("extra", "missing", "calls")}, "full record:", path, flush=True)
        sys.exit(1)
    n_ok += 1
    if n_ok % 20 == 0:
        print("ok", n_ok, dict(k=k, m=m, b=b, pb=pb, n=len(reads)), flush=True)
print("done", n_ok, "cases")
