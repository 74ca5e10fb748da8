"""Rebuild one case of tests/fuzz_parity.py on the CPU (no GPU, no library call) and say, for a k-mer the GPU index got wrong,
which reads hold it, where the wrong nucleotide lies in the read and in the packed stream, which super-k-mer of the oracle's
enumerator it belongs to, and which host path every insert call of the case takes.
    python tools/replay_fuzz_case.py [--v1] SEED CASE [WRONG_KMER RIGHT_KMER [--observed NB_KMERS NB_BUCKETS LINE...]]
The random sequence is the fuzz's own (same draws in the same order), so the inputs are those of the failed run exactly.
--observed: what the failed run reported (entry count, bucket count, some of its "KMER idx count" lines); every read that holds
the right k-mer is then run through the oracle with that one nucleotide changed, and the runs that give exactly the observed
numbers and lines are marked -- next to how the k-mers that cover the nucleotide spread over the read's super-k-mers, which is
what a fault in ONE record (downstream of the scan's input) could have changed at most.
profiles/r03_fuzz_failure_root_cause.txt is this tool's output for the failure round 2 recorded."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle


def cases(seed, order="v2"):
    """order "v2": the committed fuzz (splits and peeks drawn up front); "v1": the fuzz as it ran when the round-2 failure was
    recorded (read sample, immediate flag, then one split and one peek draw per insert call)."""
    rng = random.Random(seed)

    def rand_reads():
        glen = rng.choice([300, 2000, 20000])
        genome = "".join(rng.choice("ACGT") for _ in range(glen))
        if rng.random() < 0.5:
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
        return out, genome

    case = -1
    while True:
        case += 1
        k = rng.randint(12, 63)
        m = rng.choice([x for x in range(5, min(k - 1, 31) + 1, 2)])
        b = rng.randint(1, min(m, 14))
        pb = rng.choice([0, 0, 0, rng.randint(1, min(2 * b, 20))])
        if 2 * (k - b) + 6 > 128:
            yield case, None
            continue
        reads, genome = rand_reads()
        if order == "v1":
            qsel = sorted(rng.sample(range(len(reads)), min(len(reads), 60)))
            immediate = rng.random() < 0.3
            splits, peeks, i = [], [], 0
            while i < len(reads):
                n = rng.choice([1, 7, 64, 300, len(reads)])
                splits.append(n); i += n
                peeks.append(rng.random() < 0.1)
            yield case, dict(k=k, m=m, b=b, pb=pb, reads=reads, genome=genome, immediate=immediate, splits=splits, peeks=peeks, qsel=qsel)
            continue
        immediate, splits, peeks = rng.random() < 0.3, [rng.choice([1, 7, 64, 300, 10**9]) for _ in range(4000)], [rng.random() < 0.1 for _ in range(4000)]
        qsel = sorted(rng.sample(range(len(reads)), min(len(reads), 60)))
        yield case, dict(k=k, m=m, b=b, pb=pb, reads=reads, genome=genome, immediate=immediate, splits=splits, peeks=peeks, qsel=qsel)


def explain_by_one_input_nt(O, reads, k, m, b, got, max_sims=600):
    """Which single-nucleotide changes of the INPUT reproduce a wrong index exactly?  `got` = (sorted multiset lines, nb_kmers,
    nb_buckets) as the fuzz compares them.  Every k-mer the wrong index holds and the right one does not is tried against the
    right index's k-mers at Hamming distance one; every read that holds such a neighbour gives a candidate (read, offset, nt), and
    every candidate is run through the oracle.  A match says the fault lies upstream of the scan (the whole read was scanned with
    that nucleotide wrong); no match among the candidates says it does not (a single record, an entry, a count).
    Returns (matches, candidates tried)."""
    import collections
    want = O.count(reads, k, m, b)
    cg, cw = collections.Counter(got[0]), collections.Counter(want[0])
    extra = [l.split()[0] for l in (cg - cw)]
    have = set(l.split()[0] for l in want[0])
    comp = str.maketrans("ACGT", "TGCA")
    cands = {}
    for e in extra:
        if e in have:
            continue  # a right k-mer under a changed count or idx: the k-mers that exist nowhere in the input lead to the nucleotide
        for j in range(k):
            for nt in "ACGT":
                if nt == e[j]:
                    continue
                r = e[:j] + nt + e[j + 1:]
                if r not in have:
                    continue
                for ri, rd in enumerate(reads):
                    for fwd, km in ((True, r), (False, r[::-1].translate(comp))):
                        p = rd.find(km)
                        while p >= 0:
                            off = p + (j if fwd else k - 1 - j)
                            new = e[j] if fwd else e[j].translate(comp)
                            cands[(ri, off, new)] = None
                            p = rd.find(km, p + 1)
    matches, tried = [], 0
    for (ri, off, new) in sorted(cands):
        if tried >= max_sims:
            break
        tried += 1
        mut = list(reads)
        mut[ri] = reads[ri][:off] + new + reads[ri][off + 1:]
        if O.count(mut, k, m, b) == tuple(got):
            matches.append(dict(read=ri, read_len=len(reads[ri]), offset=off, was=reads[ri][off], became=new))
    return matches, tried


def main():
    order = "v2"
    if "--v1" in sys.argv:
        sys.argv.remove("--v1"); order = "v1"
    seed, want_case = int(sys.argv[1]), int(sys.argv[2])
    for case, c in cases(seed, order):
        if case == want_case:
            break
    if c is None:
        print("case skipped by the fuzz (outside the envelope)")
        return
    k, m, b, reads = c["k"], c["m"], c["b"], c["reads"]
    print("case", want_case, "seed", seed, dict(k=k, m=m, b=b, pb=c["pb"], n=len(reads), glen=len(c["genome"]), immediate=c["immediate"]))
    # the calls of the case, as the fuzz issues them
    i = step = 0
    calls = []
    while i < len(reads):
        n = c["splits"][step % len(c["splits"])]
        batch = reads[i:i + n]
        calls.append((i, len(batch), sum(len(r) for r in batch), c["peeks"][step % len(c["peeks"])]))
        i += n
        step += 1
    print("insert calls: %d; (first read, reads, nts, stats() behind it):" % len(calls))
    for cl in calls:
        print("   ", cl)
    observed = None
    if "--observed" in sys.argv:
        at = sys.argv.index("--observed")
        observed = (int(sys.argv[at + 1]), int(sys.argv[at + 2]), sys.argv[at + 3:])
        del sys.argv[at:]
    if len(sys.argv) > 4:
        wrong, right = sys.argv[3], sys.argv[4]
        O = oracle.Oracle() if observed else None
        want = O.count(reads, k, m, b) if observed else None
        if observed:
            print("right index (oracle): nb_kmers %d nb_buckets %d; observed: nb_kmers %d nb_buckets %d" % (want[1], want[2], observed[0], observed[1]))
        comp = str.maketrans("ACGT", "TGCA")
        diff = [j for j in range(len(wrong)) if wrong[j] != right[j]]
        print("wrong vs right k-mer differ at k-mer offsets", diff, [(right[j], wrong[j]) for j in diff])
        offs = np.cumsum([0] + [len(r) for r in reads])
        for ri, r in enumerate(reads):
            for orient, km in (("fwd", right), ("rc", right[::-1].translate(comp))):
                p = r.find(km)
                while p >= 0:
                    d = diff[0] if orient == "fwd" else k - 1 - diff[0]
                    rp = p + d
                    # call the read belongs to, and its stream index inside that call's batch
                    ci = max(j for j, cl in enumerate(calls) if cl[0] <= ri)
                    q = int(offs[ri] - offs[calls[ci][0]]) + rp
                    line = "  read %d (len %d, call %d) holds the right k-mer %s at %d: flipped nt at read offset %d, nt %s; batch stream index %d (mod 16: %d, mod 32: %d, word %d)" % (
                        ri, len(r), ci, orient, p, rp, r[rp], q, q % 16, q % 32, q // 16)
                    if observed:
                        new_nt = wrong[diff[0]] if orient == "fwd" else wrong[diff[0]].translate(comp)
                        mut = list(reads)
                        mut[ri] = r[:rp] + new_nt + r[rp + 1:]
                        got = O.count(mut, k, m, b)
                        have = set(got[0])
                        ok = got[1] == observed[0] and got[2] == observed[1] and all(l in have for l in observed[2])
                        # the k-mers covering the nucleotide, per super-k-mer of the RIGHT read (what one wrong record could change)
                        ret, nn, lo, hi, idx, mini = O.enumerate(r, k, m)
                        per, t = [], 0
                        for j in range(len(nn)):
                            starts = []
                            for i in range(int(nn[j])):
                                kmj = oracle.kmer2str(int(lo[t + i]), int(hi[t + i]), k)
                                pj = r.find(kmj)
                                if pj < 0:
                                    pj = r.find(kmj[::-1].translate(comp))
                                starts.append(pj)
                            t += int(nn[j])
                            c = sum(1 for pj in starts if pj <= rp < pj + k)
                            if c:
                                per.append(c)
                        line += "\n      whole read scanned with %s there: nb_kmers %d nb_buckets %d, observed lines present: %s%s; k-mers covering it per super-k-mer of the right read: %s" % (
                            new_nt, got[1], got[2], all(l in have for l in observed[2]), "   <== EXACTLY the observed failure" if ok else "", per)
                    print(line)
                    p = r.find(km, p + 1)


if __name__ == "__main__":
    main()
