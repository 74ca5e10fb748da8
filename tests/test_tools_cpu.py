"""tools/make_traffic_json.py (no GPU): the traffic JSON that bench.py's roofline.traffic comes from is built from the NEWEST
counter pass only -- gpurun merges every call's gpurun_out/ into the local one, so an earlier collection's files lie beside the
latest, and a sum over both would be the average of two different kernels."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD = "Kernel_Name,Counter_Name,Counter_Value\n"


def _pass(d, pid, ctr, scan, insert, mtime):
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, f"{pid}_counter_collection.csv")
    with open(p, "w") as f:
        f.write(HEAD)
        f.write(f'"void k_scan2<0, 0, 63, 21>(BriskParams)",{ctr},{scan}\n')
        f.write(f'"void k_insert_fast<3u, 49u, 4u>(BriskParams)",{ctr},{insert}\n')
    os.utime(p, (mtime, mtime))


def test_only_the_newest_pass_counts(tmp_path):
    src = tmp_path / "prof_x"
    now = time.time()
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        _pass(str(src / sub / "runc"), 100, ctr, 1000.0, 2000.0, now - 3600)  # an earlier collection
        _pass(str(src / sub / "runc"), 200, ctr, 10.0, 20.0, now)             # the latest
    (src / "src.sha256").write_text("abc123\n")
    (src / "args.txt").write_text("--reads 5\n")
    dst = tmp_path / "out" / "x"
    os.makedirs(dst.parent)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), str(src), str(dst), "5", "63", "21", "14"])
    d = json.load(open(str(dst) + "_pmc_traffic.json"))
    assert d["kernel_source_sha256"] == "abc123" and d["workload"] == {"reads": 5, "k": 63, "m": 21, "b": 14}
    ks = d["kernels"]
    assert ks["k_scan2<0, 0, 63, 21>"] == {"launches": 1, "FETCH_SIZE": 10.0, "WRITE_SIZE": 10.0}
    assert ks["k_insert_fast<3u, 49u, 4u>"] == {"launches": 1, "FETCH_SIZE": 20.0, "WRITE_SIZE": 20.0}


def test_replay_tool_finds_a_planted_input_nucleotide():
    """tools/replay_fuzz_case.py (no GPU): the analysis the parity soak runs on a wrong index.  One nucleotide of one read is
    changed before the oracle counts -- what a corrupted upload looks like from the index's side -- and the tool must name
    that read, offset and nucleotide; a wrong COUNT (nothing of the input changed) must give no match."""
    import random
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle
    import replay_fuzz_case as R
    oracle.build(ref=False)
    O = oracle.Oracle()
    rng = random.Random(5)
    genome = "".join(rng.choice("ACGT") for _ in range(1500))
    reads = []
    for _ in range(120):
        p = rng.randrange(0, len(genome) - 200)
        s = genome[p:p + rng.randint(60, 200)]
        reads.append(s if rng.random() < 0.5 else s[::-1].translate(str.maketrans("ACGT", "TGCA")))
    k, m, b = 31, 11, 6
    ri, off = 37, 45
    new = "A" if reads[ri][off] != "A" else "C"
    mut = list(reads)
    mut[ri] = reads[ri][:off] + new + reads[ri][off + 1:]
    got = O.count(mut, k, m, b)
    matches, tried = R.explain_by_one_input_nt(O, reads, k, m, b, got)
    # (reads that cover the same place of the genome with the same k-mers are indistinguishable by the index: they all match)
    assert tried > 0 and dict(read=ri, read_len=len(reads[ri]), offset=off, was=reads[ri][off], became=new) in matches
    comp = str.maketrans("ACGT", "TGCA")
    for mt in matches:
        r = reads[mt["read"]]
        ctx = r[max(0, mt["offset"] - 8):mt["offset"] + 9]
        assert ctx in genome or ctx[::-1].translate(comp) in genome
    want = O.count(reads, k, m, b)
    w = want[0][0].split()
    bad_count = (sorted([" ".join(w[:2] + [str((int(w[2]) + 1) % 256)])] + want[0][1:]), want[1], want[2])
    assert R.explain_by_one_input_nt(O, reads, k, m, b, bad_count)[0] == []


def test_replay_tool_rebuilds_the_recorded_round2_case():
    """the failure round 2 recorded (profiles/r02_fuzz_soak_one_failure.txt) is rebuilt from its seed and case number alone"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay_fuzz_case as R
    for case, c in R.cases(102, "v1"):
        if case == 739:
            break
    assert (c["k"], c["m"], c["b"], c["pb"], len(c["reads"]), c["immediate"]) == (47, 13, 8, 4, 556, False)
    assert c["reads"][292][843] == "T" and len(c["reads"][292]) == 869
