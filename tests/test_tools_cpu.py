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
