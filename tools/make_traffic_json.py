"""profiles/<tag>_pmc_traffic.json and <tag>_kernel_stats.csv from what tools/collect_profiles.sh wrote:
python tools/make_traffic_json.py gpurun_out/prof_TAG profiles/TAG READS K M B

The JSON carries the sha256 of the kernel sources the passes were taken on (src.sha256, written on the GPU box next to
the counters) and the list of kernels seen: bench.py reports roofline.traffic from it only while its own sources hash
to the same value."""
import collections, csv, glob, json, os, shutil, sys

src, dst, reads, k, m, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kern = collections.defaultdict(lambda: {"launches": 0})
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    seen = collections.Counter()
    # gpurun merges every call's gpurun_out/ into the local one: an earlier collection's files lie beside the latest.  Only the
    # newest pass counts (a sum over two generations would be an average of two different kernels).
    files = sorted(glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            kern[name][ctr] = kern[name].get(ctr, 0.0) + float(r["Counter_Value"])
            seen[name] += 1
    for name, n in seen.items():
        kern[name]["launches"] = max(kern[name]["launches"], n)
sha = open(os.path.join(src, "src.sha256")).read().strip() if os.path.exists(os.path.join(src, "src.sha256")) else None
args = open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else ""
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline %s (two passes, tools/collect_profiles.sh)" % args,
    "unit": "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them, summed over the launches of one job",
    "correction": {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0,
                   "why": "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (16 B per lane), WRITE_SIZE "
                          "the bytes exactly for 16-B-per-lane stores; k_insert and k_scan2 read records / packed reads and write entries / records with 16-B accesses. "
                          "Narrower accesses are uncalibrated in the guide (this repo's own calibration on k_scatter's 8-B-per-lane reads: x1/0.70)."},
    "kernel_source_sha256": sha,
    "workload": {"reads": reads, "k": k, "m": m, "b": b},
    "kernels": {n: v for n, v in sorted(kern.items()) if n.startswith("k_")},
}
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
for f in sorted(glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1:]:
    shutil.copy(f, dst + "_kernel_stats.csv")
if os.path.exists(os.path.join(src, "bench_under_trace.json")):
    shutil.copy(os.path.join(src, "bench_under_trace.json"), dst + "_bench_under_trace.json")
print("wrote", dst + "_pmc_traffic.json", "sources", sha)
