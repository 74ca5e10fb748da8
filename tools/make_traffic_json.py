"""profiles/<tag>_pmc_traffic.json and <tag>_kernel_stats.csv from what tools/collect_profiles.sh wrote:
python tools/make_traffic_json.py gpurun_out/prof_TAG profiles/TAG READS K M B"""
import collections, csv, glob, json, shutil, sys

src, dst, reads, k, m, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kern = collections.defaultdict(lambda: {"launches": 0})
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    seen = collections.Counter()
    for f in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            kern[name][ctr] = kern[name].get(ctr, 0.0) + float(r["Counter_Value"])
            seen[name] += 1
    for name, n in seen.items():
        kern[name]["launches"] = max(kern[name]["launches"], n)
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (two passes, tools/collect_profiles.sh)",
    "unit": "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them, summed over the launches of one job",
    "note": "gfx950: FETCH_SIZE under-reports reads (calibrated on k_scatter, whose reads are exactly one record per thread: see DESIGN.md section 4)",
    "workload": {"reads": reads, "k": k, "m": m, "b": b},
    "kernels": {n: v for n, v in sorted(kern.items()) if n.startswith("k_")},
}
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
for f in glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, dst + "_kernel_stats.csv")
print("wrote", dst + "_pmc_traffic.json")
