"""Sum rocprofv3 counter_collection CSVs per kernel name: python tools/sum_pmc.py DIR [DIR...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:40]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (name, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key); n[name] += 1
    print("==", d)
    for name in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("SQ_INSTS_SALU", 0))):
        print(f"{name:42s} launches={n[name]:3d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(tot[name].items())))
