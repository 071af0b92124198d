"""Per-kernel sums of a rocprofv3 --pmc pass: python scripts/pmc_sum.py <dir with p_counter_collection.csv> -> counters per kernel (summed over its launches, in 1e6) and per wave"""
import csv, sys, collections, glob, os
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = max(c.get("SQ_WAVES", 1.0), 1.0)
    print(f"{k:42s} launches {n[k]:5d} waves {w:10.0f} | per wave: " + "  ".join(f"{name} {v / w:.1f}" for name, v in sorted(c.items()) if name != "SQ_WAVES"))
