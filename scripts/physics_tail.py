"""Durations of the sampler's physics launches against the share of deep (> 8 layers) chains: run under rocprofv3 --kernel-trace,
then  python scripts/physics_tail.py <kernel_trace.csv>  prints percentiles of k_rj_physics per stage; without an argument it runs
the workload (config 5's block, 8 192 chains, 2 000 iterations) and prints the layer-count distribution at the end."""
import sys, os
if len(sys.argv) > 1:
    import csv, numpy as np
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rj_physics" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]) / 1e3
    for name, part in (("first 10 % of the run", d[: len(d) // 10]), ("last 50 %", d[len(d) // 2:])):
        print(name, "launches", part.size, "percentiles 5 25 50 75 95 99 (us):", np.round(np.percentile(part, [5, 25, 50, 75, 95, 99]), 1))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
sys.argv = ["x", "rjmcmc_8192"]
src = open(os.path.join(ROOT, "scripts", "prof_case.py")).read().replace("warm, n_it, calls = (50, 300, 1)", "warm, n_it, calls = (50, 2000, 1)")
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "scripts", "prof_case.py")}
exec(compile(src, "prof_case.py", "exec"), g)
dc = g["dc"]
k = dc.k.cpu().numpy()
print("layer counts at the end: mean %.2f, share > 8: %.4f, > 4: %.3f, max %d" % (k.mean(), (k > 8).mean(), (k > 4).mean(), k.max()))
