"""The dependency chain of one sub-block from a rocprofv3 kernel trace (python scripts/trace_chain.py <kernel_trace.csv> [skip_fraction]):
per queue, over the steady part of the run: mean duration of every kernel, mean gap between the end of a kernel and the start of the
next one on the same queue (by pair), and the iteration period (propose to propose)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rj_" in r["Kernel_Name"]]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
short = lambda n: n.split("(")[0].replace("rj::", "").replace("void ", "").split("<")[0]
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
for q, v in byq.items():
    v.sort()
    v = v[int(len(v) * skip):]
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    prop = [s for s, e, n in v if "propose_staged" in n or "propose_thread" in n or "step8" in n]
    for i, (s, e, n) in enumerate(v):
        if n == "k_rj_physics" and i > 0:                       # stage 0 follows a proposal (or the fused accept + proposal), stage 1 the Newton stage
            n = "k_rj_physics stage " + ("1" if "newton" in v[i - 1][2] else "0")
            v[i] = (s, e, n)
        dur[n].append(e - s)
        if i + 1 < len(v):
            gap[n + " -> " + v[i + 1][2]].append(v[i + 1][0] - e)
    period = (prop[-1] - prop[0]) / max(1, len(prop) - 1) / 1e3 if len(prop) > 1 else float("nan")
    print("queue %s: %d dispatches, iteration period %.1f us" % (q, len(v), period))
    tot_d = tot_g = 0.0
    for n, d in dur.items():
        per_it = sum(d) / max(1, len(prop)) / 1e3
        tot_d += per_it
        print("   %-28s mean %7.2f us x %.2f per iteration = %7.2f us" % (n, sum(d) / len(d) / 1e3, len(d) / max(1, len(prop)), per_it))
    for n, g in sorted(gap.items()):
        per_it = sum(g) / max(1, len(prop)) / 1e3
        tot_g += per_it
        print("   gap %-46s mean %7.2f us (min %6.2f) -> %6.2f us per iteration" % (n, sum(g) / len(g) / 1e3, min(g) / 1e3, per_it))
    print("   kernels %.1f us + gaps %.1f us per iteration" % (tot_d, tot_g))
