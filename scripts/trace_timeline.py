"""Per-iteration timeline of the lock-step sampler from a rocprofv3 kernel trace CSV: per kernel the mean duration, and the
mean idle gap between consecutive dispatches on the device (python scripts/trace_timeline.py <kt_kernel_trace.csv> <iterations>)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_it = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                      # drop set-up and warm-up
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
dur = collections.defaultdict(list)
busy, last_end, gap = 0, None, 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[r["Kernel_Name"].split("(")[0][-60:]].append(e - s)
    if last_end is not None and s > last_end:
        gap += s - last_end
    last_end = e if last_end is None else max(last_end, e)
print("window %.3f ms, %d dispatches, idle (no kernel running) %.1f %%" % ((t1 - t0) / 1e6, len(rows), 100.0 * gap / (t1 - t0)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s n=%6d mean %8.1f us  share of window %5.1f %%" % (k, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / (t1 - t0)))
