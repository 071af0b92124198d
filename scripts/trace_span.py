"""Span of the sampler's dispatches in a rocprofv3 kernel trace against the host's clock for the same iterations
(python scripts/trace_span.py <kernel_trace.csv>): first start / last end of the k_rj_* rows, iterations = propose launches / sub-blocks."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rj_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
by_q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in rows)
n_prop = sum(1 for r in rows if "propose" in r["Kernel_Name"])
print("sampler dispatches %d, propose launches %d, span %.3f ms, queues/streams %s" % (len(rows), n_prop, (t1 - t0) / 1e6, dict(by_q)))
# the same per queue: busy time (sum of durations) and span
for q in by_q:
    rq = [r for r in rows if (r.get("Queue_Id"), r.get("Stream_Id")) == q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rq)
    span = max(int(r["End_Timestamp"]) for r in rq) - int(rq[0]["Start_Timestamp"])
    print("  queue/stream %s: %d dispatches, busy %.3f ms of span %.3f ms" % (q, len(rq), busy / 1e6, span / 1e6))
