"""Per kernel of a rocprofv3 kernel trace: while it runs, how much of the time is some OTHER stream's kernel running too, and which
(python scripts/trace_overlap.py <kernel_trace.csv>) -- shows which stages of concurrent sub-blocks really overlap."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rj_" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].split("::")[-1][:22]) for r in rows]
ev.sort()
ev = ev[len(ev) // 3:]
import bisect
starts = [e[0] for e in ev]
acc = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for i, (s, e, q, name) in enumerate(ev):
    other = collections.Counter()
    covered = []
    j = bisect.bisect_left(starts, s - 400000)
    while j < len(ev) and ev[j][0] < e:
        s2, e2, q2, n2 = ev[j]
        if q2 != q and e2 > s:
            lo, hi = max(s, s2), min(e, e2)
            other[n2] += hi - lo
            covered.append((lo, hi))
        j += 1
    covered.sort()
    tot, cur_lo, cur_hi = 0, None, None
    for lo, hi in covered:
        if cur_hi is None or lo > cur_hi:
            if cur_hi is not None: tot += cur_hi - cur_lo
            cur_lo, cur_hi = lo, hi
        else:
            cur_hi = max(cur_hi, hi)
    if cur_hi is not None: tot += cur_hi - cur_lo
    a = acc[name]
    a[0] += e - s; a[1] += tot; a[2].update(other)
for name, (dur, cov, other) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-24s runs %7.1f ms; another stream's kernel is running %5.1f %% of that time; with: %s" % (
        name, dur / 1e6, 100.0 * cov / dur, ", ".join("%s %.0f %%" % (n, 100.0 * t / dur) for n, t in other.most_common(4))))
