"""Where the survey's host time goes after the chains: python scripts/prof_survey_writer.py [N] [lines]
Wraps the container writer's entry points with clocks (per thread) and lists the members of one container by bytes."""
import os, sys, time, tempfile, threading, zipfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import survey, hdf, h5lite
G = os.path.join(ROOT, "tests", "golden")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 16
src = os.path.join(G, "resolve_glacial_clean.csv")
hdr = open(src).readline().strip()
raw = np.loadtxt(src, delimiter=",", skiprows=1)
rng = np.random.default_rng(1)
rows = raw[rng.integers(0, raw.shape[0], N)].copy()
rows[:, 6:] *= 1.0 + 0.02 * rng.standard_normal((N, raw.shape[1] - 6))
rows[:, 0] = np.repeat(np.arange(n_lines), -(-N // n_lines))[:N] + 100.0
rows[:, 1] = np.arange(N)
d = tempfile.mkdtemp()
np.savetxt(os.path.join(d, "survey.csv"), rows, delimiter=",", header=hdr, comments="")
ds = survey.FdemData.read_csv(os.path.join(d, "survey.csv"), os.path.join(G, "resolve.stm"))
log, T0 = [], [0.0]


def clocked(obj, name):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((name, threading.get_ident(), t - T0[0], time.perf_counter() - T0[0]))
    setattr(obj, name, wrap)


clocked(hdf.NpzGroup, "save"); clocked(hdf, "save_npz"); clocked(h5lite, "_deflated_rows"); clocked(hdf, "write_device_rows"); clocked(hdf, "create_inference1d")
import torch
for rep in range(2):
    log.clear()
    out = os.path.join(d, "out%d" % rep)
    tm = {}
    T0[0] = time.perf_counter()
    survey.infer(os.path.join(G, "resolve_options_small"), data=ds, n_markov_chains=2000, burn_in_min_iterations=500, results_directory=out,
                 output=os.path.join(d, "summary%d.npz" % rep), timings=tm)
    total = time.perf_counter() - T0[0]
print("total %.3f s" % total, {k: round(v, 3) for k, v in tm.items()})
threads = sorted({t for _, t, _, _ in log})
for name in ("create_inference1d", "write_device_rows", "save", "save_npz"):
    ev = [e for e in log if e[0] == name]
    if ev:
        print("%-20s n=%3d  busy %.3f s  first start %.3f  last end %.3f  longest %.3f  threads %d" % (
            name, len(ev), sum(e[3] - e[2] for e in ev), min(e[2] for e in ev), max(e[3] for e in ev), max(e[3] - e[2] for e in ev),
            len({e[1] for e in ev})))
for e in sorted(log, key=lambda e: e[2]):
    if e[0] in ("save", "save_npz"):
        print("   %-9s thread %2d  %.3f -> %.3f  (%.3f)" % (e[0], threads.index(e[1]), e[2], e[3], e[3] - e[2]))
print("container type", hdf.container_type(), "| files", sorted(os.listdir(out))[:3], "| total %.1f MB" % (sum(os.path.getsize(os.path.join(out, n)) for n in os.listdir(out)) / 1e6))
f = ([n for n in sorted(os.listdir(out)) if n.endswith(".npz")] or [None])[0]
with (zipfile.ZipFile(os.path.join(out, f)) if f else open(os.devnull)) as z:
  if f:
    inf = sorted(z.infolist(), key=lambda i: -i.file_size)
    print(f, "members", len(inf), "raw %.1f MB" % (sum(i.file_size for i in inf) / 1e6), "stored %.1f MB" % (sum(i.compress_size for i in inf) / 1e6))
    for i in inf[:14]:
        print("   %-60s raw %9.3f MB  file %8.3f MB  %s" % (i.filename[:60], i.file_size / 1e6, i.compress_size / 1e6, "stored" if i.compress_type == 0 else "deflate"))
with zipfile.ZipFile(os.path.join(d, "summary1.npz")) as z:
    inf = sorted(z.infolist(), key=lambda i: -i.file_size)
    print("summary: raw %.1f MB file %.1f MB" % (sum(i.file_size for i in inf) / 1e6, sum(i.compress_size for i in inf) / 1e6))
    for i in inf[:8]:
        print("   %-40s raw %9.3f MB  file %8.3f MB  %s" % (i.filename[:40], i.file_size / 1e6, i.compress_size / 1e6, "stored" if i.compress_type == 0 else "deflate"))
print("usable cores", survey._usable_cores())
