"""Rehearsal of the multi-GPU survey run on the hardware there is: the command line with N ranks over gloo sharing ONE
GPU against the one-rank run -- same data file, same options, posterior payload at size (hit maps, traces, all containers) -- and
file-for-file equality of everything written.  python scripts/rehearse_ranks.py [ranks 8] [soundings 8192] [lines 16] [n_markov_chains 2000]
Default schedule of the command line (auto -> whole lines per rank: every rank writes its own containers, only the summaries travel through
all_gather_into_tensor).  REHEARSE_SCHEDULE=dynamic ships the posterior rows point to point instead (gloo-tested only)."""
import os, shutil, socket, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n_mc = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
d = tempfile.mkdtemp()
src = os.path.join(G, "resolve_glacial_clean.csv")
hdr = open(src).readline().strip()
raw = np.loadtxt(src, delimiter=",", skiprows=1)
rng = np.random.default_rng(1)
rows = raw[rng.integers(0, raw.shape[0], N)].copy()
rows[:, 6:] *= 1.0 + 0.02 * rng.standard_normal((N, raw.shape[1] - 6))
rows[:, 0] = np.repeat(np.arange(n_lines), -(-N // n_lines))[:N] + 100.0
rows[:, 1] = np.arange(N)
np.savetxt(os.path.join(d, "survey.csv"), rows, delimiter=",", header=hdr, comments="")
shutil.copy(os.path.join(G, "resolve.stm"), d)
opt = open(os.path.join(G, "resolve_options_small")).read().replace("n_markov_chains = 6000", "n_markov_chains = %d" % n_mc)
open(os.path.join(d, "options"), "w").write(opt)
common = [os.path.join(d, "options"), None, "--data_directory", d, "--data_filename", "survey.csv"]
if os.environ.get("REHEARSE_SCHEDULE"):
    common += ["--schedule", os.environ["REHEARSE_SCHEDULE"]]
env = dict(os.environ, GBP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
times = {}
for tag, n in (("one", 1), ("many", ranks)):
    out = os.path.join(d, tag); os.makedirs(out)
    args = [a if a is not None else out for a in common]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = ([sys.executable, "-m", "geobipy_amd"] if n == 1 else
           [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
            "-m", "geobipy_amd"]) + args
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    times[tag] = time.perf_counter() - t0
    print(tag, "rc", r.returncode, "%.1f s" % times[tag], "|", " | ".join(l for l in r.stdout.splitlines() if "soundings" in l or "schedule" in l or "containers" in l)[:400], flush=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
from geobipy_amd import hdf
names = sorted(os.listdir(os.path.join(d, "one")))
assert names == sorted(os.listdir(os.path.join(d, "many"))), (names, sorted(os.listdir(os.path.join(d, "many"))))
n_arrays = n_bytes = 0
def walk(a, b, path):
    global n_arrays, n_bytes
    if isinstance(a, dict):
        assert sorted(a) == sorted(b), path
        for k in a:
            walk(a[k], b[k], path + "/" + k)
    else:
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and a.dtype == b.dtype and (np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b)), path
        n_arrays += 1; n_bytes += a.nbytes
for n in names:
    p1, p2 = os.path.join(d, "one", n), os.path.join(d, "many", n)
    if n.endswith(".h5"):
        (a, aa), (b, ba) = hdf.load_results(p1), hdf.load_results(p2)
        walk(dict(a), dict(b), n)
        assert sorted(aa) == sorted(ba), n + " (attributes)"
    elif n.endswith(".npz"):
        a, b = np.load(p1, allow_pickle=False), np.load(p2, allow_pickle=False)
        walk({k: a[k] for k in a.files}, {k: b[k] for k in b.files}, n)
    else:
        assert open(p1, "rb").read() == open(p2, "rb").read(), n
size = sum(os.path.getsize(os.path.join(d, "many", n)) for n in names)
print("REHEARSAL OK: %d ranks (gloo, one GPU) wrote the %d files (%.0f MB) the one-rank run writes: %d arrays, %.2f GB compared equal; "
      "%d soundings, %d lines, n_markov_chains %d; wall %.1f s with one rank, %.1f s with %d" % (
          ranks, len(names), size / 1e6, n_arrays, n_bytes / 1e9, N, n_lines, n_mc, times["one"], times["many"], ranks))
shutil.rmtree(d, ignore_errors=True)
