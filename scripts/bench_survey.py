"""End-to-end time of a survey inversion (CSV -> chains on the device -> results containers): a synthetic Resolve survey of
N soundings built from the reference's wedge file (rows repeated with 2 % noise), the options of tests/golden/resolve_options_small.
python scripts/bench_survey.py [N] [n_markov_chains] [lines]     (SURVEY_CHUNK=n: the dynamic schedule in chunks of n soundings)"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import survey
G = os.path.join(ROOT, "tests", "golden")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 8
KIND = os.environ.get("SURVEY_KIND", "resolve")                               # resolve | skytem (two moments, 45 gates)
first = 6 if KIND == "resolve" else 15                                        # first data column of the file
src = os.path.join(G, KIND + "_glacial_clean.csv")
hdr = open(src).readline().strip()
raw = np.loadtxt(src, delimiter=",", skiprows=1)
rng = np.random.default_rng(1)
rows = raw[rng.integers(0, raw.shape[0], N)].copy()
rows[:, first:] *= 1.0 + 0.02 * rng.standard_normal((N, raw.shape[1] - first))
rows[:, 0] = np.repeat(np.arange(n_lines), -(-N // n_lines))[:N] + 100.0     # line numbers
rows[:, 1] = np.arange(N)                                                     # fiducials
d = tempfile.mkdtemp()
np.savetxt(os.path.join(d, "survey.csv"), rows, delimiter=",", header=hdr, comments="")
t0 = time.perf_counter()
if KIND == "resolve":
    ds = survey.FdemData.read_csv(os.path.join(d, "survey.csv"), os.path.join(G, "resolve.stm"))
else:
    ds = survey.TdemData.read_csv(os.path.join(d, "survey.csv"), survey.read_options(os.path.join(G, "skytem_options_small"))["system_filename"])
t1 = time.perf_counter()
import torch
res = survey.infer(os.path.join(G, KIND + "_options_small"), data=ds, n_markov_chains=n_mc, burn_in_min_iterations=n_mc // 4,
                   results_directory=os.path.join(d, "out"), output=os.path.join(d, "summary.npz"),
                   **(dict(schedule="dynamic", chunk=int(os.environ["SURVEY_CHUNK"])) if "SURVEY_CHUNK" in os.environ else {}))
torch.cuda.synchronize()
t2 = time.perf_counter()
import resource
peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6                # GB (Linux: KiB)
size = sum(os.path.getsize(os.path.join(d, "out", f)) for f in os.listdir(os.path.join(d, "out")))
print(f"{KIND} N={N} n_markov_chains={n_mc} lines={n_lines}: read csv {t1 - t0:.2f} s, infer + containers {t2 - t1:.2f} s "
      f"({N / (t2 - t1):.0f} soundings/s), containers {size / 1e6:.0f} MB in {len(os.listdir(os.path.join(d, 'out')))} files, peak host memory {peak:.1f} GB")
