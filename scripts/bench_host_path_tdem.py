"""Rate of the reference-signature serial loop on a time-domain sounding (two-moment SkyTEM, 45 gates): TdemDataPoint + Inference1D,
every forward / Jacobian a B = 1 launch (python scripts/bench_host_path_tdem.py [iterations]; PROFILE=1 for a profile)."""
import os, sys, time, cProfile, pstats
import numpy as np
from numpy.random import Generator, PCG64DXSM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from geobipy_amd import Inference1D
from test_tdem_object_api import _skytem_point, SKYTEM_OPTIONS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dp = _skytem_point(row=20, noise=3)
inf = Inference1D(prng=Generator(PCG64DXSM(11)), world=None, **dict(SKYTEM_OPTIONS, n_markov_chains=n))
inf.initialize(dp)
pr = cProfile.Profile() if os.environ.get("PROFILE") else None
t0 = time.perf_counter()
if pr: pr.enable()
inf.infer(n_iterations=n)
if pr: pr.disable()
dt = time.perf_counter() - t0
print(f"{n} iterations in {dt:.2f} s = {n / dt:.0f} iterations/s (the reference, per its documentation: 43 - 83 /s per core)")
if pr: pstats.Stats(pr).sort_stats("tottime").print_stats(22)
