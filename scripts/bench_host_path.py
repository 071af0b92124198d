"""Rate of the reference-signature serial loop on the GPU objects: Inference1D(prng, world, **options).initialize(datapoint);
.infer(hdf) -- one sounding, every forward / Jacobian / likelihood a B = 1 launch (python scripts/bench_host_path.py [n_markov_chains])."""
import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from geobipy_amd import Inference1D, FdemDataPoint, FdemSystem
from test_object_api import RESOLVE_OPTIONS, GOLDEN, generator_at
d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
n_mc = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dp = FdemDataPoint(x=0.0, y=0.0, z=float(d["z"]), elevation=0.0, data=d["data"], std=None,
                   system=FdemSystem.read(os.path.join(GOLDEN, "resolve.stm")), lineNumber=0.0, fiducial=30.0)
options = dict(RESOLVE_OPTIONS, n_markov_chains=n_mc, save_hdf5=False, interactive_plot=False, covariance_scaling=None, parameter_limits=None,
               seed=1, data_type="FdemData", system_filename="resolve.stm")
inf = Inference1D(prng=generator_at(d["rng_state"][0]), world=None, **options)
inf.initialize(dp)
t0 = time.perf_counter()
pr = cProfile.Profile() if os.environ.get("PROFILE") else None
if pr: pr.enable()
inf.infer(hdf_file_handle=None)
if pr: pr.disable()
dt = time.perf_counter() - t0
print(f"{inf.iteration} iterations in {dt:.2f} s = {inf.iteration / dt:.0f} iterations/s (the reference: 165 /s per core)")
if pr: pstats.Stats(pr).sort_stats("tottime").print_stats(25)
