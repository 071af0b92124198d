"""Latency of the per-sounding (B = 1) interface: FdemDataPoint.forward / sensitivity / data_misfit+likelihood."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import FdemDataPoint, Model, RectilinearMesh1D
G = os.path.join(ROOT, "tests", "golden")
mod = Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, 5.0, 7.5, np.inf]), values=np.r_[1e-2, 1e-1, 0.03333333])
dp = FdemDataPoint(z=30.0, data=np.full(12, 100.0), system=os.path.join(G, "resolve.stm"))
dp.relative_error, dp.additive_error = 0.05, 5.0
for name, fn in [("forward", lambda: dp.forward(mod)), ("sensitivity", lambda: dp.sensitivity(mod)),
                 ("data_misfit", lambda: dp.data_misfit()), ("likelihood", lambda: dp.likelihood(log=True))]:
    for _ in range(20): fn()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n): fn()
    print(f"{name:12s} {1e6*(time.perf_counter()-t0)/n:8.1f} us per call")
