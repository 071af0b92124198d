"""Throughput of the device-resident rjMCMC (geobipy_amd/rjmcmc_gpu.py -> gbp_rj_run;
SURVEY row f-2, BASELINE config 5 shape): B Resolve soundings, all chains advancing in lockstep, ten launches per iteration,
no host synchronisation between iterations."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem
from geobipy_amd.rjmcmc_gpu import DeviceChains
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "mcmc_detail.npz"))
s = FdemSystem.read(os.path.join(G, "resolve.stm"))
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 200
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,8192,65536").split(",")]:
    rng = np.random.default_rng(1)
    data = np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None]
    heights = rng.uniform(25.0, 40.0, B)
    t0 = time.perf_counter()
    dc = DeviceChains(s, heights, data, seed=1, exact_jacobian=True, forward_waves=int(os.environ.get('RJ_WAVES', 2)),
                      hankel_eps_ppm=(float(os.environ['RJ_EPS']) if 'RJ_EPS' in os.environ else None), **o)
    torch.cuda.synchronize(); t_init = time.perf_counter() - t0
    dc.run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); dc.run(n_it); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sm = dc.summaries().cpu().numpy()
    print(f"B={B}: init {t_init:.2f} s; {1e3*dt/n_it:.2f} ms per lockstep iteration -> {B*n_it/dt:.0f} chain-iterations/s "
          f"(acceptance {sm[:,4].mean():.2f}, mean k {sm[:,3].mean():.2f}, median misfit {np.median(sm[:,0]):.1f}); "
          f"reference: ~165 iterations/s per CPU core", flush=True)
