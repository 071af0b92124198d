"""Where one lockstep iteration of DeviceChains spends its time (torch profiler, top ops)."""
import os, sys
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
rng = np.random.default_rng(1)
dc = DeviceChains(s, rng.uniform(25.0, 40.0, B), np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None], seed=1,
                  exact_jacobian=True, **o)
dc.run(200); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    dc.run(20); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=60))
