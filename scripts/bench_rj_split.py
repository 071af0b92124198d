"""Experiment: B chains as ONE lock-step block vs P sub-blocks advanced concurrently on P streams (one host thread each).
python scripts/bench_rj_split.py [B] [iterations]"""
import os, sys, time, threading
import numpy as np, torch
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
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 500
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
rng = np.random.default_rng(1)
data = np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None]
heights = rng.uniform(25.0, 40.0, B)
for P in (1, 2, 4):
    n = B // P
    streams = [torch.cuda.Stream() for _ in range(P)]
    dcs = []
    for p in range(P):
        with torch.cuda.stream(streams[p]):
            dc = DeviceChains(s, heights[p * n:(p + 1) * n], data[p * n:(p + 1) * n], seed=1, exact_jacobian=True, forward_waves=2,
                              first_chain=p * n, **o)
            dc.run_mode = 1
            dcs.append(dc)
    def work(p, k):
        with torch.cuda.stream(streams[p]):
            dcs[p].run(k)
    def go(k):
        th = [threading.Thread(target=work, args=(p, k)) for p in range(P)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    go(50)
    t0 = time.perf_counter(); go(n_it); dt = time.perf_counter() - t0
    ks = torch.cat([dc.k for dc in dcs]).double().mean().item()
    print(f"B={B} as {P} block(s) on {P} stream(s): {1e3 * dt / n_it:.3f} ms per iteration -> {B * n_it / dt / 1e6:.2f} M chain-it/s (mean k {ks:.3f})", flush=True)
