"""Throughput of the batched rjMCMC driver (BASELINE config 5 shape scaled down):
B Resolve soundings in lockstep, per-chain numpy generators (reference-compatible streams), batched GPU kernels."""
import os, sys, time
import numpy as np
from numpy.random import Generator, PCG64DXSM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem
from geobipy_amd.inference import BatchedInference
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "mcmc_detail.npz"))
s = FdemSystem.read(os.path.join(G, "resolve.stm"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(1)
data = np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None]
heights = rng.uniform(25.0, 40.0, B)
t0 = time.perf_counter()
bi = BatchedInference(s, heights, data, [Generator(PCG64DXSM(b)) for b in range(B)], **RESOLVE_OPTIONS)
t_init = time.perf_counter() - t0
for _ in range(3): bi.step()
t0 = time.perf_counter()
acc = 0
for _ in range(n_it): acc += bi.step().mean()
dt = time.perf_counter() - t0
print(f"B={B}: init {t_init:.2f} s; {1e3*dt/n_it:.1f} ms per lockstep iteration -> {B*n_it/dt:.0f} chain-iterations/s "
      f"(acceptance {acc/n_it:.2f}, mean k {np.mean([st.k for st in bi.states]):.2f}); reference: ~165 iterations/s per CPU core")
