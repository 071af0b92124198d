"""End-to-end demonstration of row f-2 on the reference's own synthetic wedge (tests/data_checks/resolve_glacial_clean.csv,
79 soundings): noisy data -> BatchedInference (all soundings in lockstep on one GPU) -> misfit and layer statistics."""
import os, sys, time
import numpy as np
from numpy.random import Generator, PCG64DXSM
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem
from geobipy_amd.inference import BatchedInference
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
exact = len(sys.argv) > 2 and sys.argv[2] == "exact"     # use the true Jacobian in the stochastic-Newton proposals
clean = np.loadtxt(os.path.join(G, "resolve_glacial_clean.csv"), delimiter=",", skiprows=1)[:, 6:]
rng = np.random.default_rng(0)
std = np.sqrt((0.05 * clean) ** 2 + 5.0 ** 2)
data = clean + rng.normal(size=clean.shape) * std          # FdemData.create_synthetic_data noise model
s = FdemSystem.read(os.path.join(G, "resolve.stm"))
B = data.shape[0]
bi = BatchedInference(s, np.full(B, 30.0), data, [Generator(PCG64DXSM(1000 + b)) for b in range(B)], exact_jacobian=exact, **RESOLVE_OPTIONS)
m0 = np.array([st.misfit for st in bi.states])
t0 = time.perf_counter(); acc = 0.0
for it in range(n_it):
    acc += bi.step().mean()
dt = time.perf_counter() - t0
m1 = np.array([st.misfit for st in bi.states]); k = np.array([st.k for st in bi.states])
print("Jacobian in proposals:", "exact derivative" if exact else "reference expression")
print(f"{B} soundings x {n_it} iterations in {dt:.1f} s ({B*n_it/dt:.0f} chain-iterations/s), acceptance {acc/n_it:.2f}")
print(f"misfit (12 channels): half-space start median {np.median(m0):.1f} -> after {n_it} iterations median {np.median(m1):.1f}, "
      f"90th pct {np.percentile(m1, 90):.1f}; layers: mean {k.mean():.2f}, max {k.max()}")
zw = np.linspace(50, 1, 79) / 10
top = np.array([st.edges[0] if st.k > 1 else np.nan for st in bi.states])
ok = ~np.isnan(top)
print(f"first interface vs true wedge depth (soundings with >= 2 layers: {ok.sum()}): median |error| {np.nanmedian(np.abs(top - zw)):.2f} m")
