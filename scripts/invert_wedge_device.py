"""Row f-2 end to end on the reference's own synthetic wedge (tests/data_checks/resolve_glacial_clean.csv, 79 soundings),
device-resident sampler: noisy data -> DeviceChains (R independent chains per sounding, all in lockstep on one GPU)
-> misfit, layer statistics, recovered interface depth from the interface-depth posterior."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem, DeviceChains
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
exact = not (len(sys.argv) > 2 and sys.argv[2] == "reference")
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
clean = np.loadtxt(os.path.join(G, "resolve_glacial_clean.csv"), delimiter=",", skiprows=1)[:, 6:]
rng = np.random.default_rng(0)
std = np.sqrt((0.05 * clean) ** 2 + 5.0 ** 2)
data = clean + rng.normal(size=clean.shape) * std          # FdemData.create_synthetic_data noise model
s = FdemSystem.read(os.path.join(G, "resolve.stm"))
S = data.shape[0]
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
dc = DeviceChains(s, np.full(S * R, 30.0), np.repeat(data, R, axis=0), seed=7, exact_jacobian=exact, **o)
m0 = dc.misfit.cpu().numpy()
torch.cuda.synchronize(); t0 = time.perf_counter()
dc.run(n_it // 2, accumulate=False)                         # burn-in
dc.run(n_it - n_it // 2, accumulate=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
sm = dc.summaries().cpu().numpy()
print("Jacobian in proposals:", "exact derivative" if exact else "reference expression")
print(f"{S} soundings x {R} chains x {n_it} iterations in {dt:.2f} s ({S*R*n_it/dt/1e6:.2f} M chain-iterations/s), acceptance {sm[:,4].mean():.2f}")
print(f"misfit (12 channels): half-space start median {np.median(m0):.1f} -> final median {np.median(sm[:,0]):.1f}, 90th pct {np.percentile(sm[:,0], 90):.1f}; "
      f"layers: mean {sm[:,3].mean():.2f}, max {int(sm[:,3].max())}")
eh = dc.edge_hist.view(S, R, -1).sum(dim=1).cpu().numpy().astype(float)      # pooled interface-depth posterior per sounding
zc = (np.arange(eh.shape[1]) + 0.5) * dc.depth_bin_width
mode = zc[np.argmax(eh, axis=1)]
zw = np.linspace(50, 1, 79) / 10                                             # true depth of the wedge's base under each sounding
ok = eh.sum(axis=1) > 0
print(f"interface-depth posterior mode vs true wedge depth ({ok.sum()} soundings with interfaces): median |error| {np.median(np.abs(mode - zw)[ok]):.2f} m; "
      f"deep half (wedge > 2.5 m): {np.median(np.abs(mode - zw)[ok & (zw > 2.5)]):.2f} m")
