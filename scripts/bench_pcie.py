"""PCIe-inclusive rate: every step uploads a fresh conductivity set (65 536 x 8 fp64 = 4.2 MB, pinned host
memory) before the fused kernel and downloads (chi2, logL) (1 MB) after it.  Never the bench headline."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import FdemBatch, synthetic
s = synthetic.syn10_system()
B, L = 65536, 8
nl, sig, thk, h = synthetic.draw_models(B, L, seed=1)
clean = FdemBatch(s, nl, sig, thk, h).forward().cpu().numpy()
b = FdemBatch(s, nl, sig, thk, h, data=synthetic.noisy_observations(clean), relative_error=np.full(B, 0.05),
              additive_error=np.full(B, 5.0))
host_sets = [torch.from_numpy(synthetic.redraw_sigma(B, L, seed=10 + i)).pin_memory() for i in range(4)]
out_host = torch.empty((2, B), dtype=torch.float64).pin_memory()
def step(i, copy):
    if copy:
        b.sigma.copy_(host_sets[i % 4], non_blocking=True)
    c2, ll = b.forward_loglike(want_pred=False)
    if copy:
        out_host[0].copy_(c2, non_blocking=True); out_host[1].copy_(ll, non_blocking=True)
for copy in (False, True):
    for i in range(10): step(i, copy)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
    for i in range(n): step(i, copy)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'PCIe-inclusive' if copy else 'HBM-resident  '}: {1e3*dt/n:.3f} ms/step -> {B*n/dt/1e6:.2f} M evals/s")
