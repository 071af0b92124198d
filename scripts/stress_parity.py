"""Randomised wide-range parity sweep of the GPU forward/likelihood/Jacobian against the C oracle (GPU box).
sigma 1e-6..1e2 S/m, thickness 0.01..1000 m, altitude 0.5..500 m, 1..30 layers, all three golden systems."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import FdemBatch, FdemSystem
from oracle import fdem_oracle as fo
G = os.path.join(ROOT, "tests", "golden")
rng = np.random.default_rng(2026)
B, Lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 60000, 30
for name in ["syn10", "resolve", "mixed"]:
    s, o = FdemSystem.read(os.path.join(G, name + ".stm")), fo.OracleSystem.read(os.path.join(G, name + ".stm"))
    nl = rng.integers(1, Lmax + 1, size=B).astype(np.int32)
    sig = np.exp(rng.uniform(np.log(1e-6), np.log(1e2), (B, Lmax)))
    thk = np.exp(rng.uniform(np.log(1e-2), np.log(1e3), (B, Lmax)))
    h = np.exp(rng.uniform(np.log(0.5), np.log(500.0), B))
    obs = np.abs(rng.normal(size=(B, 2 * s.nFrequencies))) * 100
    b = FdemBatch(s, nl, sig, thk, h, data=obs, relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
    c2, ll = b.forward_loglike(); torch.cuda.synchronize()
    p = b.predicted.cpu().numpy()
    t0 = time.time()
    pr, cr, lr = fo.forward_loglike_batch(o, nl, sig, thk, h, obs, np.full(B, 0.05), np.full(B, 5.0), nthreads=0)
    dt = time.time() - t0
    err = np.abs(p - pr); tol = 1e-7 + 1e-9 * np.abs(pr)
    bad = np.flatnonzero((err > tol).any(axis=1) | ~np.isfinite(p).all(axis=1))
    print(f"{name}: B={B} oracle {dt:.1f}s  max err/tol {np.nanmax(err / tol):.3f}  non-finite gpu {int((~np.isfinite(p)).sum())} oracle {int((~np.isfinite(pr)).sum())}  violations {bad.size}")
    for i in bad[:5]:
        k = int(np.argmax(err[i] / tol[i]))
        print("   sounding", i, "L", nl[i], "h %.2f" % h[i], "chan", k, "gpu", p[i, k], "oracle", pr[i, k], "err/tol %.2f" % (err[i, k] / tol[i, k]))
    ce = np.abs(c2.cpu().numpy() - cr); print("   chi2 max err/(1e-6+1e-9|ref|)", np.nanmax(ce / (1e-6 + 1e-9 * np.abs(cr))))
    # Jacobian on a subset
    idx = rng.choice(B, 300, replace=False)
    J = FdemBatch(s, nl[idx], sig[idx], thk[idx], h[idx]).sensitivity().cpu().numpy()
    worst = 0.0
    for q, i in enumerate(idx):
        L = nl[i]
        Jo = fo.sensitivity(o, sig[i, :L], thk[i, :L], h[i]); Jo = np.vstack([Jo.real, Jo.imag])
        worst = max(worst, np.nanmax(np.abs(J[q, :, :L] - Jo) / (1e-7 + 1e-9 * np.abs(Jo))))
    print("   jacobian max err/tol (300 soundings)", worst)
