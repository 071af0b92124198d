"""Long-run check of the device sampler: many chains under the reference's schedule with its default 100 000 iterations
(Resolve fixture sounding scaled per chain; syn10 synthetic survey), then every invariant that must hold at the end:
finite state, structural constraints, cached prediction / likelihood = from-scratch evaluation, posterior counts."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem, FdemBatch, DeviceChains, synthetic, _lib
from geobipy_amd import rjmcmc_gpu as rg
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
o = dict(RESOLVE_OPTIONS, n_markov_chains=n_mc)
d = np.load(os.path.join(G, "mcmc_detail.npz"))
rng = np.random.default_rng(1)
cases = []
s = FdemSystem.read(os.path.join(G, "resolve.stm"))
cases.append(("resolve, reference Jacobian", s, rng.uniform(25, 40, B), np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None], False))
s10 = synthetic.syn10_system()
nl, sig, thk, h = synthetic.draw_models(B, 4, seed=11)
clean = FdemBatch(s10, nl, sig, thk, h).forward().cpu().numpy()
noisy = clean + rng.normal(size=clean.shape) * np.sqrt((0.05 * clean) ** 2 + 25.0)
cases.append(("syn10, exact Jacobian", s10, h, noisy, True))
for name, system, height, data, exact in cases:
    dc = DeviceChains(system, height, data, seed=5, exact_jacobian=exact, hitmap=True, n_value_bins=100, reference_schedule=True, **o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    failed = dc.infer(check_every=2000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = dc.k.cpu().numpy(); st = dc.status.cpu().numpy(); bi = dc.burned_in_iteration.cpu().numpy()
    ok = all(bool(torch.isfinite(getattr(dc, n)).all()) for n in ("sigma", "rel", "add", "pred", "misfit", "like", "prior", "best_sigma"))
    thk_ = rg.layer_widths(dc.edges, dc.k.to(torch.int64))
    ok &= bool((torch.where(thk_ > 0, thk_, torch.full_like(thk_, 9.0)) > dc.min_width).all()) and k.min() >= 1 and k.max() <= dc.K
    fb = FdemBatch(system, k, dc.sigma.cpu().numpy(), thk_.cpu().numpy(), height, data=data, relative_error=dc.rel[:, 0].cpu().numpy(),
                   additive_error=dc.add[:, 0].cpu().numpy(), waves=dc._o.forward_waves)
    chi2, logl = fb.forward_loglike()
    ok &= bool(torch.allclose(fb.predicted, dc.pred, rtol=1e-9, atol=1e-7)) and bool(torch.allclose(chi2, dc.misfit, rtol=1e-7))
    kh = dc.k_hist.cpu().numpy().sum(axis=1)
    gave_up = ((dc.limited == 1) & (dc.n_resets == 3)).cpu().numpy()          # restarted five times without a single accepted step
    restarted = int((dc.iteration0 > 0).sum())
    expect = np.where(st == 1, n_mc + 2, np.where(gave_up, 0, n_mc))
    ok &= bool(np.array_equal(kh, expect)) and bool(np.array_equal(dc.hitmap.sum(dim=(1, 2)).cpu().numpy(), kh * dc.n_depth_bins))
    ok &= bool(np.array_equal(dc.rel_hist.sum(dim=(1, 2)).cpu().numpy(), kh)) and not (st == 0).any()
    it_total = np.where(st == 1, bi + n_mc + 1, n_mc).sum()
    print(f"{name}: B={B} n_markov_chains={n_mc}: {dt:.1f} s ({it_total/dt/1e6:.1f} M chain-iterations/s incl. idle rows), done {int((st==1).sum())}, "
          f"failed {failed} (given up after restarts {int(gave_up.sum())}), restarted {restarted}, median burn-in iteration {int(np.median(bi[st==1])) if (st==1).any() else -1}, mean k {k.mean():.2f}, max k {k.max()}, "
          f"median misfit {np.median(dc.misfit.cpu().numpy()):.1f}; invariants {'OK' if ok else 'VIOLATED'}", flush=True)
    del dc; torch.cuda.empty_cache()
