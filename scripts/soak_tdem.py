"""Long-run check of the time-domain device sampler under the reference's schedule (burn-in, stop, restarts, re-packing):
SkyTEM low moment, synthetic soundings; every chain must end done or failed with a finite state and consistent counters."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd.tdem import TdemDeviceChains
from test_tdem_sampler import _survey, OFFSET
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
eps = float(os.environ["TD_EPS"]) if "TD_EPS" in os.environ else None
s, h, data, scale, opts, _ = _survey(B, seed=3)
opts = dict(opts, n_markov_chains=n_mc)
dc = TdemDeviceChains(s, h, data, OFFSET, seed=2, reference_schedule=True, hitmap=True, n_value_bins=60, hankel_eps=eps, **opts)
torch.cuda.synchronize(); t0 = time.perf_counter()
failed = dc.infer(check_every=1000)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st, bi = dc.status.cpu().numpy(), dc.burned_in_iteration.cpu().numpy()
gave_up = ((dc.limited == 1) & (dc.n_resets == 3)).cpu().numpy()
ok = all(bool(torch.isfinite(getattr(dc, n)).all()) for n in ("sigma", "rel", "add", "pred", "misfit", "like", "prior", "best_sigma"))
kh = dc.k_hist.cpu().numpy().sum(axis=1)
expect = np.where(st == 1, n_mc + 2, np.where(gave_up, 0, n_mc))
ok &= bool(np.array_equal(kh, expect)) and not (st == 0).any()
ok &= bool(np.array_equal(dc.hitmap.sum(dim=(1, 2)).cpu().numpy(), kh * dc.n_depth_bins))
print(f"TDEM soak: B={B} n_markov_chains={n_mc} window eps={eps}: {dt:.1f} s, done {int((st == 1).sum())}, failed {failed} (given up {int(gave_up.sum())}), "
      f"restarted {int((dc.iteration0 > 0).sum())}, median burn-in {int(np.median(bi[st == 1])) if (st == 1).any() else -1}, "
      f"median misfit of the done chains {np.median(dc.misfit.cpu().numpy()[st == 1]) if (st == 1).any() else float('nan'):.1f} ({data.shape[1]} gates); "
      f"invariants {'OK' if ok else 'VIOLATED'}")
