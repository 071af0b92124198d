"""Throughput of the device sampler on time-domain data (SkyTEM low moment, 19 gates): TdemDeviceChains -> gbp_rj_run_td."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):                     # A/B builds of the library (scripts/ab/*.so): this script only, never the product
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd.tdem import TdemDeviceChains
from test_tdem_sampler import _survey, OFFSET
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,8192").split(",")]:
    s, h, data, scale, opts, _ = _survey(B, seed=2, stm=tuple(os.environ.get("TD_STM", "SkytemLM.stm").split(",")))
    dc = TdemDeviceChains(s, h, data, OFFSET, seed=1, hankel_eps=(float(os.environ['TD_EPS']) if 'TD_EPS' in os.environ else None), **dict(opts, **({'forward_waves': int(os.environ['TD_FW'])} if 'TD_FW' in os.environ else {})))
    m0 = float(dc.misfit.median())
    dc.run(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); dc.run(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sm = dc.summaries().cpu().numpy()
    print(f"B={B}: {1e3*dt/300:.2f} ms per lockstep iteration -> {B*300/dt/1e6:.2f} M chain-iterations/s "
          f"(nodes {dc._W.shape[0]//2}, gates {data.shape[1]}, mean k {sm[:,3].mean():.2f}, median misfit {m0:.0f} -> {np.median(sm[:,0]):.1f})", flush=True)
