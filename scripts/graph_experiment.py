"""Does a hipGraph help the sampler loop?  Captures 20 iterations (torch.cuda.CUDAGraph around gbp_rj_run; the iteration
numbers are baked in, so this is a timing experiment only) and replays them next to plain stream launches."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.filterwarnings("ignore")
from geobipy_amd import FdemSystem, DeviceChains
from test_rjmcmc import RESOLVE_OPTIONS
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "mcmc_detail.npz")); s = FdemSystem.read(os.path.join(G, "resolve.stm"))
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
for B in (256, 1024, 4096):
    rng = np.random.default_rng(1)
    dc = DeviceChains(s, rng.uniform(25, 40, B), np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, B)[:, None], seed=1, exact_jacobian=True, **o)
    dc.run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); dc.run(400); torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / 400
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(st):
            dc.run(20); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                dc.run(20)
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 400
        print(f"B={B}: plain {1e3*t_plain:.3f} ms/iter, graph replay {1e3*t_graph:.3f} ms/iter", flush=True)
    except Exception as e:
        print(f"B={B}: plain {1e3*t_plain:.3f} ms/iter, graph capture failed: {str(e)[:200]}", flush=True)
