"""A/B timing of library variants on the SAME box, interleaved: python scripts/ab_headline.py libA.so libB.so [...]
Each variant runs the headline workload (65 536 soundings x 10 frequencies x 8 layers, fused forward + likelihood) in its own
process, three rounds, alternating."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import geobipy_amd._lib as L
L.LIB_PATH = sys.argv[1]
from geobipy_amd import FdemBatch, synthetic
s = synthetic.syn10_system()
B = 65536
nl, sig, thk, h = synthetic.draw_models(B, 8, seed=synthetic.SEED)
fb = FdemBatch(s, nl, sig, thk, h, data=np.full((B, 20), 80.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0),
               hankel_eps_ppm=(0.0 if len(sys.argv) > 2 and sys.argv[2] == "all" else None))
for _ in range(20): fb.forward_loglike()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(200): fb.forward_loglike()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200)
print("%%.4f" %% best)
''' % ROOT
libs = sys.argv[1:]
mode = os.environ.get("AB_MODE", "")
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)] + ([mode] if mode else []), capture_output=True, text=True)
        try:
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(out.stderr[-2000:]); raise
for l in libs:
    print("%-50s ms per launch: %s  -> best %.4f ms = %.2f M evals/s" % (os.path.basename(l), res[l], min(res[l]), 65536 / min(res[l]) / 1e3))
