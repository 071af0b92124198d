"""BASELINE config 4: 16 384 TDEM soundings x 6 layers on 1 MI355X (SkyTEM: 26 high-moment + 19 low-moment gates).
Prints evals/s for the frequency-domain kernel + time-domain GEMM, and the split between the two."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobipy_amd import synthetic
from geobipy_amd.tdem import TdemBatch, TdemSystem
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
B, L = 16384, 6
nl, sig, thk, h = synthetic.draw_models(B, L, seed=4)
systems = [TdemSystem(os.path.join(G, "SkytemHM.stm")), TdemSystem(os.path.join(G, "SkytemLM.stm"))]
eps = float(os.environ['TD_EPS']) if 'TD_EPS' in os.environ else None      # None: default per-sounding abscissa windows (1e-12); 0: all abscissae
b = TdemBatch(systems, nl, sig, thk, h, (-13.0, 0.0, 2.0), hankel_eps=eps)
b.forward(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    b.forward()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
nodes = [s.node_frequencies().size * s.n_components for s in systems]
import ctypes
from geobipy_amd import _lib
pts = 0
for hh in b._h:
    n_ = ctypes.c_int(0); _lib.check(_lib.load().gbp_fdem_system_npoints(hh.ptr, ctypes.byref(n_))); pts += n_.value
print(f"TDEM config 4: B={B} L={L} gates=45 nodes={nodes}: {ms:.3f} ms/forward -> {B/ms*1e3/1e6:.3f} M evals/s "
      f"({pts} abscissa points per sounding, {B*pts*(72*L+33)/ms*1e3/1e12:.2f} TFLOP/s min-flop)")
