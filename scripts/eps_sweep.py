import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from geobipy_amd import FdemBatch, synthetic
s=synthetic.syn10_system()
B,L=65536,8
nl,sig,thk,h=synthetic.draw_models(B,L,seed=synthetic.SEED+2)
ex=FdemBatch(s,nl,sig,thk,h,hankel_eps_ppm=0.0,data=np.full((B,20),80.0),relative_error=np.full(B,.05),additive_error=np.full(B,5.0)); ex.forward_loglike()
p0=ex.predicted.clone()
for eps in [1e-12,3e-12,1e-11,3e-11,1e-10,1e-9]:
    b=FdemBatch(s,nl,sig,thk,h,hankel_eps_ppm=eps,data=np.full((B,20),80.0),relative_error=np.full(B,.05),additive_error=np.full(B,5.0))
    pts=[b._h.bin_points(a) for a in range(25,46,2)]
    for _ in range(5): b.forward_loglike(want_pred=False)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): b.forward_loglike(want_pred=False)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    b.forward_loglike()
    print(eps, pts, '%.3f ms %.1f M/s'%(dt*1e3,B/dt/1e6), 'maxdiff %.2e'%float((b.predicted-p0).abs().max()))
