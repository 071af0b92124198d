"""ASan/UBSan smoke run of the C oracle (`make -C oracle asan`): every entry point on all tensor ids, 1..30 layers."""
import ctypes, numpy as np, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fdem_oracle as fo
fo._LIB = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libgbp_oracle_asan.so'))
fo._LIB.oracle_fdem1dfwd.restype = ctypes.c_int; fo._LIB.oracle_fdem1dsen.restype = ctypes.c_int
fo._LIB.oracle_fdem_forward_loglike_batch.restype = ctypes.c_int; fo._LIB.oracle_gauss_loglike.restype = None
for name in ['resolve', 'mixed', 'syn10']:
    s = fo.OracleSystem.read(os.path.join(ROOT, 'tests', 'golden', f'{name}.stm'))
    for L in [1, 2, 7, 30]:
        rng = np.random.default_rng(L)
        sig = np.exp(rng.uniform(-6, 0, L)); thk = np.r_[np.exp(rng.uniform(0, 4, L-1)), np.inf]
        p = fo.predicted_data(s, sig, thk, 30.0); J = fo.sensitivity(s, sig, thk, 30.0)
        fo.gauss_loglike(p, p*1.01, 0.05, 5.0)
    B = 50; nl = np.random.default_rng(1).integers(1, 9, B); sig = np.exp(np.random.default_rng(2).uniform(-6,0,(B,8))); thk = np.exp(np.random.default_rng(3).uniform(0,4,(B,8)))
    fo.forward_loglike_batch(s, nl, sig, thk, np.full(B, 30.0), np.ones((B, 2*s.nF)), 0.05, 5.0, nthreads=1)
print('asan/ubsan clean')
