"""Synthetic systems / soundings of SURVEY 8(d) -- inputs only (no arithmetic of the hot path).

Used by bench.py and the tests so that both draw the same seeded batches:
RNG numpy Generator(PCG64DXSM(seed)); sigma log-uniform [1e-3, 1] S/m; finite thicknesses log-uniform
[1, 50] m; height uniform [25, 45] m; relative error 5 %, additive error 5 ppm.
"""
import numpy as np
from numpy.random import Generator, PCG64DXSM

from .system import CircularLoop, FdemSystem

SEED = 20260928


def syn10_system():
    """10 vertical-coplanar (zz) frequencies 380 Hz .. 129.55 kHz, separation 7.9 m, unit moments."""
    fr = np.logspace(np.log10(380.0), np.log10(129550.0), 10)
    n = fr.size
    tx = CircularLoop(orientation=["z"] * n, moment=np.ones(n), x=np.zeros(n), y=np.zeros(n), z=np.zeros(n))
    rx = CircularLoop(orientation=["z"] * n, moment=np.ones(n), x=np.full(n, 7.9), y=np.zeros(n), z=np.zeros(n))
    return FdemSystem(fr, tx, rx)


def draw_models(B, L, seed=SEED, Lmax=None):
    """(nlayers[B], sigma[B,Lmax], thk[B,Lmax], height[B]) -- thk[:, L-1] is 0 (half-space, never read)."""
    rng = Generator(PCG64DXSM(seed))
    Lmax = L if Lmax is None else Lmax
    sigma = np.ones((B, Lmax))
    thk = np.zeros((B, Lmax))
    sigma[:, :L] = np.exp(rng.uniform(np.log(1e-3), np.log(1.0), size=(B, L)))
    thk[:, :L - 1] = np.exp(rng.uniform(np.log(1.0), np.log(50.0), size=(B, L - 1)))
    height = rng.uniform(25.0, 45.0, size=B)
    return np.full(B, L, dtype=np.int32), sigma, thk, height


def redraw_sigma(B, L, seed, Lmax=None):
    """Independent proposal conductivities (so a forward solve cannot be cached between rounds)."""
    rng = Generator(PCG64DXSM(seed))
    Lmax = L if Lmax is None else Lmax
    sigma = np.ones((B, Lmax))
    sigma[:, :L] = np.exp(rng.uniform(np.log(1e-3), np.log(1.0), size=(B, L)))
    return sigma


def noisy_observations(clean, seed=SEED + 1, rel=0.05, add=5.0):
    """observed = clean * (1 + rel N(0,1)) + add N(0,1)  (ppm)."""
    rng = Generator(PCG64DXSM(seed))
    return clean * (1.0 + rel * rng.normal(size=clean.shape)) + add * rng.normal(size=clean.shape)
