"""ctypes front-end of oracle/fdem1d_oracle.c (TEST INFRASTRUCTURE, not product code).

Restates, on the CPU in scalar fp64, the reference path
``FdemDataPoint.forward -> fdem1dfwd -> nbFdem1dfwd`` (+ ``nbFdem1dsen``) and
``DataPoint.data_misfit / likelihood`` -- see the C file for the line-by-line citations.
Parity pin: tests/test_oracle_golden.py checks it against fixtures dumped from the imported
reference and against the reference's own ``tests/data_checks/resolve_*_clean.csv``.
"""
import ctypes
import os
import subprocess

import numpy as np

from .gs_filters import W0_J0_120, W1_J1_140, base_abscissae

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile libgbp_oracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "libgbp_oracle.so")
    src = os.path.join(_HERE, "fdem1d_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgbp_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgbp_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_fdem1dfwd.restype = ctypes.c_int
        _LIB.oracle_fdem1dsen.restype = ctypes.c_int
        _LIB.oracle_fdem_forward_loglike_batch.restype = ctypes.c_int
        _LIB.oracle_gauss_loglike.restype = None
        _LIB.oracle_set_exact_jacobian.restype = None
    return _LIB


def set_exact_jacobian(exact):
    """Test switch: ``sensitivity`` returns the true derivative of the forward recursion instead of the reference's M1_1
    expression (which is not one).  Process-wide; the CPU replays of exact-Jacobian device chains set it."""
    lib().oracle_set_exact_jacobian(ctypes.c_int(1 if exact else 0))


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class OracleSystem:
    """Plain-array stand-in for the tables of ``FdemSystem`` (system/FdemSystem.py:52-109,199-203).

    orientation letters follow EmLoop.py:168-181 (x, y, z = 0, 1, 2);
    tensor_id = 1 + 3*rx_orient + tx_orient.
    """

    _ORI = {"x": 0, "y": 1, "z": 2}

    def __init__(self, frequencies, tx_orient, tx_moment, tx_xyz, rx_orient, rx_moment, rx_xyz):
        self.frequencies = np.asarray(frequencies, dtype=np.float64)
        F = self.frequencies.size
        to = np.array([self._ORI[str(o).strip()] for o in tx_orient], dtype=np.int32)
        ro = np.array([self._ORI[str(o).strip()] for o in rx_orient], dtype=np.int32)
        self.tid = (1 + ro * 3 + to).astype(np.int32)
        self.tx_moment = np.asarray(tx_moment, dtype=np.float64)
        self.rx_moment = np.asarray(rx_moment, dtype=np.float64)
        self.tx_xyz = np.asarray(tx_xyz, dtype=np.float64).reshape(F, 3)
        self.rx_xyz = np.asarray(rx_xyz, dtype=np.float64).reshape(F, 3)
        off = self.rx_xyz - self.tx_xyz                      # FdemSystem.py:52-57
        self.rx_off = np.ascontiguousarray(off[:, 0])        # fdem1d.py:34
        self.separation = np.linalg.norm(off.T, axis=0)      # FdemSystem.py:59-61
        self.scale = self.tx_moment * self.rx_moment         # fdem1d.py:33
        l0, l1 = base_abscissae()
        r = 1.0 / self.separation
        self.lamda0 = np.ascontiguousarray(l0[None, :] * r[:, None])   # FdemSystem.py:67-83
        self.lamda1 = np.ascontiguousarray(l1[None, :] * r[:, None])   # FdemSystem.py:85-101
        self.lamda02 = self.lamda0 ** 2.0
        self.lamda12 = self.lamda1 ** 2.0
        self.w0 = W0_J0_120
        self.w1 = W1_J1_140

    @property
    def nF(self):
        return self.frequencies.size

    @classmethod
    def read(cls, filename):
        """Parse a .stm FDEM system file (FdemSystem.py:146-183): header + one CSV row per frequency."""
        rows = []
        with open(filename) as f:
            next(f)
            for line in f:
                p = [x.strip() for x in line.split(",")]
                if len(p) >= 11:
                    rows.append(p)
        fr = [float(p[0]) for p in rows]
        return cls(fr, [p[1] for p in rows], [float(p[2]) for p in rows],
                   [[float(p[3]), float(p[4]), float(p[5])] for p in rows],
                   [p[6] for p in rows], [float(p[7]) for p in rows],
                   [[float(p[8]), float(p[9]), float(p[10])] for p in rows])


def _sysargs(s, height):
    tH = height + s.tx_xyz[:, 2]            # fdem1d.py:31
    rH = -tH + s.rx_xyz[:, 2]               # fdem1d.py:32
    return tH, rH


def forward(system, conductivity, thickness, height):
    """nbFdem1dfwd for one sounding -> complex128[F]."""
    s = system
    sig, psig = _d(conductivity)
    L = sig.size
    thk, pthk = _d(thickness)
    assert thk.size == L
    zeros, pz = _d(np.zeros(L))
    tH, rH = _sysargs(s, float(height))
    out = np.empty(2 * s.nF)
    keep = [_i(s.tid), _d(s.frequencies), _d(tH), _d(rH), _d(s.tx_moment), _d(s.rx_off), _d(s.separation),
            _d(s.w0), _d(s.lamda0), _d(s.lamda02), _d(s.w1), _d(s.lamda1), _d(s.lamda12), _d(s.scale)]
    rc = lib().oracle_fdem1dfwd(ctypes.c_int(s.nF), ctypes.c_int(L), *[k[1] for k in keep],
                                psig, pz, pz, pthk, out.ctypes.data_as(_dp))
    if rc != 0:
        raise ValueError("unsupported tensor id in system (reference leaves H undefined)")
    return out[0::2] + 1j * out[1::2]


def sensitivity(system, conductivity, thickness, height):
    """nbFdem1dsen for one sounding -> complex128[F, L] (d/dln sigma_k, ppm)."""
    s = system
    sig, psig = _d(conductivity)
    L = sig.size
    thk, pthk = _d(thickness)
    zeros, pz = _d(np.zeros(L))
    tH, rH = _sysargs(s, float(height))
    out = np.empty(2 * s.nF * L)
    keep = [_i(s.tid), _d(s.frequencies), _d(tH), _d(rH), _d(s.tx_moment), _d(s.rx_off), _d(s.separation),
            _d(s.w0), _d(s.lamda0), _d(s.lamda02), _d(s.w1), _d(s.lamda1), _d(s.lamda12), _d(s.scale)]
    rc = lib().oracle_fdem1dsen(ctypes.c_int(s.nF), ctypes.c_int(L), *[k[1] for k in keep],
                                psig, pz, pz, pthk, out.ctypes.data_as(_dp))
    if rc != 0:
        raise ValueError("unsupported tensor id in system")
    return (out[0::2] + 1j * out[1::2]).reshape(s.nF, L)


def predicted_data(system, conductivity, thickness, height):
    """FdemDataPoint._forward1D layout: [Re(out), Im(out)] (FdemDataPoint.py:544-545)."""
    c = forward(system, conductivity, thickness, height)
    return np.concatenate([c.real, c.imag])


def gauss_loglike(pred, obs, rel, add):
    """(std[N], chi2, logL, n_active) -- DataPoint.py:268-282, 491-525; MvNormalDistribution.py:201-216."""
    p, pp = _d(pred)
    o, po = _d(obs)
    N = p.size
    sd = np.empty(N)
    chi2 = ctypes.c_double()
    logl = ctypes.c_double()
    na = ctypes.c_int()
    lib().oracle_gauss_loglike(ctypes.c_int(N), pp, po, ctypes.c_double(rel), ctypes.c_double(add),
                               sd.ctypes.data_as(_dp), ctypes.byref(chi2), ctypes.byref(logl), ctypes.byref(na))
    return sd, chi2.value, logl.value, na.value


def forward_loglike_batch(system, nlayers, sigma, thk, height, obs=None, rel=None, add=None, nthreads=1):
    """Batched oracle (and the "port" CPU baseline of bench.py): returns pred[B,2F], chi2[B], logL[B]."""
    s = system
    nl, pnl = _i(nlayers)
    B = nl.size
    sig, psig = _d(sigma)
    Lmax = sig.shape[1]
    th, pth = _d(thk)
    h, ph = _d(height)
    assert s.nF <= 64
    pred = np.empty((B, 2 * s.nF))
    chi2 = np.full(B, np.nan)
    logl = np.full(B, np.nan)
    if obs is not None:
        o, po = _d(obs)
        r, pr = _d(np.broadcast_to(rel, (B,)))
        a, pa = _d(np.broadcast_to(add, (B,)))
    else:
        po = pr = pa = None
    keep = [_i(s.tid), _d(s.frequencies), _d(s.tx_xyz[:, 2]), _d(s.rx_xyz[:, 2]), _d(s.tx_moment), _d(s.scale),
            _d(s.rx_off), _d(s.separation), _d(s.w0), _d(s.lamda0), _d(s.lamda02), _d(s.w1), _d(s.lamda1),
            _d(s.lamda12)]
    rc = lib().oracle_fdem_forward_loglike_batch(
        ctypes.c_int(B), ctypes.c_int(s.nF), ctypes.c_int(Lmax), *[k[1] for k in keep], pnl, psig, pth, ph,
        po, pr, pa, pred.ctypes.data_as(_dp), chi2.ctypes.data_as(_dp), logl.ctypes.data_as(_dp),
        ctypes.c_int(nthreads))
    if rc != 0:
        raise ValueError("unsupported tensor id in system")
    return pred, chi2, logl
