"""CPU oracle for the TDEM path -- TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

The reference delegates this arithmetic to GA-AEM's ``gatdaem1d`` C++ library
(forwardmodelling/Electromagnetic/TD/tdem1d.py:89-96), which is not vendored, not version-pinned and absent
from this environment, so there is no reference implementation to restate line by line.  This file restates
the published pipeline (layered-earth frequency response -> log-frequency spline -> waveform spectrum ->
inverse FFT -> window averaging) in plain numpy/scipy, independently of geobipy_amd/tdem.py (direct tanh
recursion, explicit per-window quadrature, no precomputed operator), and is itself pinned only against the
reference's CSV fixtures (tests/golden/skytem_*_clean.csv, tempest_*_clean.csv) to the level GA-AEM's own
numerics allow (typically < 1 % on gates with signal; see tests/test_tdem.py for the measured bounds) and against
the closed-form step-off transient of a vertical dipole on a half-space (0.5 - 1.5 % of the largest gate).
"""
import numpy as np
from scipy.interpolate import CubicSpline
from scipy.special import j1

from .gs_filters import W0_J0_120, W1_J1_140, base_abscissae

MU0 = 4e-7 * np.pi


def parse_stm(path):
    d = {"wave": [], "windows": []}
    mode = None
    for line in open(path):
        s = line.split("//")[0].strip()
        if not s:
            continue
        if "WaveFormCurrent Begin" in s:
            mode = "wave"
        elif "WindowTimes Begin" in s:
            mode = "windows"
        elif s.endswith(" End"):
            mode = None
        elif mode:
            d[mode].append([float(x) for x in s.split()])
        elif "=" in s:
            k, v = [x.strip() for x in s.split("=", 1)]
            d[k] = v
    d["wave"], d["windows"] = np.array(d["wave"]), np.array(d["windows"])
    return d


def rte(lam, omega, sig, thk):
    """Quasi-static TE reflection coefficient with the textbook tanh recursion (Ward & Hohmann 4.29-4.31)."""
    L = len(sig)
    u = [np.sqrt(lam ** 2 + 1j * omega * MU0 * s) for s in sig]
    Y = u[L - 1]
    for k in range(L - 2, -1, -1):
        T = np.tanh(u[k] * thk[k])
        Y = u[k] * (Y + u[k] * T) / (u[k] + Y * T)
    return (lam - Y) / (lam + Y)


BOXCAR_TOLERANCE = 1.0e-7      # seconds; see scripts/tdem_study/README.md (inferred from tempest_*_clean.csv gates 7, 8)


def node_frequencies(stm, per_decade=None, below_base=1):
    """f0 * 10^(i / FrequenciesPerDecade), i = -below_base ... up to the first node at or above Nyquist."""
    f0, fs = float(stm["BaseFrequency"]), float(stm["WaveformDigitisingFrequency"])
    fpd = float(stm.get("FrequenciesPerDecade", 5)) if per_decade is None else float(per_decade)
    out, i = [], -below_base
    while True:
        out.append(f0 * 10.0 ** (i / fpd))
        if i >= 0 and out[-1] >= 0.5 * fs * (1.0 - 1e-12):
            return np.array(out)
        i += 1


def secondary_fields(stm, sig, thk, alt, dx, dy, dz, freqs):
    """(Hz, Hx) per unit moment at the receiver, horizontal loop of radius a (a = 0: vertical dipole)."""
    r = np.hypot(dx, dy)
    a = float(stm.get("ModellingLoopRadius", 0.0))
    l0, l1 = base_abscissae()
    hz, hx = np.empty(len(freqs), complex), np.empty(len(freqs), complex)
    H = 2.0 * alt + dz
    if r == 0.0:                         # receiver on the loop axis: J0(0) = 1, the J1(lam a) of the loop is the filter kernel
        assert a > 0.0, "coincident dipoles have no finite response"
        lam = l1 / a
        for i, f in enumerate(freqs):
            hz[i] = np.sum(rte(lam, 2 * np.pi * f, sig, thk) * np.exp(-lam * H) * lam / (2 * np.pi * a) * W1_J1_140) / a
            hx[i] = 0.0
        return hz, hx
    for i, f in enumerate(freqs):
        for which, lam, w in (("z", l0 / r, W0_J0_120), ("x", l1 / r, W1_J1_140)):
            R = rte(lam, 2 * np.pi * f, sig, thk)
            src = lam * j1(lam * a) / (2 * np.pi * a) if a > 0 else lam ** 2 / (4 * np.pi)
            v = np.sum(R * np.exp(-lam * H) * src * w) / r
            if which == "z":
                hz[i] = v
            else:
                hx[i] = -dx / r * v
    return hz, hx


def forward(stm, sig, thk, alt, dx, dy, dz, per_decade=None):
    """Window values in the reference's channel order (x windows then z windows, only scaled components)."""
    f0, fs = float(stm["BaseFrequency"]), float(stm["WaveformDigitisingFrequency"])
    N = int(round(fs / f0))
    wt, wc = stm["wave"][:, 0], stm["wave"][:, 1]
    t = wt[0] + np.arange(N) / fs
    if abs((wt[-1] - wt[0]) - 0.5 / f0) <= 2.0 / fs:
        c = np.interp(t[: N // 2], wt, wc)
        cur = np.concatenate([c, -c])
    else:
        cur = np.interp(t, wt, wc)
    I = np.fft.rfft(cur)
    fk = np.arange(N // 2 + 1) * f0
    fn = node_frequencies(stm, per_decade)
    hz, hx = secondary_fields(stm, sig, thk, alt, dx, dy, dz, fn)
    moment = float(stm.get("NumberOfTurns", 1)) * float(stm.get("PeakCurrent", 1)) * float(stm.get("LoopArea", 1))
    fac = np.full(fk.size, MU0 * moment, dtype=complex)
    if stm.get("OutputType", "dB/dt").lower().startswith("db"):
        fac *= -2j * np.pi * fk
    if "CutOffFrequency" in stm:
        for fc, n in zip(stm["CutOffFrequency"].split(), stm["Order"].split()):
            fac *= (1.0 / (1.0 + 1j * fk / float(fc))) ** int(float(n))
    out = []
    for comp, H in (("X", hx), ("Z", hz)):
        scale = float(stm.get(comp + "OutputScaling", 0.0))
        if scale == 0.0:
            continue
        lf = np.log10(np.clip(fk[1:], fn[0], fn[-1]))
        Hk = np.zeros(fk.size, complex)
        Hk[1:] = CubicSpline(np.log10(fn), H.real, bc_type="natural")(lf) + 1j * CubicSpline(np.log10(fn), H.imag, bc_type="natural")(lf)
        spec = I * fac * Hk * scale
        spec[0] = 0.0
        r = np.fft.irfft(spec, N)
        for a, b in stm["windows"]:
            if stm.get("WindowWeightingScheme", "Boxcar").lower().startswith("area"):
                q = np.linspace(a, b, 257)
                out.append(np.trapezoid(np.interp(q, t, r), q) / (b - a))
            else:
                m = (t >= a - BOXCAR_TOLERANCE) & (t <= b + BOXCAR_TOLERANCE)
                out.append(r[m].mean())
    return np.array(out)
