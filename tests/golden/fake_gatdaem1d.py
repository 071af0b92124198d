"""Stand-in for GA-AEM's ``gatdaem1d`` Python module -- BUILD-CONTAINER TOOL for the recorders in this directory, nothing else.

The reference's time-domain classes subclass / call ``gatdaem1d`` (system/TdemSystem_GAAEM.py:8-35, model/Model.py:152-159,
system/Loop_pair.py:63-77, forwardmodelling/Electromagnetic/TD/tdem1d.py:89-154), which is not installable here.  To RUN the
reference's own TdemDataPoint / Inference1D code -- for the layout of its results containers (make_hdf_schema_tdem.py) and for a
seeded trace of its sampler's decisions with several error levels (make_mcmc_trace_tdem.py) -- this module offers the same
surface with THIS REPOSITORY'S oracle (oracle/tdem_oracle.py) as the physics:

    TDAEMSystem(stmfile): .windows.centre, .nwindows, .loopRadius(), .waveform, .forwardmodel(G, E) -> .PX .. .SZ,
                          .fm_dlogc(G, E) -> (fm, JX, JY, JZ), .derivative(CONDUCTIVITYDERIVATIVE, layer)
    Earth(conductivity, thickness), Geometry(tx_height, tx_roll, tx_pitch, tx_yaw, dx, dy, dz, rx_roll, rx_pitch, rx_yaw)

What such recordings pin is therefore the reference's HOST logic around the forward operator (container layout, error-level
moves, proposals, acceptance), never GA-AEM's numbers: every forward value in them is this repository's.  Derivatives are central
differences of the oracle in ln sigma (exactly what tests' OracleTdEngine does, so both sides see identical numbers).
It is never imported by the product, by tests, or on the GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import tdem_oracle as to          # noqa: E402


class Earth:
    def __init__(self, conductivity, thickness):
        self.conductivity = np.array(conductivity, dtype=np.float64)
        self.thickness = np.array(thickness, dtype=np.float64)


class Geometry:
    def __init__(self, tx_height, tx_roll, tx_pitch, tx_yaw, txrx_dx, txrx_dy, txrx_dz, rx_roll, rx_pitch, rx_yaw):
        self.g = np.array([tx_height, tx_roll, tx_pitch, tx_yaw, txrx_dx, txrx_dy, txrx_dz, rx_roll, rx_pitch, rx_yaw], dtype=np.float64)


class _Windows:
    pass


class _Waveform:
    pass


class _Response:
    pass


class TDAEMSystem:
    CONDUCTIVITYDERIVATIVE = 1

    def __init__(self, stmfile):
        self.stm = to.parse_stm(stmfile)
        self.windows = _Windows()
        self.windows.centre = self.stm["windows"].mean(axis=1)
        self.windows.start, self.windows.end = self.stm["windows"][:, 0].copy(), self.stm["windows"][:, 1].copy()
        self.nwindows = self.windows.centre.size
        self.waveform = _Waveform()
        self.waveform.transmitterTime, self.waveform.transmitterCurrent = self.stm["wave"][:, 0], self.stm["wave"][:, 1]
        self.waveform.time, self.waveform.current = self.waveform.transmitterTime, self.waveform.transmitterCurrent
        self._last = None

    def loopRadius(self):
        return float(self.stm.get("ModellingLoopRadius", 0.0))

    # GA-AEM's own sign convention: the reference negates Z afterwards (TdemDataPoint.py:1013-1015), so Z leaves here negated
    # relative to the oracle's reference-convention output; X / Y as they are.
    def _windows_xyz(self, g, sigma, thk):
        stm_all = dict(self.stm)
        out = {}
        full = to.forward_geometry(stm_all, sigma, thk, g)
        n, k = self.nwindows, 0
        for c in "XYZ":
            if float(self.stm.get(c + "OutputScaling", 0.0)) != 0.0:
                out[c] = full[k * n:(k + 1) * n] * (-1.0 if c == "Z" else 1.0)
                k += 1
            else:
                out[c] = np.zeros(n)
        return out

    def forwardmodel(self, G, E):
        self._last = (G.g.copy(), E.conductivity.copy(), E.thickness.copy())
        w = self._windows_xyz(*self._last)
        r = _Response()
        r.SX, r.SY, r.SZ = w["X"], w["Y"], w["Z"]
        p = np.zeros(3)
        k = 0
        pf = to.primary_field(self.stm, G.g)
        for j, c in enumerate("XYZ"):
            if float(self.stm.get(c + "OutputScaling", 0.0)) != 0.0:
                p[j] = pf[k] * (-1.0 if c == "Z" else 1.0)
                k += 1
        r.PX, r.PY, r.PZ = p
        return r

    def _dlog(self, g, sigma, thk, layer, eps=1e-4):
        sp, sm = sigma.copy(), sigma.copy()
        sp[layer] *= np.exp(eps)
        sm[layer] *= np.exp(-eps)
        a, b = self._windows_xyz(g, sp, thk), self._windows_xyz(g, sm, thk)
        return {c: (a[c] - b[c]) / (2 * eps) for c in "XYZ"}

    def fm_dlogc(self, G, E):
        fm = self.forwardmodel(G, E)
        g, sigma, thk = self._last
        J = {c: np.empty((sigma.size, self.nwindows)) for c in "XYZ"}
        for m in range(sigma.size):
            d = self._dlog(g, sigma, thk, m)
            for c in "XYZ":
                J[c][m] = d[c]
        return fm, J["X"], J["Y"], J["Z"]

    def derivative(self, kind, layer):
        """d response / d sigma_layer (1-based layer) of the model of the last forwardmodel call."""
        g, sigma, thk = self._last
        d = self._dlog(g, sigma, thk, layer - 1)
        r = _Response()
        r.SX, r.SY, r.SZ = (d[c] / sigma[layer - 1] for c in "XYZ")
        return r
