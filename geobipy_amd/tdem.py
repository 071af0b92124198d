"""Time-domain EM (TDEM) forward solve on the GPU: ``TdemSystem`` + ``TdemBatch``.

Reference path: ``TdemDataPoint.forward`` (data/datapoint/TdemDataPoint.py:997-1022) ->
``tdem1dfwd`` / ``gaTdem1dfwd`` (forwardmodelling/Electromagnetic/TD/tdem1d.py:13-37, 89-96) ->
``gatdaem1d.TDAEMSystem.forwardmodel(Geometry, Earth)``.  The arithmetic lives in GA-AEM's C++ library,
which is neither vendored nor pinned by the reference and is absent here, so this module RESTATES the
published pipeline from the ``.stm`` system-file semantics and is pinned only at the reference's CSV
boundary (tests/golden/skytem_*_clean.csv, tempest_*_clean.csv; see DESIGN.md section 3.7 for the measured
agreement -- "parity unpinned" in the sense of the project rules):

  1. frequency domain: quasi-static layered-earth response of a horizontal circular loop (or vertical
     magnetic dipole) transmitter at spline-node frequencies, vertical (J0 filter) and inline horizontal
     (J1 filter) secondary field at the receiver.  This is the SAME reflection-coefficient recursion and
     Hankel digital filter as the FDEM path and runs in the same HIP kernel (``k_fdem_forward`` on a handle
     built by ``gbp_hankel_system_create_raw``).
  2. time domain: everything after the spline nodes is LINEAR in the nodal values and independent of the
     sounding -- cubic spline in log-frequency onto the harmonics of the base frequency, multiplication by
     the spectrum of the digitised (bipolar, periodic) current waveform, by i*omega*mu0 (dB/dt) or mu0 (B)
     and by the receiver low-pass filters, inverse FFT over one period, window averaging.  It is therefore
     folded ONCE per system into a dense operator W[2*n_nodes, n_windows] (numpy, host) and applied to the
     whole batch as one fp64 GEMM on the GPU:  windows[B, n_windows] = nodal[B, 2*n_nodes] @ W.

Conventions recovered from the reference's fixtures: moment = NumberOfTurns * PeakCurrent * LoopArea;
an order-n ``LowPassFilter`` is n cascaded first-order sections; a waveform table that covers half a period
is continued with opposite polarity; "dB/dt" output is the receiver voltage convention -dB/dt; the
reference negates GA-AEM's z components (TdemDataPoint.py:1013-1015), which makes Z positive-up.
Geometry: any receiver offset and any attitude (roll / pitch / yaw of both loops) per sounding, X / Y / Z outputs -- the
kernels evaluate the basis integrals of the transmitter-receiver frame and a per-row matrix mixes their spectra into the output
components inside the window kernel (geobipy_amd/tdem_geometry.py); non-zero angles are unpinned by any reference vector.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .rjmcmc_gpu import DeviceChains
from .filters import W0_J0_120, W1_J1_140, base_abscissae
from .tdem_geometry import LOOP_PAIR_SCALARS, device_angle_moves, gaaem_tuple, loop_pair_moves, loop_pair_values, ON_AXIS_RHO, GeometryMix, from_loops, gaaem_geometry

MU0 = 4.0e-7 * np.pi


def read_stm(filename):
    """Parse a GA-AEM ``.stm`` time-domain system file into a dict (blocks are flattened)."""
    d = {"WaveFormCurrent": [], "WindowTimes": []}
    mode = None
    with open(filename) as f:
        for line in f:
            s = line.split("//")[0].strip()
            if not s:
                continue
            if "WaveFormCurrent Begin" in s:
                mode = "WaveFormCurrent"
            elif "WindowTimes Begin" in s:
                mode = "WindowTimes"
            elif s.endswith(" End") or s == "End":
                mode = None
            elif mode is not None:
                d[mode].append([float(x) for x in s.split()])
            elif "=" in s:
                k, v = [x.strip() for x in s.split("=", 1)]
                d[k] = v
    d["WaveFormCurrent"] = np.asarray(d["WaveFormCurrent"], dtype=np.float64)
    d["WindowTimes"] = np.asarray(d["WindowTimes"], dtype=np.float64)
    return d


class _Windows:
    def __init__(self, times):
        self.start, self.end = times[:, 0].copy(), times[:, 1].copy()
        self.centre = 0.5 * (self.start + self.end)


class _Waveform:
    def __init__(self, tc):
        self.transmitterTime, self.transmitterCurrent = tc[:, 0].copy(), tc[:, 1].copy()


class TdemSystem:
    """Time-domain acquisition system read from a ``.stm`` file (mirrors what the reference uses of
    ``gatdaem1d.TDAEMSystem`` / system/TdemSystem_GAAEM.py: ``windows.centre``, ``nwindows``,
    ``loopRadius()``, ``waveform``, ``components``, ``off_time``)."""

    def __init__(self, system_filename, nodes_per_decade=None, nodes_below_base=1, boxcar_tolerance=1.0e-7):
        """``nodes_per_decade`` defaults to the file's ``FrequenciesPerDecade`` (GA-AEM's spline-node density; 5 when the
        block is absent); the nodes sit at BaseFrequency * 10^(i / nodes_per_decade), ``nodes_below_base`` of them below
        the base frequency.  ``boxcar_tolerance`` (seconds): a sample belongs to a Boxcar window when it lies within the
        window widened by this much -- the rule, grid phase and density are what the discretisation study
        (scripts/tdem_study/) found to reproduce the reference's known-answer files."""
        d = read_stm(system_filename)
        self.filename = system_filename
        self.base_frequency = float(d["BaseFrequency"])
        self.sample_frequency = float(d["WaveformDigitisingFrequency"])
        self.moment = float(d.get("NumberOfTurns", 1)) * float(d.get("PeakCurrent", 1)) * float(d.get("LoopArea", 1))
        self.waveform = _Waveform(d["WaveFormCurrent"])
        self.windows = _Windows(d["WindowTimes"])
        assert np.min(np.diff(self.windows.centre)) > 0.0, ValueError(
            "Receiver window times must monotonically increase for system " + system_filename)
        self.weighting = d.get("WindowWeightingScheme", "Boxcar")
        fc = [float(x) for x in d.get("CutOffFrequency", "").split()]
        od = [int(float(x)) for x in d.get("Order", "").split()]
        self.lowpass = list(zip(fc, od))
        self._loop_radius = float(d.get("ModellingLoopRadius", 0.0))
        self.output_type = d.get("OutputType", "dB/dt").strip()
        self.scaling = {c: float(d.get(c.upper() + "OutputScaling", 0.0)) for c in "xyz"}
        self._components = [c for c in "xyz" if self.scaling[c] != 0.0]
        self.frequencies_per_decade = float(d.get("FrequenciesPerDecade", 5))
        # GA-AEM's Hankel quadrature size; this implementation evaluates the same integrals with the 120 / 140-point
        # digital filters of the FDEM path (difference < 2e-5 of the largest gate, scripts/tdem_study/README.md)
        self.hankel_abscissae = int(float(d.get("NumberOfAbsiccaInHankelTransformEvaluation", 21)))
        self.nodes_per_decade = self.frequencies_per_decade if nodes_per_decade is None else float(nodes_per_decade)
        self.nodes_below_base = int(nodes_below_base)
        self.boxcar_tolerance = float(boxcar_tolerance)
        self.off_time = self.windows.centre
        self._op = None

    # -- gatdaem1d-like accessors --------------------------------------------------------------------
    @property
    def nwindows(self):
        return self.windows.centre.size

    def loopRadius(self):
        return self._loop_radius

    @property
    def components(self):
        return self._components

    @property
    def n_components(self):
        return len(self._components)

    @property
    def isGA(self):
        return True

    # -- frequency nodes and the linear time-domain operator ---------------------------------------------
    @property
    def n_samples(self):
        return int(round(self.sample_frequency / self.base_frequency))

    def node_frequencies(self):
        """Spline nodes BaseFrequency * 10^(i / nodes_per_decade), i = -nodes_below_base ... until Nyquist is covered
        (the .stm ``ForwardModelling`` block's ``FrequenciesPerDecade``)."""
        fpd = self.nodes_per_decade
        n = int(np.ceil(np.log10(0.5 * self.sample_frequency / self.base_frequency) * fpd - 1e-9)) + 1
        return self.base_frequency * 10.0 ** ((np.arange(n + self.nodes_below_base) - self.nodes_below_base) / fpd)

    def digitised_current(self):
        """One period of the transmitter current sampled at the digitising frequency; a table that spans
        half a period is continued with opposite polarity."""
        N = self.n_samples
        wt, wc = self.waveform.transmitterTime, self.waveform.transmitterCurrent
        t = wt[0] + np.arange(N) / self.sample_frequency
        T = 1.0 / self.base_frequency
        if abs((wt[-1] - wt[0]) - 0.5 * T) <= 2.0 / self.sample_frequency:
            half = N // 2
            c = np.interp(t[:half], wt, wc)
            cur = np.concatenate([c, -c, np.zeros(N - 2 * half)])
        else:
            cur = np.interp(t, wt, wc)
        return t, cur

    def window_matrix(self, t):
        """A[n_windows, N]: window value = A @ time series (area under the linear interpolant / width, or
        boxcar mean of the samples inside the window)."""
        N = t.size
        dt = 1.0 / self.sample_frequency
        A = np.zeros((self.nwindows, N))
        for w, (a, b) in enumerate(zip(self.windows.start, self.windows.end)):
            if self.weighting.lower().startswith("area"):
                q = np.linspace(a, b, 257)                      # trapezoid over the linear interpolant
                wq = np.full(q.size, (b - a) / (q.size - 1))
                wq[0] *= 0.5
                wq[-1] *= 0.5
                pos = (q - t[0]) / dt
                i0 = np.clip(np.floor(pos).astype(int), 0, N - 2)
                fr = pos - i0
                np.add.at(A[w], i0, wq * (1.0 - fr) / (b - a))
                np.add.at(A[w], i0 + 1, wq * fr / (b - a))
            else:
                m = (t >= a - self.boxcar_tolerance) & (t <= b + self.boxcar_tolerance)
                A[w, m] = 1.0 / m.sum()
        return A

    def time_operator(self):
        """W[2*n_nodes, n_windows] with windows = [Re(nodal), Im(nodal)] @ W, for nodal = mu0-free
        secondary field H(f_node) per unit moment; includes moment, mu0, i*omega (dB/dt -> -dB/dt),
        low-pass filters, waveform spectrum, inverse FFT and window averaging."""
        if self._op is not None:
            return self._op
        from scipy.interpolate import CubicSpline
        fn = self.node_frequencies()
        n = fn.size
        N = self.n_samples
        t, cur = self.digitised_current()
        I = np.fft.rfft(cur)
        fk = np.arange(N // 2 + 1) * self.base_frequency
        S = np.zeros((fk.size, n))
        S[1:] = CubicSpline(np.log10(fn), np.eye(n), bc_type="natural")(np.log10(np.clip(fk[1:], fn[0], fn[-1])))
        fac = np.full(fk.size, MU0 * self.moment, dtype=complex)
        if self.output_type.lower().startswith("db"):
            fac = fac * (-1j * 2.0 * np.pi * fk)                 # receiver voltage convention: -dB/dt
        for fc, order in self.lowpass:
            fac = fac * (1.0 / (1.0 + 1j * fk / fc)) ** order
        fac[0] = 0.0
        A = self.window_matrix(t)
        G = (I * fac)[:, None] * S                                  # spectrum of the response to unit Re nodes
        W = np.empty((2 * n, self.nwindows))
        W[:n] = (A @ np.fft.irfft(G, N, axis=0)).T
        W[n:] = (A @ np.fft.irfft(1j * G, N, axis=0)).T
        self._op = W
        return W

    # -- Hankel tables of the frequency-domain stage ------------------------------------------------------
    def hankel_tables(self, rho, dz, basis=(0,)):
        """Raw point tables for gbp_hankel_system_create_raw: one "frequency" per (basis integral, node), in that order, all
        120 / 140 filter abscissae (the per-sounding abscissa windows are cut from them by gbp_hankel_system_add_bins).
        ``rho``: horizontal transmitter-receiver distance, ``dz``: receiver height above the transmitter, ``basis``: indices
        into tdem_geometry.BASIS -- with K = rTE e^{-lam (2 alt + dz)} and s(lam) = lam J1(lam a) / (2 pi a) the source term of
        the system's horizontal loop of radius a (a = 0: lam^2 / 4 pi, a vertical dipole):

            0 B0L = Int K s J0(lam rho)      1 B1L = Int K s J1(lam rho)           (vertical part of the moment)
            2 B0  = Int K lam^2 J0 / 4 pi    3 B1  = Int K lam^2 J1 / 4 pi    4 BA = Int K lam J1 / (4 pi rho)   (horizontal part)

        Level flight needs B0L (Z output) and B1L (X, Y); output sign and scaling are NOT in the tables (they are in the rows'
        mixing weights).  A receiver on the transmitter's axis (rho = 0): B0L through the loop's own J1(lam a) as the filter
        kernel, B0 at ``ON_AXIS_RHO``; the others vanish."""
        from scipy.special import j1
        rho, a, k4 = float(rho), self._loop_radius, 1.0 / (4.0 * np.pi)
        on_axis = rho == 0.0
        if on_axis and not a > 0.0:
            raise ValueError("a receiver on the axis needs a finite ModellingLoopRadius")
        l0, l1 = base_abscissae()
        fn = self.node_frequencies()
        srcz = (lambda lam: lam * j1(lam * a) / (2.0 * np.pi * a)) if a > 0.0 else (lambda lam: lam * lam * k4)
        npts, wmu, hd0, g, cols = [], [], [], [], []
        for i in basis:
            if on_axis:
                if i == 0:                                   # J0(0) = 1: the loop's own J1(lam a) is the filter kernel
                    lam = l1 / a
                    coef = lam / (2.0 * np.pi * a) * W1_J1_140 / a
                elif i == 2:
                    lam = l0 / ON_AXIS_RHO
                    coef = lam * lam * k4 * W0_J0_120 / ON_AXIS_RHO
                else:
                    raise ValueError("basis integral {} vanishes on the axis".format(i))
            elif i == 0:
                lam = l0 / rho
                coef = srcz(lam) * W0_J0_120 / rho
            elif i == 1:
                lam = l1 / rho
                coef = srcz(lam) * W1_J1_140 / rho
            elif i == 2:
                lam = l0 / rho
                coef = lam * lam * k4 * W0_J0_120 / rho
            elif i == 3:
                lam = l1 / rho
                coef = lam * lam * k4 * W1_J1_140 / rho
            else:
                lam = l1 / rho
                coef = lam * k4 / rho * W1_J1_140 / rho
            for f in fn:
                npts.append(lam.size)
                wmu.append(2.0 * np.pi * f * MU0)
                hd0.append(-dz)
                g.extend([1.0, 0.0])
                z = np.zeros_like(lam)
                cols.append(np.stack([lam * lam, lam, z, coef, z, lam, z]))   # a, u0, coef, ue
        tables = np.concatenate(cols, axis=1)                      # [7, P]
        return (np.asarray(npts, np.int32), np.asarray(wmu), np.asarray(hd0), np.asarray(g),
                np.ascontiguousarray(tables))

    def primary_field(self, dx, dy, dz, attitude=None):
        """Free-space field of the (rotated) transmitter dipole along the receiver's axes in the output units, one value per
        output component in channel order, with the reference's negated-z convention (TdemDataPoint.py:1004-1015), e.g.
        Tempest PX / PZ.  ``attitude``: (tx roll, pitch, yaw, rx roll, pitch, yaw) in GA-AEM's convention, or None."""
        return GeometryMix([self], gaaem_geometry([0.0], (dx, dy, dz), attitude)).primary_field()[0]


class NativeTdemSystem:
    """The C-level time-domain system (``gbp_tdem_system_create`` on the text of a ``.stm`` file): GA-AEM's
    ``TDAEMSystem(stmfile)`` / ``forwardmodel(Geometry, Earth)`` boundary without Python in the path
    (include/geobipy_amd.h "Time-domain systems").  ``forward(geometry[B, 10], nlayers, sigma, thk)`` -> windows on the device."""

    def __init__(self, system_filename):
        lib = _lib.load()
        self._lib = lib
        h = ctypes.c_void_p()
        text = open(system_filename, "rb").read()
        w0, w1 = np.ascontiguousarray(W0_J0_120), np.ascontiguousarray(W1_J1_140)
        _lib.check(lib.gbp_tdem_system_create(text, w0.ctypes.data_as(_lib.c_double_p), w1.ctypes.data_as(_lib.c_double_p), ctypes.byref(h)))
        self.ptr = h
        nw, nc, nn, a = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        _lib.check(lib.gbp_tdem_system_info(h, ctypes.byref(nw), ctypes.byref(nc), ctypes.byref(nn), ctypes.byref(a)))
        self.nwindows, self.n_components, self.n_nodes, self._loop_radius = nw.value, nc.value, nn.value, a.value

    def loopRadius(self):
        return self._loop_radius

    def tables(self):
        """(window centres, spline-node frequencies, W[2 n_nodes, n_windows]) as the library built them."""
        c, f, W = np.empty(self.nwindows), np.empty(self.n_nodes), np.empty((2 * self.n_nodes, self.nwindows))
        dp = lambda x: x.ctypes.data_as(_lib.c_double_p)
        _lib.check(self._lib.gbp_tdem_system_tables(self.ptr, dp(c), dp(f), dp(W)))
        return c, f, W

    def forward(self, geometry, nlayers, sigma, thk, device=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        geometry = np.ascontiguousarray(geometry, dtype=np.float64)
        B = geometry.shape[0]
        assert geometry.shape == (B, 10), ValueError("geometry must have shape [B, 10] (Loop_pair.py:70-77)")
        sg = torch.as_tensor(np.asarray(sigma), dtype=torch.float64).to(device).contiguous()
        th = torch.as_tensor(np.asarray(thk), dtype=torch.float64).to(device).contiguous()
        nl = torch.as_tensor(np.broadcast_to(np.asarray(nlayers), (B,)).copy(), dtype=torch.int32).to(device)
        out = torch.empty((B, self.n_components * self.nwindows), dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            _lib.check(self._lib.gbp_tdem_forward(self.ptr, B, geometry.ctypes.data_as(_lib.c_double_p), sg.shape[1], nl.data_ptr(),
                                                  sg.data_ptr(), th.data_ptr(), out.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
        return out

    def fm_dlogc(self, geometry, nlayers, sigma, thk, device=None):
        """(windows [B, N], J [B, N, Lmax] = d windows / d ln sigma) -- gbp_tdem_fm_dlogc, the C-level counterpart of
        ga_fm_dlogc / gatdaem1d's derivative call (TD/tdem1d.py:98-154)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        geometry = np.ascontiguousarray(geometry, dtype=np.float64)
        B = geometry.shape[0]
        assert geometry.shape == (B, 10), ValueError("geometry must have shape [B, 10] (Loop_pair.py:70-77)")
        sg = torch.as_tensor(np.asarray(sigma), dtype=torch.float64).to(device).contiguous()
        th = torch.as_tensor(np.asarray(thk), dtype=torch.float64).to(device).contiguous()
        nl = torch.as_tensor(np.broadcast_to(np.asarray(nlayers), (B,)).copy(), dtype=torch.int32).to(device)
        N = self.n_components * self.nwindows
        out = torch.empty((B, N), dtype=torch.float64, device=device)
        J = torch.empty((B, N, sg.shape[1]), dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            _lib.check(self._lib.gbp_tdem_fm_dlogc(self.ptr, B, geometry.ctypes.data_as(_lib.c_double_p), sg.shape[1], nl.data_ptr(),
                                                   sg.data_ptr(), th.data_ptr(), out.data_ptr(), J.data_ptr(),
                                                   torch.cuda.current_stream(device).cuda_stream))
        return out, J

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._lib.gbp_tdem_system_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


DEFAULT_TDEM_HANKEL_EPS = 1.0e-12      # relative to the inductive-limit value of every nodal sum (gbp_hankel_system_add_bins)


def _altitude_bins(heights):
    """(first altitude in metres, number of 1 m bins) covering the given altitudes (at most 1024 bins)."""
    h = np.asarray(heights, dtype=np.float64)
    lo = max(0, int(np.floor(np.nanmin(h)))) if h.size else 0
    hi = max(lo, int(np.floor(np.nanmax(h)))) if h.size else lo
    return lo, min(hi - lo + 1, 1024)


class _RawHandle:
    def __init__(self, npts, wmu, hd0, g, tables, eps=0.0, bins=None, more_sets=()):
        """``eps`` > 0 and ``bins`` = (first altitude, count): per-sounding abscissa windows in 1 m altitude bins, each nodal
        sum within ``eps`` times its inductive-limit value of the full sum (gbp_hankel_system_add_bins, relative budget).
        ``more_sets``: [(hd0, tables), ...] further table sets of the same layout (other transmitter-receiver offsets,
        gbp_hankel_system_add_set); the launches' ``set_of_row`` argument then says which set every row uses."""
        lib = _lib.load()
        self._lib = lib
        h = ctypes.c_void_p()
        dp = lambda x: x.ctypes.data_as(_lib.c_double_p)
        self._keep = (npts, wmu, hd0, g, tables)
        _lib.check(lib.gbp_hankel_system_create_raw(int(npts.size), npts.ctypes.data_as(_lib.c_int32_p), dp(wmu),
                                                    dp(hd0), dp(g), dp(tables), ctypes.byref(h)))
        self.ptr = h
        self.nF = int(npts.size)
        self.npoints = int(npts.sum())      # abscissa points of the full tables
        self.bins = None
        self.n_sets = 1 + len(more_sets)
        for hd0_k, tables_k in more_sets:
            hd0_k, tables_k = np.ascontiguousarray(hd0_k, dtype=np.float64), np.ascontiguousarray(tables_k, dtype=np.float64)
            assert tables_k.shape == tables.shape and hd0_k.shape == hd0.shape, ValueError("table sets must share one layout")
            _lib.check(lib.gbp_hankel_system_add_set(h, dp(hd0_k), dp(tables_k)))
        if eps > 0.0 and bins is not None:
            _lib.check(lib.gbp_hankel_system_add_bins(h, float(eps), 1, int(bins[0]), int(bins[1])))
            self.bins = (int(bins[0]), int(bins[1]))
        elif more_sets:
            _lib.check(lib.gbp_hankel_system_add_bins(h, 0.0, 1, 0, 0))      # descriptors of the sets' full tables only

    def bin_points(self, altitude):
        """Abscissa points a sounding at this altitude is evaluated with."""
        n = ctypes.c_int(0)
        _lib.check(self._lib.gbp_fdem_system_bin_points(self.ptr, int(np.floor(altitude)), ctypes.byref(n)))
        return n.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._lib.gbp_fdem_system_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _Mix:
    """Device side of a tdem_geometry.GeometryMix for one raw handle: gbp_td_mix + the rows' table sets."""

    def __init__(self, gm, device):
        self.gm = gm
        dev = lambda a, dt: torch.as_tensor(a, dtype=dt).to(device).contiguous()
        self.weights = dev(gm.weights, torch.float64)
        self.src, self.col = dev(gm.src, torch.int32), dev(gm.col, torch.int32)
        self.set_of_row = dev(gm.set_of_row, torch.int32) if gm.set_keys.shape[0] > 1 else None
        self.n_in, self.n_out = gm.n_in, gm.n_out

    def struct(self, weights=None):
        m = _lib.TdMix()
        m.n_in, m.terms, m.n_weights = self.n_in, self.src.shape[1], self.weights.shape[1]
        m.src, m.col = self.src.data_ptr(), self.col.data_ptr()
        m.weights = (self.weights if weights is None else weights).data_ptr()
        return m


def _raw_handle(systems, gm, eps, bins):
    """One raw Hankel handle for ``systems`` merged (all (system, basis integral, node) triples are its frequencies) with one
    table set per distinct (rho, dz) of the geometry mix ``gm``."""
    def merged(key):
        parts = [gm.tables(s_, key) for s_ in systems]
        cat = lambda j, ax=0: np.ascontiguousarray(np.concatenate([p_[j] for p_ in parts], axis=ax))
        return cat(0).astype(np.int32), cat(1), cat(2), cat(3), cat(4, 1)
    sets = [merged(k_) for k_ in gm.set_keys]
    return _RawHandle(*sets[0], eps=eps, bins=bins, more_sets=[(t_[2], t_[4]) for t_ in sets[1:]])


class TdemBatch:
    """B TDEM soundings on one GPU, all systems of a (multi-moment) acquisition in one object.

    ``systems``: list of TdemSystem (e.g. SkyTEM high and low moment); ``offset`` = (dx, dy, dz) of the receiver relative to
    the transmitter (Loop_pair, system/Loop_pair.py:63-77) and ``attitude`` = (tx roll, pitch, yaw, rx roll, pitch, yaw) in
    degrees in GA-AEM's convention (what Loop_pair.Geometry passes: roll, -pitch, -yaw; None = level flight): one tuple shared
    by the batch, or one per sounding ([B, 3] / [B, 6]).  All soundings run in the same launches whatever their geometry: a
    handle holds one table set per distinct (horizontal distance, dz) and a row's azimuth, attitude, output signs and scalings
    are its mixing weights (tdem_geometry.GeometryMix); only receivers exactly on the transmitter's axis (other filters) form a
    group of their own.  Channel layout of ``predicted``: system 0 components x, y, z (those it outputs), each over its
    windows, then system 1 ... (the reference's ``predicted_secondary_field`` layout).
    """

    def __init__(self, systems, nlayers, sigma, thk, height, offset, data=None, relative_error=None,
                 additive_error=None, device=None, hankel_eps=None, min_altitude=None, attitude=None):
        """``hankel_eps``: accuracy budget of the abscissa window every sounding is evaluated with -- the filter abscissae whose
        terms can add up to more than that fraction of a nodal sum's inductive-limit value at the sounding's OWN altitude (1 m
        bins, |rTE| <= 1; the same bound as FdemBatch's), so a sounding's numbers do not depend on its batch.
        Default 1e-12 (about half of the 120 / 140 abscissae at survey altitudes); 0: all abscissae.  ``min_altitude`` is
        accepted for compatibility and ignored."""
        if not torch.cuda.is_available():
            raise _lib.NativeLibraryError("TdemBatch needs a HIP device; there is no CPU fallback")
        self.systems = [systems] if isinstance(systems, TdemSystem) else list(systems)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._groups = None
        self.B, self.Lmax = np.asarray(sigma).shape
        h_all = np.broadcast_to(np.asarray(height, dtype=np.float64), (self.B,))
        geom = gaaem_geometry(h_all, offset, attitude)
        self.geometry = geom
        self.offset = tuple(float(v) for v in np.asarray(offset, dtype=np.float64)) if np.ndim(offset) == 1 else np.asarray(offset, dtype=np.float64)
        self.nChannels = sum(s.n_components * s.nwindows for s in self.systems)
        on = np.hypot(geom[:, 4], geom[:, 5]) == 0.0
        if on.any() and not on.all():      # receivers on and off the axis: other filters, other layout -> one child batch each
            sub = lambda a, m: None if a is None else np.asarray(a)[m]
            nl_all = np.broadcast_to(np.asarray(nlayers), (self.B,))
            self._groups = []
            for flag in (False, True):
                m = np.nonzero(on == flag)[0]
                child = TdemBatch(self.systems, nl_all[m], np.asarray(sigma)[m], np.asarray(thk)[m], h_all[m], geom[m, 4:7],
                                  data=sub(data, m), relative_error=sub(relative_error, m), additive_error=sub(additive_error, m),
                                  device=self.device, hankel_eps=hankel_eps, attitude=np.c_[geom[m, 1:4], geom[m, 7:10]])
                self._groups.append((torch.as_tensor(m, device=self.device), child))
            self.predicted = torch.empty((self.B, self.nChannels), dtype=torch.float64, device=self.device)
            self.chi2 = torch.empty(self.B, dtype=torch.float64, device=self.device)
            self.logL = torch.empty(self.B, dtype=torch.float64, device=self.device)
            return
        dev = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), dtype=dt).to(self.device).contiguous()
        self.sigma, self.thk = dev(sigma), dev(thk)
        self.nlayers = dev(np.broadcast_to(np.asarray(nlayers), (self.B,)), torch.int32)
        self.height = dev(h_all)
        self._h, self._W, self._nodal, self._mix = [], [], [], []
        self.hankel_eps = DEFAULT_TDEM_HANKEL_EPS if hankel_eps is None else float(hankel_eps)
        bins = _altitude_bins(h_all) if self.B > 0 else None
        with torch.cuda.device(self.device):
            for s in self.systems:
                gm = GeometryMix([s], geom)
                self._h.append(_raw_handle([s], gm, self.hankel_eps, bins))
                self._mix.append(_Mix(gm, self.device))
                n = s.node_frequencies().size
                W = s.time_operator()
                # block-diagonal over components: nodal layout is [Re(comp0 nodes), Re(comp1 nodes), Im(...), Im(...)]
                nc = s.n_components
                Wb = np.zeros((2 * nc * n, nc * s.nwindows))
                for c in range(nc):
                    Wb[c * n:(c + 1) * n, c * s.nwindows:(c + 1) * s.nwindows] = W[:n]
                    Wb[nc * n + c * n: nc * n + (c + 1) * n, c * s.nwindows:(c + 1) * s.nwindows] = W[n:]
                self._W.append(dev(Wb))
                self._nodal.append(torch.empty((self.B, gm.n_in), dtype=torch.float64, device=self.device))
        self.predicted = torch.empty((self.B, self.nChannels), dtype=torch.float64, device=self.device)
        # window blocks of the systems (k_td_apply writes dense [B, n_windows] rows; several systems are then laid side by side)
        self._win, c0 = {}, 0
        for s in self.systems:
            n = s.n_components * s.nwindows
            if len(self.systems) > 1:
                self._win[c0] = torch.empty((self.B, n), dtype=torch.float64, device=self.device)
            c0 += n
        self.data = None if data is None else dev(data)
        self.relative_error = None if relative_error is None else dev(relative_error)
        self.additive_error = None if additive_error is None else dev(additive_error)
        self.chi2 = torch.empty(self.B, dtype=torch.float64, device=self.device)
        self.logL = torch.empty(self.B, dtype=torch.float64, device=self.device)
        self._max_layers = None

    def primary_field(self):
        """[B, sum of the systems' components] free-space field at the receiver, reference convention (GeometryMix.primary_field)."""
        return GeometryMix(self.systems, self.geometry).primary_field()

    def forward(self):
        """predicted[B, nChannels]: frequency-domain HIP kernel per system (nodal spectra of the basis integrals), then geometry
        mixing + the window operator (k_td_apply)."""
        if self._groups is not None:
            for rows, child in self._groups:
                self.predicted[rows] = child.forward()
            return self.predicted
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        col = 0
        with torch.cuda.device(self.device):
            for s, h, W, nodal, mix in zip(self.systems, self._h, self._W, self._nodal, self._mix):
                rows = None if mix.set_of_row is None else mix.set_of_row.data_ptr()
                _lib.check(lib.gbp_fdem_forward_rows_ex(h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(),
                                                        self.sigma.data_ptr(), self.thk.data_ptr(), self.height.data_ptr(),
                                                        nodal.data_ptr(), rows, 0, stream))
                n = W.shape[1]
                out = self._win[col] if len(self.systems) > 1 else self.predicted
                mx = mix.struct()
                _lib.check(lib.gbp_td_apply_mix(self.B, self.Lmax, W.shape[0], n, self.nlayers.data_ptr(), W.data_ptr(), nodal.data_ptr(),
                                                None, out.data_ptr(), None, ctypes.byref(mx), stream))
                if out is not self.predicted:
                    self.predicted[:, col:col + n] = out
                col += n
        return self.predicted

    def sensitivity(self):
        """J[B, nChannels, Lmax] = d predicted / d ln(sigma) (the reference obtains it from gatdaem1d's
        derivative call, TD/tdem1d.py:98-154): exact frequency-domain Jacobian of the nodal values (Jacobian
        kernel on the raw handle) pushed through the same geometry mixing and linear time-domain operator (k_td_apply)."""
        return self.fm_dlogc()[1]

    def fm_dlogc(self):
        """(predicted, J) from one pass (TdemDataPoint.fm_dlogc, data/datapoint/TdemDataPoint.py:1031-1055)."""
        if self._groups is not None:
            out = torch.empty((self.B, self.nChannels, self.Lmax), dtype=torch.float64, device=self.device)
            for rows, child in self._groups:
                p, J = child.fm_dlogc()
                out[rows], self.predicted[rows] = J, p
            return self.predicted, out
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        out = torch.empty((self.B, self.nChannels, self.Lmax), dtype=torch.float64, device=self.device)
        if self._max_layers is None:
            self._max_layers = int(self.nlayers.max().item()) if self.B > 0 else 1
        col = 0
        with torch.cuda.device(self.device):
            for h, W, nodal, mix in zip(self._h, self._W, self._nodal, self._mix):
                Jn = torch.empty((self.B, mix.n_in, self.Lmax), dtype=torch.float64, device=self.device)
                rows = None if mix.set_of_row is None else mix.set_of_row.data_ptr()
                _lib.check(lib.gbp_fdem_fm_dlogc_rows_ex(h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(),
                                                         self.sigma.data_ptr(), self.thk.data_ptr(),
                                                         self.height.data_ptr(), nodal.data_ptr(), Jn.data_ptr(), self._max_layers, 1,
                                                         rows, 0, stream))
                n = W.shape[1]
                Jw = torch.empty((self.B, n, self.Lmax), dtype=torch.float64, device=self.device)
                pw = torch.empty((self.B, n), dtype=torch.float64, device=self.device)
                mx = mix.struct()
                _lib.check(lib.gbp_td_apply_mix(self.B, self.Lmax, W.shape[0], n, self.nlayers.data_ptr(), W.data_ptr(), nodal.data_ptr(),
                                                Jn.data_ptr(), pw.data_ptr(), Jw.data_ptr(), ctypes.byref(mx), stream))
                out[:, col:col + n, :] = Jw
                self.predicted[:, col:col + n] = pw
                col += n
        return self.predicted, out

    def std(self):
        """TdemDataPoint.std (data/datapoint/TdemDataPoint.py:361-365):
        sigma_i^2 = (rel_{sys,comp} * d_i)^2 + (add_sys * sqrt(1e-3 / t_i))^2 -- ``relative_error[B, sum_i n_components_i]``
        holds one level per (system, component) in the reference's order ``(i * n_components) + j``; ``additive_error[B,
        n_systems]`` one per system.  A ``relative_error`` with one column per SYSTEM is accepted for single-component
        systems only (where the two layouts coincide)."""
        if self._groups is not None:
            out = torch.empty((self.B, self.nChannels), dtype=torch.float64, device=self.device)
            for rows, child in self._groups:
                out[rows] = child.std()
            return out
        n_groups = sum(s.n_components for s in self.systems)
        assert self.relative_error.shape[1] == n_groups, ValueError(
            "relative_error needs one level per (system, component): {} columns, got {}".format(n_groups, self.relative_error.shape[1]))
        assert self.additive_error.shape[1] == len(self.systems), ValueError("additive_error needs one level per system")
        out = torch.empty_like(self.data)
        col = g = 0
        for i, s in enumerate(self.systems):
            t = torch.as_tensor(s.off_time, dtype=torch.float64, device=self.device)
            add = self.additive_error[:, i:i + 1] * torch.sqrt(1e-3 / t)
            for j in range(s.n_components):
                rel = self.relative_error[:, g:g + 1]
                out[:, col:col + s.nwindows] = torch.sqrt((rel * self.data[:, col:col + s.nwindows]) ** 2 + add ** 2)
                col += s.nwindows
                g += 1
        return out

    def forward_loglike(self):
        """forward + chi^2 + log-likelihood (DataPoint.data_misfit / likelihood with the TDEM error model)."""
        if self._groups is not None:
            for rows, child in self._groups:
                c2, ll = child.forward_loglike()
                self.chi2[rows], self.logL[rows], self.predicted[rows] = c2, ll, child.predicted
            return self.chi2, self.logL
        self.forward()
        sd = self.std()
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_gauss_loglike_std(self.B, self.nChannels, self.predicted.data_ptr(),
                                                 self.data.data_ptr(), sd.data_ptr(), self.chi2.data_ptr(),
                                                 self.logL.data_ptr(),
                                                 torch.cuda.current_stream(self.device).cuda_stream))
        return self.chi2, self.logL


class TdemDeviceChains(DeviceChains):
    """Device-resident rjMCMC (rjmcmc_gpu.DeviceChains) for time-domain soundings: one system or the systems of a
    multi-moment acquisition (e.g. SkyTEM high + low moment).  ``offset``: the transmitter-receiver offset (dx, dy, dz) of the
    block, or one per sounding [B, 3]; ``attitude``: (tx roll, pitch, yaw, rx roll, pitch, yaw) in GA-AEM's convention, one for
    the block or one per sounding [B, 6] (None: level flight).  The handle holds one set of Hankel tables per distinct
    (horizontal distance, dz) and every chain carries its table set and its geometry-mixing weights
    (gbp_td_operator.table_set / .mix): soundings of different geometry advance in the same launches.

    The systems' spline nodes are merged into ONE frequency-domain handle (all (system, basis integral, node) triples are
    "frequencies" of the sampler's forward / Jacobian launches) and ``gbp_rj_run_td`` turns the nodal spectra -- and their
    Jacobians -- into the windows of all systems with one block matrix W.  Error model: TdemDataPoint.std
    (data/datapoint/TdemDataPoint.py:361-365): a relative level per (system, component), an additive level per system
    scaled by sqrt(1e-3 / t) per gate; the error options may be scalars or lists per level like in the reference's
    skytem / tempest options files.  Channel layout = TdemBatch's (system 0: components x, y, z windows, system 1 ...).
    The Jacobian is the exact derivative (GA-AEM's is, too).  Parity is unpinned like the rest of the TDEM path (DESIGN.md
    3.7); the sampler logic itself is the FDEM-pinned one."""

    def __init__(self, systems, heights, data, offset, attitude=None, channel_additive=None, primary_field=None, **kw):
        """``channel_additive`` [N] + ``primary_field`` [B, components] (or True): the Tempest data point's model
        (data/datapoint/Tempest_datapoint.py:106-123, 161-176) -- ``data`` then holds TOTAL fields (secondary + the component's
        primary), the predicted primary field of each row's geometry is added to the predicted windows, the additive error of a
        channel is ``channel_additive`` times a sampled multiplier per component (options initial / minimum / maximum
        _additive_error = the multiplier's), the relative error multiplies the total field."""
        systems = [systems] if isinstance(systems, TdemSystem) else list(systems)
        assert all(isinstance(s, TdemSystem) for s in systems), TypeError("systems must be geobipy_amd.TdemSystem objects")
        if kw.get("ignore_likelihood"):
            raise NotImplementedError("ignore_likelihood (prior-only sampling) is taken by the frequency-domain sampler only (DeviceChains)")
        heights = np.atleast_1d(np.asarray(heights, dtype=np.float64))
        # sampled attitude angles (solve_transmitter_pitch / _roll / _yaw, solve_receiver_pitch / _roll / _yaw: gbp_td_moves)
        self._moves = device_angle_moves(kw)
        self._pos_moves = []
        geom_rows = gaaem_geometry(heights, offset, attitude)
        force = ()
        if self._moves:
            on = np.hypot(geom_rows[:, 4], geom_rows[:, 5]) == 0.0      # (one basis layout and one gbp_td_moves.on_axis per block)
            if on.any() and not on.all():
                raise NotImplementedError("sampled attitude angles: a block holds receivers ON the transmitter's axis and off it -- other filters "
                                          "and another basis layout; invert the two groups as blocks of their own")
            on_axis = bool(on.size > 0 and on.all())
            loop = float(systems[0].loopRadius()) > 0.0
            # position moves (receiver offset, transmitter height): the chain keeps its table set and is evaluated at another distance /
            # height (gbp_td_moves.rho_scale); a changed azimuth turns the transmitter's moment in the pair's frame like a transmitter rotation
            self._pos_moves = [m_[0] for m_ in self._moves if m_[0] in ("dx", "dy", "dz", "tx_z")]
            lateral = any(n_ in ("dx", "dy") for n_ in self._pos_moves)
            if lateral and loop:
                raise NotImplementedError("solve_receiver_x / _y on a system with a transmitter LOOP (ModellingLoopRadius): the loop's source term "
                                          "lam J1(lam a) does not scale with the distance; the device sampler moves the receiver laterally for "
                                          "dipole systems (Tempest), the host sampler for any")
            if lateral and on_axis:
                raise NotImplementedError("solve_receiver_x / _y for a receiver on the transmitter's axis (other filters off the axis)")
            if lateral:                                        # the prior must keep the receiver off the axis (the distance scale is rho_set / rho)
                reach = float(np.sqrt(sum(m_[3] ** 2 for m_ in self._moves if m_[0] in ("dx", "dy"))))
                rho_min = float(np.hypot(geom_rows[:, 4], geom_rows[:, 5]).min())
                if not reach < 0.5 * rho_min:
                    raise ValueError("maximum_receiver_x / _y_change ({:.3g} m) must stay below half the smallest horizontal transmitter-receiver "
                                     "distance of the block ({:.3g} m)".format(reach, rho_min))
            tx_moves = any(m_[0].startswith("tx_") and m_[0] != "tx_z" for m_ in self._moves) or lateral
            force = ((0, 2) if tx_moves else (0,)) if on_axis else (((0, 1, 2, 3, 4) if loop else (0, 1, 4)) if tx_moves else (0, 1))
        gm = GeometryMix(systems, geom_rows, force_basis=force)
        self.td_systems, self._gm = systems, gm
        nf = [s.n_components * s.node_frequencies().size for s in systems]          # output "frequencies" per system
        nw = [s.n_components * s.nwindows for s in systems]
        nF, N = sum(nf), sum(nw)
        assert gm.n_in // 2 <= 128 and nF <= 128, ValueError(
            "the merged systems have {} spline nodes x basis integrals (limit 128)".format(gm.n_in // 2))
        Wm = np.zeros((2 * nF, N))
        rel_group, add_group, add_scale = [], [], []
        f0 = c0 = g0 = 0
        for i, s in enumerate(systems):
            nc, n = s.n_components, s.node_frequencies().size
            W = s.time_operator()
            for c in range(nc):          # block-diagonal over components; nodal layout [Re(all frequencies) | Im(all frequencies)]
                cols = slice(c0 + c * s.nwindows, c0 + (c + 1) * s.nwindows)
                Wm[f0 + c * n:f0 + (c + 1) * n, cols] = W[:n]
                Wm[nF + f0 + c * n:nF + f0 + (c + 1) * n, cols] = W[n:]
                rel_group += [g0 + c] * s.nwindows
            add_group += [i] * nw[i]
            add_scale += list(np.sqrt(1e-3 / np.tile(s.off_time, nc)))
            f0, c0, g0 = f0 + nf[i], c0 + nw[i], g0 + nc
        if channel_additive is not None:              # Tempest: per-channel additive errors x a multiplier per (system, component)
            add_group, add_scale = list(rel_group), list(np.broadcast_to(np.asarray(channel_additive, dtype=np.float64), (N,)))
            # the multipliers as the reference samples them (independence proposal about the initial values, one draw, no prior term:
            # Tempest_datapoint.py:339-341, 475-487, 503-508; gbp_rj_options.additive_independent) unless the caller wants them as a
            # random walk inside their log-uniform prior like every other level (reference_multipliers=False)
            kw.setdefault("additive_independent", bool(kw.pop("reference_multipliers", True)))
        kw.pop("reference_multipliers", None)
        self._pred_offset0 = None
        if primary_field is not None:                 # total-field channels: predicted primary of every row's geometry, per window
            pp = gm.primary_field()                                                  # [B, sum of the systems' components]
            reps = np.concatenate([[s.nwindows] * s.n_components for s in systems])
            self._pred_offset0 = np.repeat(pp, reps, axis=1)
        self._W_host, self._td_struct = Wm, None
        self._row_index = None
        outer = self

        class _Handle:                    # what DeviceChains asks of an acquisition system
            def handle(self_inner):
                if getattr(outer, "_raw", None) is None:
                    reach = sum(m_[3] * (0.5 if m_[0] == "dz" else 1.0) for m_ in outer._moves if m_[0] in ("dz", "tx_z"))   # effective heights
                    outer._raw = _raw_handle(systems, gm, outer._hankel_eps, _altitude_bins(np.r_[heights - reach, heights + reach] if reach else heights))
                return outer._raw
        if kw.get("solve_z"):
            raise NotImplementedError("solve_z on time-domain chains: the reference's forward takes the transmitter's z (system/Loop_pair.py:70), "
                                      "which the data point's z move never touches; the loop pair's geometry is not sampled")
        kw.pop("exact_jacobian", None)
        kw.pop("hankel_eps_ppm", None)
        # per-chain abscissa window (1 m altitude bins): hankel_eps relative to the inductive-limit value of every nodal sum;
        # default 1e-12, 0 = all abscissae (min_altitude: accepted and ignored)
        eps = kw.pop("hankel_eps", None)
        self._hankel_eps = DEFAULT_TDEM_HANKEL_EPS if eps is None else float(eps)
        kw.pop("min_altitude", None)
        dev = kw.get("device")
        dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
        self._mix = _Mix(gm, dev)
        extra = -sum(np.log(2.0 * m_[3]) for m_ in self._moves)                      # densities of the angles' uniform priors
        super().__init__(_Handle(), heights, data, exact_jacobian=True, add_scale=np.asarray(add_scale),
                         rel_group=np.asarray(rel_group, dtype=np.int32), add_group=np.asarray(add_group, dtype=np.int32),
                         extra_log_prior=extra, **kw)
        # carried with the chains' rows (infer() re-packs every tensor of self.t)
        self.t["mix_w"] = self._mix.weights
        if self._mix.set_of_row is not None:
            self.t["geom_id"] = self._mix.set_of_row
        if self._moves:
            B = heights.size
            f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev).contiguous()
            self.t.update(geom=f64(geom_rows), geom_p=f64(geom_rows), geom0=f64(geom_rows), mix_w_p=self._mix.weights.clone(), mix_w0=self._mix.weights.clone(),
                          geom_hist=torch.zeros((B, len(self._moves), 199), dtype=torch.int32, device=dev), best_geom=f64(geom_rows))
            if "pred_offset" in self.t:
                self.t.update(pred_offset_p=self.t["pred_offset"].clone(), pred_offset0=self.t["pred_offset"].clone())
            if self._pos_moves:
                self.t.update(rho_scale=torch.ones(B, dtype=torch.float64, device=dev), rho_scale_p=torch.ones(B, dtype=torch.float64, device=dev),
                              rho_set=f64(np.hypot(geom_rows[:, 4], geom_rows[:, 5])), dz_set=f64(geom_rows[:, 6]),
                              height=self.t["height"].clone(), height_p=self.t["height"].clone(), height_set=self.t["height"].clone())
                self._bind()
            i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(dev).contiguous()
            self._mv_layout = dict(comp=i32(gm.block_comp), scale=f64(gm.block_scale), primary=f64(gm.block_primary), windows=i32(gm.block_windows))
            self._td_struct = None

    def _bind(self):
        if self._pred_offset0 is not None and "pred_offset" not in self.t:    # (before the first launch: _initialize evaluates with it)
            self.t["pred_offset"] = torch.as_tensor(self._pred_offset0, dtype=torch.float64).to(self.device).contiguous()
        super()._bind()

    def predicted_primary(self):
        """[B, components] predicted primary field of every row's geometry (reference convention), or None."""
        return None if self._pred_offset0 is None else self._gm.primary_field()

    def _set_row_map(self, index):
        self._row_index = index
        self._td_struct = None            # pointers of the rows the next launches evaluate

    def _rows(self, name, default):
        t = self.__dict__.get("t")
        v = t[name] if t is not None and name in t else default
        return v if (v is None or self._row_index is None) else v[self._row_index].contiguous()

    def _td(self):
        if self._td_struct is None:
            dev = self.device
            if getattr(self, "_W", None) is None:
                self._W = torch.as_tensor(self._W_host, dtype=torch.float64).to(dev).contiguous()
            rows = self.t["k"].shape[0]          # scratch of the sampler's launches: the block's rows
            n_in = self._mix.n_in
            if getattr(self, "_nodal", None) is None or self._nodal.shape[0] < rows:
                self._nodal = torch.empty((rows, n_in), dtype=torch.float64, device=dev)
                self._J_nodal = torch.empty((rows, n_in, self.K), dtype=torch.float64, device=dev)
            self._w_rows = self._rows("mix_w", self._mix.weights)
            self._set_rows = self._rows("geom_id", self._mix.set_of_row)
            self._off_rows = self._rows("pred_offset", None)
            td = _lib.TdOperator()
            td.n_nodal, td.W, td.nodal, td.J_nodal = self._W.shape[0], self._W.data_ptr(), self._nodal.data_ptr(), self._J_nodal.data_ptr()
            td.mix = self._mix.struct(self._w_rows)
            td.mix.offset = None if self._off_rows is None else self._off_rows.data_ptr()
            td.table_set = None if self._set_rows is None else self._set_rows.data_ptr()
            if self._moves and self._row_index is None and "geom" in self.t:      # (not during the initialisation's row maps)
                mv, t, lay, gm = td.moves, self.t, self._mv_layout, self._gm
                mv.n_moves = len(self._moves)
                for q, (_, e, sg, hw, sc, nb) in enumerate(self._moves):
                    mv.entry[q], mv.sign[q], mv.half_width[q], mv.scale[q], mv.n_bins[q] = e, sg, hw, sc, nb
                mv.geom, mv.geom_p, mv.geom0 = t["geom"].data_ptr(), t["geom_p"].data_ptr(), t["geom0"].data_ptr()
                mv.weights, mv.weights_p = self._w_rows.data_ptr(), t["mix_w_p"].data_ptr()
                if self._off_rows is not None:
                    mv.offset, mv.offset_p = self._off_rows.data_ptr(), t["pred_offset_p"].data_ptr()
                mv.hist, mv.best_geom = t["geom_hist"].data_ptr(), t["best_geom"].data_ptr()
                mv.n_blocks, mv.n_basis, mv.loop, mv.on_axis = int(gm.block_comp.size), len(gm.basis), int(gm.loop), int(gm.on_axis)
                for i_, b_ in enumerate(gm.basis):
                    mv.basis[i_] = int(b_)
                mv.block_comp, mv.block_scale = lay["comp"].data_ptr(), lay["scale"].data_ptr()
                mv.block_primary, mv.block_windows = lay["primary"].data_ptr(), lay["windows"].data_ptr()
                if self._pos_moves:
                    mv.rho_scale, mv.rho_scale_p = self._rows("rho_scale", None).data_ptr(), t["rho_scale_p"].data_ptr()
                    mv.rho_set, mv.dz_set = t["rho_set"].data_ptr(), t["dz_set"].data_ptr()
            self._td_struct = td
        return self._td_struct

    def _restart_more(self, r):
        if self._moves:
            t = self.t
            t["geom"][r] = t["geom0"][r]
            t["best_geom"][r] = t["geom0"][r]
            t["mix_w"][r] = t["mix_w0"][r]
            t["geom_hist"][r] = 0
            if "pred_offset0" in t:
                t["pred_offset"][r] = t["pred_offset0"][r]
            if self._pos_moves:
                t["rho_scale"][r] = 1.0
                t["height"][r] = t["height_set"][r]

    def sampled_angles(self, which="geom"):
        """{name: [B] values} of the sampled angles in the reference's own convention (the loops' pitch / roll / yaw) -- ``which``:
        "geom" the chains' current tuples, "best_geom" those of the highest-posterior states, "geom0" the measured ones."""
        g = self.t[which]
        return {name: (sg * g[:, e]).clone() for name, e, sg, _, _, _ in self._moves}

    def mix_for_geometry(self, geom_rows):
        """(mixing weights [B, n_w] on the device, predicted-primary offset [B, N] on the device or None, primary field [B, components]
        numpy or None) of the GA-AEM tuples ``geom_rows`` [B, 10] in THIS block's basis layout -- e.g. of the highest-posterior
        attitudes (``t["best_geom"]``) when the results containers are filled: rotations change neither the table sets nor the
        layout (the layout keeps every basis integral a sampled rotation can switch on), only the weights and the primary field."""
        g = geom_rows.detach().cpu().numpy() if torch.is_tensor(geom_rows) else np.asarray(geom_rows, dtype=np.float64)
        gm = GeometryMix(self.td_systems, g, force_basis=tuple(self._gm.basis))
        assert gm.basis == self._gm.basis and gm.weights.shape[1] == self._mix.weights.shape[1], "a rotation cannot change the basis layout"
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device).contiguous()
        self._geometry_rows_extra = None
        if self._pos_moves:                  # a moved position: the chain's table set at another distance / effective height (gbp_td_moves)
            rho_set, dz_set = self.t["rho_set"].cpu().numpy(), self.t["dz_set"].cpu().numpy()
            self._geometry_rows_extra = dict(height=dev(g[:, 0] + 0.5 * (g[:, 6] - dz_set)), scale=dev(np.where(rho_set > 0.0, rho_set / np.maximum(np.hypot(g[:, 4], g[:, 5]), 1e-300), 1.0)))   # (on-axis sets: no distance to scale)
        if self._pred_offset0 is None:
            return dev(gm.weights), None, None
        pp = gm.primary_field()
        reps = np.concatenate([[s_.nwindows] * s_.n_components for s_ in self.td_systems])
        return dev(gm.weights), dev(np.repeat(pp, reps, axis=1)), pp

    def geometry_rows_extra(self):
        """{"height": effective heights, "scale": distance scales} of the tuples handed to the last ``mix_for_geometry`` call when
        positions are sampled (what ``_eval_loglike(height=..., scale=...)`` takes), else None."""
        return getattr(self, "_geometry_rows_extra", None)

    def _eval_loglike(self, k, sigma, thk, height, data, rel, add, pred, chi2, logl, weights=None, offset=None, scale=None):
        """``weights`` / ``offset``: per-row mixing weights and predicted-primary offsets to evaluate with instead of the chains'
        current ones (mix_for_geometry); ``scale``: per-row distance scale of the rows' table sets (sampled positions; ``height`` is
        then the effective height, geometry_rows_extra)."""
        td = self._td()
        lib, n = _lib.load(), k.numel()
        nodal = torch.empty((n, self._mix.n_in), dtype=torch.float64, device=self.device)
        if scale is None and self._pos_moves and self._row_index is None and "rho_scale" in self.t and n == self.t["rho_scale"].shape[0]:
            scale = self.t["rho_scale"]                      # the chains' current positions
        _lib.check(lib.gbp_fdem_forward_rows_scaled(self._h.ptr, n, self.K, k.data_ptr(), sigma.data_ptr(), thk.data_ptr(), height.data_ptr(),
                                                    nodal.data_ptr(), td.table_set, None if scale is None else scale.data_ptr(),
                                                    self.forward_waves, self._stream()))
        p = torch.empty((n, self._W.shape[1]), dtype=torch.float64, device=self.device)
        mix = td.mix
        if weights is not None or offset is not None:
            mix = self._mix.struct(self._w_rows if weights is None else weights)
            mix.offset = td.mix.offset if offset is None else offset.data_ptr()
        _lib.check(lib.gbp_td_apply_mix(n, self.K, self._W.shape[0], self._W.shape[1], k.data_ptr(), self._W.data_ptr(), nodal.data_ptr(),
                                        None, p.data_ptr(), None, ctypes.byref(mix), self._stream()))
        rg = self.t["rel_group"].long() if self.t["rel_group"] is not None else torch.zeros(data.shape[1], dtype=torch.long, device=self.device)
        ag = self.t["add_group"].long() if self.t["add_group"] is not None else torch.zeros(data.shape[1], dtype=torch.long, device=self.device)
        sd = torch.sqrt((rel[:, rg] * data) ** 2 + (add[:, ag] * self.t["add_scale"][None, :]) ** 2).contiguous()
        _lib.check(lib.gbp_gauss_loglike_std(n, p.shape[1], p.data_ptr(), data.data_ptr(), sd.data_ptr(), chi2.data_ptr(),
                                             logl.data_ptr(), self._stream()))
        if pred is not None:
            pred.copy_(p)

    def _eval_jacobian(self, k, sigma, thk, height, J, max_layers):
        td = self._td()
        n = k.numel()
        Jn = torch.empty((n, self._mix.n_in, self.K), dtype=torch.float64, device=self.device)
        nodal = torch.empty((n, self._mix.n_in), dtype=torch.float64, device=self.device)
        scale = self.t["rho_scale"] if (self._pos_moves and self._row_index is None and "rho_scale" in self.t and n == self.t["rho_scale"].shape[0]) else None
        _lib.check(_lib.load().gbp_fdem_fm_dlogc_rows_scaled(self._h.ptr, n, self.K, k.data_ptr(), sigma.data_ptr(), thk.data_ptr(),
                                                             height.data_ptr(), nodal.data_ptr(), Jn.data_ptr(), int(max_layers), 1,
                                                             td.table_set, None if scale is None else scale.data_ptr(), 0, self._stream()))
        p = torch.empty((n, self._W.shape[1]), dtype=torch.float64, device=self.device)
        _lib.check(_lib.load().gbp_td_apply_mix(n, self.K, self._W.shape[0], self._W.shape[1], k.data_ptr(), self._W.data_ptr(),
                                                nodal.data_ptr(), Jn.data_ptr(), p.data_ptr(), J.data_ptr(), ctypes.byref(td.mix), self._stream()))

    def _launch(self, n, accumulate):
        _lib.check(_lib.load().gbp_rj_run_td(self._h.ptr, self._td(), self._o, self._c, self.iteration, int(n), int(bool(accumulate)),
                                             self._stream()))


class TdemEngine:
    """forward(edges, values) / sensitivity(edges, values) of one sounding's geometry -- what the host sampler
    (inference.Inference1D, rjmcmc.accept_reject) asks of a data point -- on persistent TdemBatch objects (tables, windows and
    mixing weights are built once; a call refills the model tensors and launches)."""

    def __init__(self, systems, height, offset, attitude=None, lmax=32, hankel_eps=None, loop_pair=None, total_field=False):
        self.systems, self.height, self.offset, self.attitude = list(systems), float(height), tuple(offset), attitude
        self.lmax, self.hankel_eps = int(lmax), hankel_eps
        self.total_field = bool(total_field)      # Tempest: predictions = secondary + the free-space primary field of the geometry
        self._b = {}
        self.loop_pair = None if loop_pair is None else dict(loop_pair)      # tdem_geometry.loop_pair_values: base of geometry moves

    def _batch(self, models, geometry=None):
        """``geometry``: {name: value} overrides of the loop pair's scalars (a sampled geometry, rjmcmc.ScalarMove) -- a batch per
        distinct geometry is built and kept while it is the current or the proposed one (two are alive at a time)."""
        n = len(models)
        height, offset, attitude, key = self.height, self.offset, self.attitude, n
        if geometry:
            assert self.loop_pair is not None, ValueError("geometry moves need the engine's loop pair (TdemDataPoint.make_engine)")
            g = gaaem_tuple(dict(self.loop_pair, **geometry))
            height, offset, attitude = float(g[0]), tuple(g[4:7]), tuple(np.r_[g[1:4], g[7:10]])
            key = (n,) + tuple(float(x) for x in g)
            if key not in self._b and len(self._b) > 8:
                self._b = {k_: v_ for k_, v_ in list(self._b.items())[-4:]}
        b = self._b.get(key)
        if b is None:
            b = self._b[key] = TdemBatch(self.systems, np.ones(n, dtype=np.int32), np.ones((n, self.lmax)), np.zeros((n, self.lmax)),
                                         np.full(n, height), offset, attitude=attitude, hankel_eps=self.hankel_eps)
        nl = np.array([v.size for _, v in models], dtype=np.int32)
        assert nl.max() <= self.lmax, ValueError("model has more layers than the engine was sized for")
        sig, thk = np.ones((n, self.lmax)), np.zeros((n, self.lmax))
        for i, (e, v) in enumerate(models):
            sig[i, : v.size] = v
            thk[i, : v.size - 1] = np.diff(np.r_[0.0, e])
        b.sigma.copy_(torch.as_tensor(sig))
        b.thk.copy_(torch.as_tensor(thk))
        b.nlayers.copy_(torch.as_tensor(nl))
        b._max_layers = int(nl.max())
        return b, nl

    def forward_many(self, models, geometry=None):
        b = self._batch(models, geometry)[0]
        out = b.forward().cpu().numpy()
        if self.total_field:
            out = out + np.repeat(b.primary_field(), [s_.nwindows for s_ in self.systems for _ in range(s_.n_components)], axis=1)
        return out

    def forward(self, edges, values, geometry=None):
        return self.forward_many([(np.asarray(edges, dtype=np.float64), np.asarray(values, dtype=np.float64))], geometry)[0]

    def sensitivity(self, edges, values, geometry=None):
        b, nl = self._batch([(np.asarray(edges, dtype=np.float64), np.asarray(values, dtype=np.float64))], geometry)
        return b.sensitivity().cpu().numpy()[0][:, : nl[0]]

    def fm_dlogc(self, edges, values, geometry=None):
        b, nl = self._batch([(np.asarray(edges, dtype=np.float64), np.asarray(values, dtype=np.float64))], geometry)
        p, J = b.fm_dlogc()
        p = p.cpu().numpy()[0]
        if self.total_field:                 # like forward_many: Tempest predictions carry the primary field (which no layer moves)
            p = p + np.repeat(b.primary_field(), [s_.nwindows for s_ in self.systems for _ in range(s_.n_components)], axis=1)[0]
        return p, J.cpu().numpy()[0][:, : nl[0]]


class TdemDataPoint:
    """Per-sounding TDEM interface mirroring the members of the reference's ``TdemDataPoint``
    (data/datapoint/TdemDataPoint.py) the sampler touches: ``forward`` (:997-1022), ``sensitivity`` (:1024-1029),
    ``fm_dlogc`` (:1031-1055), ``std`` (:329-376), ``active``, ``deltaD``, ``data_misfit`` and ``likelihood``
    (DataPoint.py:491-525), ``perturb`` (:681), ``probability``, ``set_priors`` / ``set_proposals`` (:950-985).  ``system``
    is a list of TdemSystem (or .stm paths); the receiver offset and both loops' attitude come from the loop pair (offset =
    receiver - transmitter; angles handed on as Loop_pair.Geometry does, Loop_pair.py:70-77).  Every evaluation is a B = 1
    launch of the batched GPU path on a persistent engine (TdemEngine).  Geometry moves (solve_transmitter_* / solve_receiver_*,
    all False in the reference's options files) are not proposed."""

    kind = "tdem"          # which of the reference's container layouts hdf.create_inference1d builds for it

    def __init__(self, x=0.0, y=0.0, z=0.0, elevation=0.0, data=None, std=None, predictedData=None, system=None,
                 transmitter_loop=None, receiver_loop=None, lineNumber=0.0, fiducial=0.0):
        if isinstance(system, (str, TdemSystem)):
            system = [system]
        self.system = [TdemSystem(s) if isinstance(s, str) else s for s in system]
        self.x, self.y, self.elevation = np.float64(x), np.float64(y), np.float64(elevation)
        self.z = np.atleast_1d(np.asarray(z, dtype=np.float64)).copy()
        self.lineNumber, self.fiducial = lineNumber, fiducial
        self.transmitter, self.receiver = transmitter_loop, receiver_loop
        n = self.nChannels
        self._data = np.zeros(n) if data is None else np.asarray(data, dtype=np.float64).copy()
        self._predictedData = np.zeros(n) if predictedData is None else np.asarray(predictedData, np.float64).copy()
        self._relative_error = np.full(self.n_error_groups, 0.01)      # one per (system, component), TdemDataPoint.py:362
        self._additive_error = np.zeros(self.nSystems)
        self.units = r"$\\frac{V}{m^{2}}$"
        self._sensitivity_matrix = None
        self._rel_prior = self._add_prior = None              # rjmcmc.ErrorPrior (vector) once set_priors ran
        self._prng = None
        self._engine, self._engine_key = None, None
        self._geom_moves = []                                 # rjmcmc.ScalarMove list once set_priors saw solve_transmitter_* / solve_receiver_*
        self.engine = None                                    # TEST HOOK only (see FdemDataPoint.engine); the product never sets it

    @property
    def offset(self):
        return from_loops(self.transmitter, self.receiver)[0]

    @property
    def attitude(self):
        """(tx roll, -tx pitch, -tx yaw, rx roll, -rx pitch, -rx yaw): what Loop_pair.Geometry hands GA-AEM (Loop_pair.py:70-77)."""
        return from_loops(self.transmitter, self.receiver)[1]

    @property
    def nSystems(self):
        return len(self.system)

    @property
    def nChannels(self):
        return int(sum(s.n_components * s.nwindows for s in self.system))

    @property
    def n_error_groups(self):
        return int(sum(s.n_components for s in self.system))

    @property
    def data(self):
        return self._data

    @property
    def predictedData(self):
        return self._predictedData

    predicted_secondary_field = predictedData

    @property
    def relative_error(self):
        return self._relative_error

    @relative_error.setter
    def relative_error(self, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        assert v.size == self.n_error_groups and np.all(v > 0.0), ValueError(
            "relative_error must be > 0, one per (system, component) in the order (i * n_components) + j")
        self._relative_error = v.copy()

    @property
    def additive_error(self):
        return self._additive_error

    @additive_error.setter
    def additive_error(self, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        assert v.size == self.nSystems, ValueError("additive_error must have one value per system")
        self._additive_error = v.copy()

    @property
    def active(self):
        d = np.array(self.data, dtype=np.float64)              # (Tempest: the TOTAL field, EmDataPoint.active :45-56)
        d[d <= 0.0] = np.nan
        return ~np.isnan(d)

    @property
    def deltaD(self):
        return self._predictedData - self.data

    # -- engine -------------------------------------------------------------------------------------------------------
    def make_engine(self, lmax=32, hankel_eps=None):
        """A TdemEngine for this data point's altitude and geometry (what Inference1D.initialize asks for)."""
        return TdemEngine(self.system, self.z[0], self.offset, self.attitude, lmax=lmax, hankel_eps=hankel_eps,
                          loop_pair=loop_pair_values(self.transmitter, self.receiver))

    def geometry_moves(self, **options):
        """The loop pair's sampled scalars as rjmcmc.ScalarMove objects, in the reference's order (Loop_pair.set_priors /
        set_proposals / perturb, system/Loop_pair.py:161-192), from the options' ``solve_transmitter_* / solve_receiver_*`` keys."""
        return loop_pair_moves(loop_pair_values(self.transmitter, self.receiver), options)

    def _eng(self, n_layers):
        if self.engine is not None:
            return self.engine
        key = (float(self.z[0]), self.offset, self.attitude)
        if self._engine is None or self._engine_key != key or self._engine.lmax < n_layers:
            self._engine, self._engine_key = self.make_engine(lmax=max(32, int(n_layers))), key
        return self._engine

    def error_model(self, reference_fm_dlogc=True):
        """rjmcmc.ErrorModel of TdemDataPoint.std: channel -> (system, component) relative level, channel -> system additive
        level, sqrt(1e-3 / t) per gate.  ``reference_fm_dlogc``: the host sampler forms its stochastic-Newton gradient as the
        reference does for time-domain data -- with the prediction of the current model next to the Jacobian of the remapped
        one (TdemDataPoint.fm_dlogc :1031-1055 keeps only the Jacobian); False uses the remapped model's own prediction, as the
        device sampler does (both are valid proposals: the acceptance ratio uses the density actually drawn from)."""
        from . import rjmcmc
        rg, ag, sc, g = [], [], [], 0
        for i, s_ in enumerate(self.system):
            for _ in range(s_.n_components):
                rg += [g] * s_.nwindows
                ag += [i] * s_.nwindows
                sc += list(np.sqrt(1e-3 / s_.off_time))
                g += 1
        return rjmcmc.ErrorModel(rg, ag, sc, stale_prediction=reference_fm_dlogc)

    @staticmethod
    def _model_arrays(mod):
        assert np.isinf(mod.mesh.edges[-1]), ValueError("mod.edges must have last entry be infinity")
        return np.asarray(mod.mesh.edges[1:-1], dtype=np.float64), np.asarray(mod.values, dtype=np.float64)

    @property
    def std(self):
        return self.error_model().std(self._data, self._relative_error, self._additive_error)

    @property
    def sensitivity_matrix(self):
        return self._sensitivity_matrix

    def forward(self, mod):
        """TdemDataPoint.forward (:997-1022): predicted_secondary_field of every system, components x, y, z, z negated."""
        e, v = self._model_arrays(mod)
        assert self.z[0] >= mod.mesh.relative_to, "Sensor altitude must be above the top of the model"      # TD/tdem1d.py:28
        self._predictedData[:] = self._eng(v.size).forward(e, v)

    def sensitivity(self, mod, ix=None, model_changed=False):
        """J[nChannels, nLayers] = d predicted / d ln(sigma) (TdemDataPoint.sensitivity :1024-1029 -> gaTdem1dsen,
        TD/tdem1d.py:125-154: gatdaem1d's conductivity derivative x sigma); ``ix``: the layers wanted (default all)."""
        e, v = self._model_arrays(mod)
        J = np.asarray(self._eng(v.size).sensitivity(e, v))
        self._sensitivity_matrix = J if ix is None else J[:, np.asarray(ix)]
        return self._sensitivity_matrix

    def fm_dlogc(self, mod):
        """Prediction and Jacobian from one pass (TdemDataPoint.fm_dlogc :1031-1055 -> ga_fm_dlogc, TD/tdem1d.py:98-123).  The
        reference keeps only the Jacobian from that call; here the prediction of the same pass is stored as well."""
        e, v = self._model_arrays(mod)
        eng = self._eng(v.size)
        if hasattr(eng, "fm_dlogc"):
            self._predictedData[:], self._sensitivity_matrix = eng.fm_dlogc(e, v)
        else:
            self._predictedData[:], self._sensitivity_matrix = eng.forward(e, v), np.asarray(eng.sensitivity(e, v))

    def _loglike(self):
        if self.engine is not None:                      # CPU test tier
            from . import rjmcmc
            return rjmcmc.gauss_loglike(self._predictedData, self.data, self.std)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)[None, :]).to(dev)
        p, d, sd = t(self._predictedData), t(self.data), t(self.std)      # (self.data: Tempest's channels hold the total field)
        out = torch.empty(2, dtype=torch.float64, device=dev)
        _lib.check(_lib.load().gbp_gauss_loglike_std(1, self.nChannels, p.data_ptr(), d.data_ptr(), sd.data_ptr(), out.data_ptr(),
                                                     out.data_ptr() + 8, torch.cuda.current_stream(dev).cuda_stream))
        o = out.cpu().numpy()
        return float(o[0]), float(o[1])

    def data_misfit(self):
        return np.float64(self._loglike()[0])

    def likelihood(self, log):
        ll = self._loglike()[1]
        return np.float64(ll) if log else np.float64(np.exp(ll))

    def find_best_halfspace(self, minConductivity=1e-4, maxConductivity=1e4, nSamples=100):
        """Half-space Model that best fits the data: brute-force search over ``nSamples`` log-spaced conductivities
        (EmDataPoint.find_best_halfspace, data/datapoint/EmDataPoint.py:148-186) -- ONE launch over the trial half-spaces here.
        Like the reference's loop, the search leaves ``predictedData`` at the LAST trial (maxConductivity), not at the best one
        (EmDataPoint.py:176-183: ``data_misfit()`` right after it is the misfit of that last half-space -- the number the
        reference's gallery prints, tests/golden/make_tdem_doc_pins.py)."""
        from . import rjmcmc
        from .model import Model, RectilinearMesh1D
        assert maxConductivity > minConductivity, ValueError("Maximum conductivity must be greater than the minimum")
        c = np.logspace(np.log10(minConductivity), np.log10(maxConductivity), nSamples)
        none = np.zeros(0)
        eng = self._eng(1)
        models = [(none, np.array([ci])) for ci in c]
        preds = eng.forward_many(models) if hasattr(eng, "forward_many") else np.stack([eng.forward(e, v) for e, v in models])
        data, std = np.asarray(self.data), self.std
        phi = [rjmcmc.gauss_loglike(p_, data, std)[0] for p_ in preds]
        self._predictedData[:] = preds[-1]
        return Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, np.inf]), values=np.array([c[int(np.argmin(phi))]]))

    # -- rjMCMC members (DataPoint.py:454-489, 531-644; TdemDataPoint.py:681, 950-985): error-level priors, proposals, moves -----
    def set_priors(self, relative_error_prior=None, additive_error_prior=None, data_prior=None, **kwargs):
        """TdemDataPoint.set_priors (:950-971) -> DataPoint.set_priors (:575-595): log-uniform priors [minimum, maximum] on the
        error levels that are solved for -- lists per level in the time-domain options files (minimum_relative_error = [..] per
        system x component, minimum_additive_error = [..] per system); scalars are broadcast.  ``prng`` is remembered for
        perturb().  Transmitter / receiver priors (solve_transmitter_* / solve_receiver_*: Loop_pair.set_priors, system/Loop_pair.py
        :166-178) become rjmcmc.ScalarMove objects centred on the loops' current values; their proposal scales come with set_proposals."""
        from . import rjmcmc
        if any((k_.startswith("solve_transmitter_") or k_.startswith("solve_receiver_")) and v_ for k_, v_ in kwargs.items()):
            stems = {stem for _, stem, _ in LOOP_PAIR_SCALARS if kwargs.get("solve_" + stem, False)}
            self._geom_moves = loop_pair_moves(loop_pair_values(self.transmitter, self.receiver),
                                               dict({st + "_proposal_variance": 0.0 for st in stems}, **kwargs))
        self._prng = kwargs.get("prng", self._prng)
        if relative_error_prior is None and kwargs.get("solve_relative_error", False):
            relative_error_prior = (kwargs["minimum_relative_error"], kwargs["maximum_relative_error"])
        if additive_error_prior is None and kwargs.get("solve_additive_error", False):
            additive_error_prior = (kwargs["minimum_additive_error"], kwargs["maximum_additive_error"])
        var = lambda p_: p_.var if p_ is not None else 0.0
        vec = lambda v_, n_: np.broadcast_to(np.asarray(v_, dtype=np.float64), (n_,)).copy()
        if relative_error_prior is not None:
            self._rel_prior = rjmcmc.ErrorPrior(vec(relative_error_prior[0], self.n_error_groups), vec(relative_error_prior[1], self.n_error_groups),
                                                var(self._rel_prior))
        if additive_error_prior is not None:
            self._add_prior = rjmcmc.ErrorPrior(vec(additive_error_prior[0], self.nSystems), vec(additive_error_prior[1], self.nSystems),
                                                var(self._add_prior))

    def set_proposals(self, relative_error_proposal=None, additive_error_proposal=None, **kwargs):
        """TdemDataPoint.set_proposals (:973-985) -> DataPoint.set_proposals (:597-644): log-normal random walks with the
        options file's proposal variances (a list per level, or a scalar); Loop_pair.set_proposals (:180-192) for the loop pair's moves."""
        self._prng = kwargs.get("prng", self._prng)
        stem_of = {name: stem for name, stem, _ in LOOP_PAIR_SCALARS}
        for m_ in self._geom_moves:
            if stem_of[m_.name] + "_proposal_variance" in kwargs:
                m_.scale = float(kwargs[stem_of[m_.name] + "_proposal_variance"])
        if relative_error_proposal is None and kwargs.get("solve_relative_error", False):
            relative_error_proposal = kwargs["relative_error_proposal_variance"]
        if additive_error_proposal is None and kwargs.get("solve_additive_error", False):
            additive_error_proposal = kwargs["additive_error_proposal_variance"]
        for prior, v_, n_ in ((self._rel_prior, relative_error_proposal, self.n_error_groups), (self._add_prior, additive_error_proposal, self.nSystems)):
            if v_ is not None:
                assert prior is not None, ValueError("set_priors must come before set_proposals")
                prior.var = np.broadcast_to(np.asarray(v_, dtype=np.float64), (n_,)).copy()

    def perturb(self):
        """TdemDataPoint.perturb (:681) -> DataPoint.perturb (:531-573): the relative levels, then the additive levels -- each
        set proposed jointly and redrawn while outside its prior (the current values are kept at the 10th redraw)."""
        if self._rel_prior is not None and np.any(np.asarray(self._rel_prior.var) > 0.0):
            self._relative_error = np.atleast_1d(self._rel_prior.propose(self._prng, self._relative_error))
        if self._add_prior is not None and np.any(np.asarray(self._add_prior.var) > 0.0):
            self._additive_error = np.atleast_1d(self._add_prior.propose(self._prng, self._additive_error))
        if self._geom_moves:                                  # self.loop_pair.perturb() (TdemDataPoint.py:683)
            cur = loop_pair_values(self.transmitter, self.receiver)
            self.set_loop_pair({m_.name: m_.propose(self._prng, cur[m_.name]) for m_ in self._geom_moves})

    def set_loop_pair(self, values):
        """Write sampled scalars {name: value} (tdem_geometry.LOOP_PAIR_SCALARS) back into the two loops; an offset component moves
        the receiver (the pair's offset is receiver - transmitter)."""
        tx, rx = self.transmitter, self.receiver
        put = lambda loop, attr, v_: setattr(loop, attr, np.atleast_1d(np.float64(v_)))
        for name, v_ in values.items():
            if name in ("dx", "dy", "dz"):
                ax = name[1]
                put(rx, ax, float(np.atleast_1d(getattr(tx, ax))[0]) + v_)
            elif name.startswith("tx_"):
                if name[3:] in ("x", "y", "z"):               # (the offset stays: Loop_pair keeps it as its own Point)
                    old = float(np.atleast_1d(getattr(tx, name[3:]))[0])
                    put(rx, name[3:], float(np.atleast_1d(getattr(rx, name[3:]))[0]) + (v_ - old))
                put(tx, name[3:], v_)
            else:
                put(rx, name[3:], v_)

    @property
    def probability(self):
        """DataPoint.probability (:454-489): sum of the log priors of the error levels that have one (+ the loop pair's,
        Loop_pair.probability :294-295)."""
        p = np.float64(0.0)
        if self._geom_moves:
            cur = loop_pair_values(self.transmitter, self.receiver)
            p += sum(m_.log_prior(cur[m_.name]) for m_ in self._geom_moves)
        if self._rel_prior is not None:
            p += self._rel_prior.log_prior(self._relative_error)
        if self._add_prior is not None:
            p += self._add_prior.log_prior(self._additive_error)
        return p


class TempestDataPoint(TdemDataPoint):
    """The reference's ``Tempest_datapoint`` (data/datapoint/Tempest_datapoint.py) at the object level: the channels hold the TOTAL
    field -- ``data = secondary_field + primary_field[component]`` (:106-113), ``predictedData = predicted secondary + predicted
    primary`` of the loop pair's geometry (:119-124) --, the additive error is a per-channel array times a sampled MULTIPLIER per
    component (:161-176), the relative error multiplies the total field.  The multipliers are the levels ``additive_error`` of this
    class's sampler interface (error_model(): rjmcmc.ErrorModel(tempest=True) carries the reference's three quirks about them:
    proposed last, one draw without redraw, no prior term in ``probability``)."""

    kind = "tempest"

    def __init__(self, *args, secondary_field=None, primary_field=None, channel_additive_error=None, additive_error_multiplier=None, **kw):
        super().__init__(*args, data=secondary_field, **kw)
        nc = self.n_error_groups
        self.primary_field = np.zeros(nc) if primary_field is None else np.asarray(primary_field, dtype=np.float64).copy()
        self.channel_additive_error = (np.ones(self.nChannels) if channel_additive_error is None
                                       else np.broadcast_to(np.asarray(channel_additive_error, dtype=np.float64), (self.nChannels,)).copy())
        self._additive_error = np.ones(nc) if additive_error_multiplier is None else np.asarray(additive_error_multiplier, dtype=np.float64).copy()
        self._multiplier0 = self._additive_error.copy()       # where the multipliers' proposal stays centred (set_proposals re-centres it)
        self._windows_per_group = [s_.nwindows for s_ in self.system for _ in range(s_.n_components)]

    @property
    def secondary_field(self):
        return self._data

    @property
    def data(self):                                           # Tempest_datapoint.data :106-113
        return self._data + np.repeat(self.primary_field, self._windows_per_group)

    @property
    def additive_error_multiplier(self):
        return self._additive_error

    @property
    def additive_error(self):
        return self._additive_error

    @additive_error.setter
    def additive_error(self, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        assert v.size == self.n_error_groups and np.all(v > 0.0), ValueError("one positive additive-error multiplier per (system, component)")
        self._additive_error = v.copy()

    @property
    def predicted_primary_field(self):
        """Free-space field of the transmitter dipole along the receiver's axes for the loops' current geometry, per component."""
        g = gaaem_tuple(loop_pair_values(self.transmitter, self.receiver))
        return GeometryMix(self.system, g[None, :]).primary_field()[0]

    @property
    def std(self):                                            # Tempest_datapoint.std :161-176
        return self.error_model().std(self.data, self._relative_error, self._additive_error)

    def make_engine(self, lmax=32, hankel_eps=None):
        return TdemEngine(self.system, self.z[0], self.offset, self.attitude, lmax=lmax, hankel_eps=hankel_eps,
                          loop_pair=loop_pair_values(self.transmitter, self.receiver), total_field=True)

    def error_model(self, reference_fm_dlogc=True):
        from . import rjmcmc
        grp = np.repeat(np.arange(self.n_error_groups), self._windows_per_group)
        return rjmcmc.ErrorModel(grp, grp, self.channel_additive_error, stale_prediction=reference_fm_dlogc, tempest=True,
                                 add_centre=self._multiplier0)

    def set_priors(self, relative_error_prior=None, additive_error_prior=None, data_prior=None, **kwargs):
        """Tempest_datapoint.set_priors (:478-487): the options' additive-error bounds become the MULTIPLIERS' prior (which
        Tempest_datapoint.probability never evaluates); everything else as TdemDataPoint."""
        from . import rjmcmc
        if additive_error_prior is None and kwargs.get("solve_additive_error", False):
            additive_error_prior = (kwargs["minimum_additive_error"], kwargs["maximum_additive_error"])
        super().set_priors(relative_error_prior, None, data_prior, **dict(kwargs, solve_additive_error=False))
        if additive_error_prior is not None:
            vec = lambda v_: np.broadcast_to(np.asarray(v_, dtype=np.float64), (self.n_error_groups,)).copy()
            self._add_prior = rjmcmc.ErrorPrior(vec(additive_error_prior[0]), vec(additive_error_prior[1]),
                                                self._add_prior.var if self._add_prior is not None else 0.0)

    def set_proposals(self, relative_error_proposal=None, additive_error_proposal=None, **kwargs):
        super().set_proposals(relative_error_proposal, None, **dict(kwargs, solve_additive_error=False))
        if additive_error_proposal is None and kwargs.get("solve_additive_error", False):
            additive_error_proposal = kwargs["additive_error_proposal_variance"]
        if additive_error_proposal is not None:
            assert self._add_prior is not None, ValueError("set_priors must come before set_proposals")
            self._add_prior.var = np.broadcast_to(np.asarray(additive_error_proposal, dtype=np.float64), (self.n_error_groups,)).copy()
            self._multiplier0 = self._additive_error.copy()   # (set_additive_error_proposal :503-508 centres it on the current multipliers)

    def perturb(self):
        """Tempest_datapoint.perturb (:339-341): TdemDataPoint.perturb (relative levels, loop pair), then the multipliers -- one
        joint draw, no redraw against their prior, from a proposal that stays where set_proposals put it."""
        add_prior, self._add_prior = self._add_prior, None
        try:
            super().perturb()
        finally:
            self._add_prior = add_prior
        if add_prior is not None and np.any(np.asarray(add_prior.var) > 0.0):
            self._additive_error = np.atleast_1d(add_prior.propose(self._prng, self._additive_error, redraw=False, centre=self._multiplier0))

    @property
    def probability(self):
        """Tempest_datapoint.probability (:475-476) = TdemDataPoint.probability with no additive-error prior set: the relative
        levels' (and the loop pair's) priors only."""
        add_prior, self._add_prior = self._add_prior, None
        try:
            return TdemDataPoint.probability.fget(self)
        finally:
            self._add_prior = add_prior
