"""Acquisition-system classes: host-side mirror of the reference interface the hot path consumes.

Mirrors (names and argument meaning) ``geobipy/src/classes/system/FdemSystem.py`` and
``CircularLoop.py`` / ``EmLoop.py`` of the reference, restricted to what the forward solve reads:
frequencies, loop orientations / moments / offsets, the Hankel filter tables, ``tensor_id``,
``loop_offsets``, ``loop_separation`` and the ``.stm`` reader.  Plotting, HDF and MPI shipping of the
reference classes are out of scope.
"""
import numpy as np

from . import _lib
from .filters import W0_J0_120, W1_J1_140, base_abscissae

_ORI = {"x": 0, "y": 1, "z": 2}
_ORI_NAMES = ("x", "y", "z")


class CircularLoop:
    """Loops of an EM system, one entry per frequency (reference: CircularLoop.py:8-44, EmLoop.py:160-181)."""

    def __init__(self, x=None, y=None, z=None, elevation=None, orientation=None, moment=None, pitch=None,
                 roll=None, yaw=None, radius=None, **kwargs):
        n = 1 if orientation is None else np.size(orientation)

        def arr(v, fill=0.0):
            if v is None:
                return np.full(n, fill, dtype=np.float64)
            return np.atleast_1d(np.asarray(v, dtype=np.float64)).copy()

        self.x, self.y, self.z = arr(x), arr(y), arr(z)
        self.moment = arr(moment, 1.0)
        self.pitch, self.roll, self.yaw = arr(pitch), arr(roll), arr(yaw)
        self.radius = arr(radius, 0.0)              # the reference default (CircularLoop.py:36-40: zeros)
        self.elevation = arr(elevation, 0.0)
        self._orientation = np.full(n, 2, dtype=np.int32)
        if orientation is not None:
            self.orientation = orientation

    @property
    def nPoints(self):
        return self._orientation.size

    @property
    def orientation(self):
        return [_ORI_NAMES[i] for i in self._orientation]

    @orientation.setter
    def orientation(self, values):
        values = [values] if isinstance(values, str) else list(values)
        self._orientation = np.asarray([_ORI[str(v).replace(" ", "")] for v in values], dtype=np.int32)


class FdemSystem:
    """Frequency-domain EM acquisition system (reference: system/FdemSystem.py:14-337)."""

    def __init__(self, frequencies, transmitter, receiver, n_frequencies=None):
        self._filename = None
        self.frequencies = np.atleast_1d(np.asarray(frequencies, dtype=np.float64)).copy()
        assert isinstance(transmitter, CircularLoop), TypeError("transmitter must have type CircularLoop")
        assert isinstance(receiver, CircularLoop), TypeError("receiver must have type CircularLoop")
        assert transmitter.nPoints == self.nFrequencies, ValueError(
            "Must have {} transmitters, one for each frequency".format(self.nFrequencies))
        assert receiver.nPoints == self.nFrequencies, ValueError(
            "Must have {} receivers, one for each frequency".format(self.nFrequencies))
        self.transmitter = transmitter
        self.receiver = receiver
        self._handle = None
        self._tables = None

    # -- reference properties ---------------------------------------------------------------
    @property
    def nFrequencies(self):
        return self.frequencies.size

    @property
    def loop_offsets(self):
        """[3, F] receiver - transmitter offsets (FdemSystem.py:52-57)."""
        return np.vstack([self.receiver.x - self.transmitter.x,
                          self.receiver.y - self.transmitter.y,
                          self.receiver.z - self.transmitter.z])

    @property
    def loop_separation(self):
        return np.linalg.norm(self.loop_offsets, axis=0)   # FdemSystem.py:59-61

    @property
    def tensor_id(self):
        """1 + 3*rx_orient + tx_orient (FdemSystem.py:199-203): zz=9, xx=1, Tx z/Rx x=3, Tx x/Rx z=7."""
        return (1 + ((self.receiver._orientation * 3) + self.transmitter._orientation)).astype(np.int32)

    def _filter_tables(self):
        if self._tables is None:
            l0, l1 = base_abscissae()
            r = 1.0 / self.loop_separation
            lam0 = np.ascontiguousarray(l0[None, :] * r[:, None])   # FdemSystem.py:67-83
            lam1 = np.ascontiguousarray(l1[None, :] * r[:, None])   # FdemSystem.py:85-101
            self._tables = (lam0, lam1)
        return self._tables

    @property
    def lamda0(self):
        return self._filter_tables()[0]

    @property
    def lamda1(self):
        return self._filter_tables()[1]

    @property
    def lamda02(self):
        return self.lamda0 ** 2.0

    @property
    def lamda12(self):
        return self.lamda1 ** 2.0

    @property
    def w0(self):
        return W0_J0_120

    @property
    def w1(self):
        return W1_J1_140

    @classmethod
    def read(cls, filename):
        """Read a ``.stm`` system file: header line then one row per frequency
        ``freq, tor, tmom, tx, ty, tz, ror, rmom, rx, ry, rz`` (FdemSystem.py:146-183)."""
        rows = []
        with open(filename) as f:
            next(f)
            for line in f:
                p = [c.strip() for c in line.split(",")]
                if len(p) >= 11:
                    rows.append(p)
        if not rows:
            raise ValueError("no frequency rows in system file {}".format(filename))
        col = lambda i: np.asarray([float(p[i]) for p in rows], dtype=np.float64)
        tx = CircularLoop(orientation=[p[1] for p in rows], moment=col(2), x=col(3), y=col(4), z=col(5))
        rx = CircularLoop(orientation=[p[6] for p in rows], moment=col(7), x=col(8), y=col(9), z=col(10))
        self = cls(col(0), tx, rx)
        self._filename = filename
        return self

    def __deepcopy__(self, memo={}):
        # the reference shares the system between copies of a datapoint (FdemDataPoint.py:83-87);
        # the native handle is immutable, so sharing is safe
        return self

    @property
    def summary(self):
        return "FdemSystem: \n{}\nfrequencies: {}\nloop_offsets:\n{}\n".format(
            self._filename, self.frequencies, self.loop_offsets)

    # -- native handle ------------------------------------------------------------------------
    def native_args(self):
        """The per-system arguments of the reference's nbFdem1dfwd (FD/fdem1d.py:31-49) as arrays."""
        lam0, lam1 = self._filter_tables()
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        return dict(
            tid=np.ascontiguousarray(self.tensor_id, dtype=np.int32), frequencies=f64(self.frequencies),
            tx_z=f64(self.transmitter.z), rx_z=f64(self.receiver.z), tx_moment=f64(self.transmitter.moment),
            scale=f64(self.transmitter.moment * self.receiver.moment),     # fdem1d.py:33
            rx_off=f64(self.loop_offsets[0, :]),                          # fdem1d.py:34
            separation=f64(self.loop_separation), w0=f64(self.w0), lamda0=f64(lam0), w1=f64(self.w1),
            lamda1=f64(lam1))

    def handle(self, eps_ppm=0.0, min_altitude=0.0):
        """Opaque gbp_fdem_system* on the CURRENT HIP device (tables uploaded once per device, then cached).

        ``eps_ppm > 0`` selects the accuracy-budgeted abscissa window (see gbp_fdem_system_create_windowed):
        valid for soundings at altitude >= ``min_altitude``.  Handles are cached per (device, eps, altitude floor): the
        tables live in the HBM of the device that was current when they were uploaded, so one FdemSystem used on two
        GPUs of a process gets two handles."""
        import torch
        dev = int(torch.cuda.current_device()) if torch.cuda.is_available() else -1
        if not isinstance(getattr(self, "_handles", None), dict):
            self._handles = {}
        windowed = eps_ppm > 0.0
        key = (dev, float(eps_ppm), float(np.floor(min_altitude))) if windowed else (dev, 0.0, 0.0)   # 1 m altitude bins keep the cache small
        if key not in self._handles:
            self._handles[key] = (NativeSystem(self.native_args(), eps_ppm=key[1], min_altitude=key[2]) if windowed
                                  else NativeSystem(self.native_args()))
        return self._handles[key]

    def handle_binned(self, eps_ppm, alt_lo, alt_hi):
        """Handle with PER-SOUNDING abscissa windows (gbp_fdem_system_create_binned): 1 m altitude bins covering
        [alt_lo, alt_hi]; a sounding is evaluated with the abscissae whose total contribution can exceed ``eps_ppm`` at the
        floor of its own altitude bin, whatever batch it is in.  Cached per (device, eps); rebuilt when a batch needs a wider
        range of altitudes."""
        import torch
        dev = int(torch.cuda.current_device()) if torch.cuda.is_available() else -1
        if not isinstance(getattr(self, "_handles", None), dict):
            self._handles = {}
        lo = max(0, int(np.floor(alt_lo))) if np.isfinite(alt_lo) else 0
        hi = max(lo, int(np.floor(alt_hi))) if np.isfinite(alt_hi) else lo
        key = (dev, "bins", float(eps_ppm))
        cur = self._handles.get(key)
        if cur is not None:
            c_lo, c_n = cur.bins
            if c_lo <= lo and hi < c_lo + c_n:
                return cur
            lo, hi = min(lo, c_lo), max(hi, c_lo + c_n - 1)
        n = min(hi - lo + 1, 1024)                   # above the last bin the last window is used (its bound still holds)
        self._handles[key] = NativeSystem(self.native_args(), eps_ppm=eps_ppm, bins=(lo, n))
        return self._handles[key]


# FdemBatch's default abscissa window: each output within this of the full filter sums.  1e-10 ppm is 75 x below the measured
# rounding error of those sums against the scalar oracle (7.5e-9 ppm) and 1000 x below the parity bar (1e-7 ppm); it is also where
# every frequency's window reaches the kernels' minimum of 64 abscissae (one 64-lane pass) at survey altitudes: 640 of 1200 points
# for the 10-frequency system, 10 passes instead of 19 (1e-12 ppm: 651 - 673 points = 11 passes, 14 % slower for nothing visible).
DEFAULT_HANKEL_EPS_PPM = 1.0e-10


class NativeSystem:
    """RAII wrapper of gbp_fdem_system_create / _create_windowed / _create_binned / _destroy."""

    def __init__(self, a, eps_ppm=0.0, min_altitude=0.0, bins=None):
        import ctypes
        lib = _lib.load()
        self._lib = lib
        self.nF = int(a["frequencies"].size)
        dp = lambda x: x.ctypes.data_as(_lib.c_double_p)
        h = ctypes.c_void_p()
        common = (self.nF, a["tid"].ctypes.data_as(_lib.c_int32_p), dp(a["frequencies"]), dp(a["tx_z"]), dp(a["rx_z"]),
                  dp(a["tx_moment"]), dp(a["scale"]), dp(a["rx_off"]), dp(a["separation"]), dp(a["w0"]),
                  dp(a["lamda0"]), dp(a["w1"]), dp(a["lamda1"]))
        if bins is not None:                         # (first altitude in metres, number of 1 m bins)
            st = lib.gbp_fdem_system_create_binned(*common, float(eps_ppm), int(bins[0]), int(bins[1]), ctypes.byref(h))
        else:
            st = lib.gbp_fdem_system_create_windowed(*common, float(eps_ppm), float(min_altitude), ctypes.byref(h))
        _lib.check(st)
        self.ptr = h
        self.bins = bins

    def bin_points(self, altitude):
        """Abscissa points evaluated for a sounding at ``altitude`` (m) with this handle."""
        import ctypes
        n = ctypes.c_int()
        _lib.check(self._lib.gbp_fdem_system_bin_points(self.ptr, int(np.floor(altitude)), ctypes.byref(n)))
        return n.value

    @property
    def npoints(self):
        import ctypes
        n = ctypes.c_int()
        _lib.check(self._lib.gbp_fdem_system_npoints(self.ptr, ctypes.byref(n)))
        return n.value

    def h0(self):
        out = np.empty(2 * self.nF)
        _lib.check(self._lib.gbp_fdem_system_h0(self.ptr, out.ctypes.data_as(_lib.c_double_p)))
        return out[0::2] + 1j * out[1::2]

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._lib.gbp_fdem_system_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass
