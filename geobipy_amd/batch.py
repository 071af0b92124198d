"""Device-resident batches of soundings: the batched entry points of the hot path.

``FdemBatch`` keeps B soundings that share one ``FdemSystem`` as fp64 tensors in HBM and evaluates,
for all of them in one launch, what the reference evaluates one sounding at a time in
``Inference1D.accept_reject`` (inversion/Inference1D.py:572-597):
``FdemDataPoint.forward`` -> ``data_misfit`` -> ``likelihood(log=True)``.

HBM layout (all C-contiguous, fp64 unless noted):
    nlayers int32[B]      sigma[B, Lmax]   thk[B, Lmax]   height[B]
    data[B, 2F]           relative_error[B]               additive_error[B]
    predicted[B, 2F]      chi2[B]          logL[B]        (outputs, allocated once and reused)
PyTorch is only the allocator / stream provider here; all arithmetic happens in libgeobipy_amd.so.
"""
import numpy as np
import torch

from . import _lib


def _dev_f64(x, device, shape=None):
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    t = t.to(device=device, dtype=torch.float64).contiguous()
    if shape is not None:
        t = t.expand(shape).contiguous() if t.dim() == 0 or tuple(t.shape) != tuple(shape) else t
    return t


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class FdemBatch:
    """B soundings on one GPU.

    Parameters mirror the per-sounding objects of the reference: ``sigma``/``thk`` are
    ``Model.values`` / ``Model.mesh.widths`` (last width = inf is ignored), ``height`` is
    ``DataPoint.z``, ``data`` the observed ``FdemDataPoint.data`` (in-phase block then quadrature
    block), ``relative_error`` / ``additive_error`` one scalar per sounding (DataPoint.py:274).
    """

    def __init__(self, system, nlayers, sigma, thk, height, data=None, relative_error=None, additive_error=None,
                 device=None, hankel_eps_ppm=None, waves=0):
        if not torch.cuda.is_available():
            raise _lib.NativeLibraryError("FdemBatch needs a HIP device (torch.cuda.is_available() is False); "
                                          "there is no CPU fallback")
        _lib.load()
        self.system = system
        # waves per workgroup of the forward kernels: 0 = chosen from the batch size; 1..16 = fixed.  A performance hint: the
        # values do not depend on it, nor on how the soundings are batched (gbp_fdem_forward_ex)
        self.waves = int(waves)
        assert 0 <= self.waves <= 16, ValueError("waves must be in [0, 16]")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self._h_exact = system.handle()
        self._h = self._h_exact
        # Abscissa window: by default a sounding is evaluated with the filter abscissae whose terms can add up
        # to more than 1e-10 ppm at its own altitude (|rTE| <= 1 bounds every term) -- about half of the 120, every output
        # within 1e-10 ppm of the full sums, i.e. far below their own rounding error (1e-8 ppm) and the parity bar (1e-7 ppm),
        # and independent of the batch the sounding is in.  The Jacobian uses the same window: rTE is analytic and bounded by
        # 1 for conductivities in the right half plane, so |d rTE / d ln sigma_k| <= 2/pi (Cauchy estimate on the strip
        # |Im ln sigma| < pi/2) and the dropped terms are bounded by the same sum.  hankel_eps_ppm=0: all 120 / 140 abscissae.
        from .system import DEFAULT_HANKEL_EPS_PPM
        self.hankel_eps_ppm = DEFAULT_HANKEL_EPS_PPM if hankel_eps_ppm is None else float(hankel_eps_ppm)
        self.F = system.nFrequencies
        self.sigma = _dev_f64(sigma, self.device)
        assert self.sigma.dim() == 2, ValueError("sigma must have shape [B, Lmax]")
        self.B, self.Lmax = self.sigma.shape
        self.thk = _dev_f64(thk, self.device)
        assert tuple(self.thk.shape) == (self.B, self.Lmax), ValueError("thk must have shape [B, Lmax]")
        nl = torch.as_tensor(np.asarray(nlayers) if not torch.is_tensor(nlayers) else nlayers)
        if nl.dim() == 0:
            nl = nl.expand(self.B)
        self.nlayers = nl.to(device=self.device, dtype=torch.int32).contiguous()
        assert self.nlayers.numel() == self.B
        self.height = _dev_f64(height, self.device, (self.B,))
        if self.hankel_eps_ppm > 0.0 and self.B > 0:
            # the 1 m altitude bins this batch touches (one sync, at construction); the window itself is per sounding
            lo_hi = torch.stack([self.height.min(), self.height.max()]).cpu().numpy()
            with torch.cuda.device(self.device):
                self._h = system.handle_binned(self.hankel_eps_ppm, float(lo_hi[0]), float(lo_hi[1]))
        self.data = None if data is None else _dev_f64(data, self.device)
        if self.data is not None:
            assert tuple(self.data.shape) == (self.B, 2 * self.F), ValueError("data must have shape [B, 2F]")
        self.relative_error = None if relative_error is None else _dev_f64(relative_error, self.device, (self.B,))
        self.additive_error = None if additive_error is None else _dev_f64(additive_error, self.device, (self.B,))
        self.predicted = torch.empty((self.B, 2 * self.F), dtype=torch.float64, device=self.device)
        self.chi2 = torch.empty(self.B, dtype=torch.float64, device=self.device)
        self._max_layers = None
        self.logL = torch.empty(self.B, dtype=torch.float64, device=self.device)

    def validate(self):
        """The reference's host asserts (FD/fdem1d.py:29, DP/FdemDataPoint.py:541) plus sigma, thk > 0, evaluated on the
        device by ``gbp_fdem_validate`` (one small launch + one sync); raises for the first kind of bad row found."""
        st = self.status(with_outputs=False)
        bad = int((st != 0).sum().item())
        if bad:
            bits = int(torch.bitwise_or(st[st != 0][0], torch.zeros((), dtype=torch.int32, device=self.device)).item())
            msg = [m for b, m in ((1, "nlayers out of range"), (2, "conductivity must be > 0"), (4, "thickness must be > 0"),
                                  (8, "Sensor altitude must be above the top of the model")) if bits & b]
            raise AssertionError(ValueError("{} bad sounding(s): {}".format(bad, ", ".join(msg))))

    def status(self, pred=None, with_outputs=True):
        """int32[B] status word per sounding (include/geobipy_amd.h GBP_ROW_*: 1 nlayers out of range, 2 bad conductivity,
        4 bad thickness, 8 bad altitude, 16 non-finite prediction).  A batch is never aborted for one bad row: rows with
        nlayers > Lmax come back as NaN, the others are untouched."""
        pred = (self.predicted if pred is None else pred) if with_outputs else None
        out = torch.empty(self.B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gbp_fdem_validate(self.B, self.Lmax, 2 * self.F if pred is not None else 0,
                                                     self.nlayers.data_ptr(), self.sigma.data_ptr(), self.thk.data_ptr(),
                                                     self.height.data_ptr(), pred.data_ptr() if pred is not None else None,
                                                     out.data_ptr(), _stream_ptr(self.device)))
        return out

    # -- launches -----------------------------------------------------------------------------
    def forward(self, out=None):
        """pred[B, 2F] (ppm) for the current sigma / thk / height; one kernel launch, no sync."""
        out = self.predicted if out is None else out
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_fdem_forward_ex(self._h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(),
                                               self.sigma.data_ptr(), self.thk.data_ptr(), self.height.data_ptr(),
                                               out.data_ptr(), self.waves, _stream_ptr(self.device)))
        return out

    def forward_loglike(self, want_pred=True):
        """Fused forward + chi^2 + log-likelihood: returns (chi2[B], logL[B]); ``self.predicted`` is
        refreshed when ``want_pred``."""
        assert self.data is not None and self.relative_error is not None and self.additive_error is not None, \
            ValueError("data, relative_error and additive_error are needed for the likelihood")
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_fdem_forward_loglike_ex(
                self._h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(), self.sigma.data_ptr(), self.thk.data_ptr(),
                self.height.data_ptr(), self.data.data_ptr(), self.relative_error.data_ptr(),
                self.additive_error.data_ptr(), self.predicted.data_ptr() if want_pred else None,
                self.chi2.data_ptr(), self.logL.data_ptr(), self.waves, _stream_ptr(self.device)))
        return self.chi2, self.logL

    def loglike(self, pred=None):
        """chi^2 and logL for given predictions (DataPoint.data_misfit / likelihood(log=True))."""
        pred = self.predicted if pred is None else _dev_f64(pred, self.device)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_gauss_loglike(self.B, 2 * self.F, pred.data_ptr(), self.data.data_ptr(),
                                             self.relative_error.data_ptr(), self.additive_error.data_ptr(),
                                             self.chi2.data_ptr(), self.logL.data_ptr(), _stream_ptr(self.device)))
        return self.chi2, self.logL

    def fm_dlogc(self, exact=False):
        """Prediction and Jacobian of every sounding from one pass (FdemDataPoint.fm_dlogc): refreshes ``self.predicted``
        and returns J[B, 2F, Lmax]."""
        J = torch.empty((self.B, 2 * self.F, self.Lmax), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gbp_fdem_fm_dlogc(
                self._h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(), self.sigma.data_ptr(), self.thk.data_ptr(),
                self.height.data_ptr(), self.predicted.data_ptr(), J.data_ptr(), self.Lmax, 1 if exact else 0,
                _stream_ptr(self.device)))
        return J

    def sensitivity(self, out=None, exact=False, max_layers=None, bucket=True):
        """J[B, 2F, Lmax] = d pred / d ln(sigma) (FdemDataPoint.sensitivity -> nbFdem1dsen).

        ``exact=False`` reproduces the reference's expression (which is not the true derivative above the
        half-space); ``exact=True`` returns the true derivative.  ``max_layers`` is an
        upper bound of ``nlayers`` (default: computed once from the batch) that sizes the kernel's LDS working
        set.  With ``bucket=True`` a ragged batch whose deepest model has more than 8 layers is split by layer
        count (<= 8, <= 16, rest) into separate launches, so that the many shallow models of an rjMCMC
        population are not run at the low occupancy the few deep ones force (1040 B of LDS per layer and wave)."""
        if out is None:
            out = torch.empty((self.B, 2 * self.F, self.Lmax), dtype=torch.float64, device=self.device)
        if max_layers is None:
            if self._max_layers is None:
                self._max_layers = int(self.nlayers.max().item()) if self.B > 0 else 1
            max_layers = self._max_layers
        if bucket and max_layers > 8 and self.B >= 512:
            lo = 0
            for hi in (8, 16, self.Lmax):
                if hi <= lo or lo >= max_layers:
                    continue
                idx = torch.nonzero((self.nlayers > lo) & (self.nlayers <= hi)).flatten()
                if idx.numel() > 0:
                    sub = torch.empty((idx.numel(), 2 * self.F, self.Lmax), dtype=torch.float64, device=self.device)
                    self._launch_sens(self.nlayers[idx].contiguous(), self.sigma[idx].contiguous(),
                                      self.thk[idx].contiguous(), self.height[idx].contiguous(), sub,
                                      min(hi, max_layers), exact)
                    out[idx] = sub
                lo = hi
            return out
        self._launch_sens(self.nlayers, self.sigma, self.thk, self.height, out, max_layers, exact)
        return out

    def _launch_sens(self, nlayers, sigma, thk, height, out, max_layers, exact):
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_fdem_sensitivity_ex(self._h.ptr, nlayers.numel(), self.Lmax, nlayers.data_ptr(),
                                                   sigma.data_ptr(), thk.data_ptr(), height.data_ptr(),
                                                   out.data_ptr(), int(max_layers), 1 if exact else 0,
                                                   _stream_ptr(self.device)))

    def find_best_halfspace(self, minConductivity=1e-4, maxConductivity=1e4, nSamples=100):
        """Best-fitting half-space conductivity of every sounding: brute-force search over a log grid
        (EmDataPoint.find_best_halfspace, data/datapoint/EmDataPoint.py:148-186 -- 100 forward + misfit
        evaluations per sounding in the reference; here ONE launch over B * nSamples half-spaces).
        Returns (sigma_best[B], chi2_best[B])."""
        assert maxConductivity > minConductivity, ValueError("Maximum conductivity must be greater than the minimum")
        c = torch.logspace(np.log10(minConductivity), np.log10(maxConductivity), nSamples, dtype=torch.float64,
                           device=self.device)
        n = self.B * nSamples
        rep = lambda t: t.repeat_interleave(nSamples, dim=0)
        trial = FdemBatch(self.system, torch.ones(n, dtype=torch.int32), c.repeat(self.B)[:, None],
                          torch.zeros((n, 1), dtype=torch.float64), rep(self.height), data=rep(self.data),
                          relative_error=rep(self.relative_error), additive_error=rep(self.additive_error),
                          device=self.device)
        chi2, _ = trial.forward_loglike(want_pred=False)
        chi2 = chi2.view(self.B, nSamples)
        best = torch.argmin(chi2, dim=1)          # first minimum, like numpy.argmin in the reference
        return c[best], chi2.gather(1, best[:, None])[:, 0]

    def time_forward_loglike(self, reps, want_pred=False):
        """Average kernel time in ms over ``reps`` launches, measured with hipEvents on the launch stream."""
        import ctypes
        lib = _lib.load()
        ms = ctypes.c_float()
        with torch.cuda.device(self.device):
            _lib.check(lib.gbp_bench_time_forward_loglike(
                self._h.ptr, self.B, self.Lmax, self.nlayers.data_ptr(), self.sigma.data_ptr(), self.thk.data_ptr(),
                self.height.data_ptr(), self.data.data_ptr(), self.relative_error.data_ptr(),
                self.additive_error.data_ptr(), self.predicted.data_ptr() if want_pred else None,
                self.chi2.data_ptr(), self.logL.data_ptr(), _stream_ptr(self.device), int(reps), ctypes.byref(ms)))
        return float(ms.value)
