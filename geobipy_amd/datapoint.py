"""FdemDataPoint: the reference's per-sounding interface on top of the batched GPU path.

Mirrors (names, argument meaning, error behaviour) the hot-path members of
``geobipy/src/classes/data/datapoint/{DataPoint,EmDataPoint,FdemDataPoint}.py``:
``forward`` (:524-545), ``sensitivity`` / ``fm_dlogc`` (:530-559), ``std`` (DataPoint.py:268-282),
``active`` (EmDataPoint.py:44-56), ``deltaD`` (DataPoint.py:200-214), ``data_misfit`` (:502-525),
``likelihood`` (:491-500), and the rjMCMC members ``Inference1D.accept_reject`` calls: ``perturb`` (:531-573),
``probability`` (:454-489), ``set_priors`` / ``set_proposals`` (:575-644).  Each forward / Jacobian / likelihood call is a
B = 1 launch of the same kernels ``FdemBatch`` uses; a datapoint keeps no CPU implementation of any of them (the CPU test
tier injects the oracle through the ``engine`` hook).
"""
from copy import deepcopy

import numpy as np

from .batch import FdemBatch
from .model import Model
from .system import FdemSystem


class FdemDataPoint:
    def __init__(self, x=0.0, y=0.0, z=0.0, elevation=0.0, data=None, std=None, predictedData=None, system=None,
                 lineNumber=0.0, fiducial=0.0):
        self.system = system
        self.x, self.y, self.elevation = np.float64(x), np.float64(y), np.float64(elevation)
        self.z = np.atleast_1d(np.asarray(z, dtype=np.float64)).copy()
        self.lineNumber, self.fiducial = lineNumber, fiducial
        self.units = "ppm"
        n = self.nChannels
        self._data = np.zeros(n) if data is None else np.asarray(data, dtype=np.float64).copy()
        assert self._data.size == n, ValueError("data must have size {}".format(n))
        if std is None:
            std = np.ones(n) if data is None else 0.1 * self._data
        self._std = np.asarray(std, dtype=np.float64).copy()
        self._predictedData = np.zeros(n) if predictedData is None else np.asarray(predictedData, np.float64).copy()
        self._relative_error = np.full(self.nSystems, 0.01)
        self._additive_error = np.zeros(self.nSystems)
        self._sensitivity_matrix = None
        self._rel_prior = self._add_prior = None              # rjmcmc.ErrorPrior once set_priors / set_proposals ran
        self._z_move = None                                   # rjmcmc.HeightMove once set_priors(solve_z=True, ...) ran
        self._prng = None
        # TEST HOOK: an object with forward(edges, values) / sensitivity(edges, values) replaces the GPU launches (the CPU
        # tier passes the C oracle); the product never sets it -- without it a missing GPU / library raises
        self.engine = None

    # -- system -------------------------------------------------------------------------------
    @property
    def system(self):
        return self._system

    @system.setter
    def system(self, value):
        if value is None:
            raise ValueError("FdemDataPoint needs an FdemSystem (or a path to a .stm file)")
        if isinstance(value, (str, FdemSystem)):
            value = [value]
        assert all(isinstance(s, (str, FdemSystem)) for s in value), TypeError(
            "System must have items of type str or geobipy.FdemSystem")
        # (the reference's loop over systems writes every system's response to predictedData[:F_i] / [F_i:] -- FdemDataPoint.py:541-545,
        #  Jacobian :555-557 -- so a second system overwrites the first: one system is what the reference itself can run)
        assert len(value) == 1, ValueError("one FdemSystem per datapoint is supported")
        self._system = [FdemSystem.read(s) if isinstance(s, str) else s for s in value]

    @property
    def nSystems(self):
        return 1

    @property
    def nFrequencies(self):
        return np.asarray([s.nFrequencies for s in self._system])

    @property
    def nChannels(self):
        return int(2 * self._system[0].nFrequencies)

    # -- data / errors (DataPoint.py) -----------------------------------------------------------
    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, values):
        v = np.asarray(values, dtype=np.float64)
        assert v.size == self.nChannels
        self._data = v.copy()

    @property
    def predictedData(self):
        return self._predictedData

    @property
    def relative_error(self):
        return self._relative_error

    @relative_error.setter
    def relative_error(self, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        assert v.size == self.nSystems, ValueError("relative_error must be a list of size equal to the number of systems")
        assert np.all(v > 0.0), ValueError("Relative error {} must be > 0.0".format(v))
        self._relative_error = v.copy()

    @property
    def additive_error(self):
        return self._additive_error

    @additive_error.setter
    def additive_error(self, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        assert v.size == self.nSystems, ValueError("additive_error must have size 1")
        self._additive_error = v.copy()

    @property
    def active(self):
        d = self._data.copy()
        d[d <= 0.0] = np.nan
        return ~np.isnan(d)                                   # EmDataPoint.py:44-56

    @property
    def n_active_channels(self):
        return self.active.sum()

    @property
    def std(self):
        assert np.all(self.relative_error > 0.0), ValueError("relative_error must be > 0.0")
        variance = ((self.relative_error * self._data) ** 2.0) + (self.additive_error ** 2.0)   # DataPoint.py:274
        self._std[:] = np.sqrt(variance)
        return self._std

    @property
    def deltaD(self):
        return self._predictedData - self._data               # DataPoint.py:200-214

    @property
    def sensitivity_matrix(self):
        return self._sensitivity_matrix

    def __deepcopy__(self, memo={}):
        out = FdemDataPoint.__new__(FdemDataPoint)
        for k, v in self.__dict__.items():
            if k == "_ws":
                continue                      # device workspace is per object, rebuilt lazily
            setattr(out, k, v if k in ("_system", "_prng", "engine") else deepcopy(v, memo))   # shared, like the reference's system / prng
        return out

    # -- hot path: every call below is a GPU launch ----------------------------------------------
    # One persistent B = 1 device workspace per datapoint: inputs are packed on the host into a pinned buffer and
    # travel in ONE copy, forward() runs the fused forward + chi^2 + logL kernel, and the results come back in ONE
    # copy; data_misfit() / likelihood() reuse them while nothing they depend on has changed.
    _LCAP = 64

    def _workspace(self):
        import torch
        ws = getattr(self, "_ws", None)
        if ws is None:
            N, Lc = self.nChannels, self._LCAP
            n_in = 2 * Lc + 1 + N + 2
            ws = dict(
                host=torch.empty(n_in, dtype=torch.float64).pin_memory(), dev=torch.empty(n_in, dtype=torch.float64, device="cuda"),
                nl=torch.ones(1, dtype=torch.int32, device="cuda"), nl_host=torch.ones(1, dtype=torch.int32).pin_memory(),
                out=torch.empty(N + 2, dtype=torch.float64, device="cuda"), out_host=torch.empty(N + 2, dtype=torch.float64).pin_memory(),
                handle=self._system[0].handle(), stamp=None)
            self._ws = ws
        return ws

    def _stamp(self):
        return (self._predictedData.tobytes(), self._data.tobytes(), self._relative_error.tobytes(),
                self._additive_error.tobytes())

    def _pack(self, ws, mod):
        import torch
        N, Lc = self.nChannels, self._LCAP
        h = ws["host"].numpy()
        L = 1
        if mod is not None:
            assert np.isinf(mod.mesh.edges[-1]), ValueError(
                "mod.edges must have last entry be infinity for forward modelling.")       # FdemDataPoint.py:541
            assert self.z[0] >= mod.mesh.relative_to, "Sensor altitude must be above the top of the model"  # fdem1d.py:29
            L = int(mod.mesh.nCells)
            if L > Lc:
                return None
            h[:L] = mod.values
            h[Lc:Lc + L - 1] = mod.mesh.widths[:-1]
        h[2 * Lc] = self.z[0]
        h[2 * Lc + 1:2 * Lc + 1 + N] = self._data
        h[2 * Lc + 1 + N] = self._relative_error[0]
        h[2 * Lc + 2 + N] = self._additive_error[0]
        ws["nl_host"][0] = L
        ws["nl"].copy_(ws["nl_host"], non_blocking=True)
        ws["dev"].copy_(ws["host"], non_blocking=True)
        return L

    def _batch(self, mod):
        assert np.isinf(mod.mesh.edges[-1]), ValueError(
            "mod.edges must have last entry be infinity for forward modelling.")       # FdemDataPoint.py:541
        assert self.z[0] >= mod.mesh.relative_to, "Sensor altitude must be above the top of the model"  # fdem1d.py:29
        L = int(mod.mesh.nCells)
        thk = np.array(mod.mesh.widths, dtype=np.float64)
        thk[-1] = 0.0          # never read by the kernel (the reference passes inf)
        return FdemBatch(self._system[0], np.array([L]), mod.values[None, :], thk[None, :], self.z[:1],
                         data=self._data[None, :], relative_error=self._relative_error[:1],
                         additive_error=self._additive_error[:1])

    def _engine_model(self, mod):
        assert np.isinf(mod.mesh.edges[-1]), ValueError("mod.edges must have last entry be infinity for forward modelling.")
        return np.asarray(mod.mesh.edges[1:-1], dtype=np.float64), np.asarray(mod.values, dtype=np.float64)

    def _engine_height(self):
        return {} if self._z_move is None else {"z": float(self.z[0])}       # (a sampled height travels with every request)

    def forward(self, mod):
        """Forward model the data from the given model (FdemDataPoint.py:524-545)."""
        assert isinstance(mod, Model), TypeError("Invalid model class for forward modeling [1D]")
        if self.engine is not None:
            self._predictedData[:] = self.engine.forward(*self._engine_model(mod), **self._engine_height())
            return
        import torch
        from . import _lib
        ws = self._workspace()
        L = self._pack(ws, mod)
        if L is None:                       # deeper than the packed workspace: generic path
            self._predictedData[:] = self._batch(mod).forward().cpu().numpy()[0]
            ws["stamp"] = None
            return
        N, Lc = self.nChannels, self._LCAP
        base, es = ws["dev"].data_ptr(), 8
        _lib.check(_lib.load().gbp_fdem_forward_loglike(
            ws["handle"].ptr, 1, Lc, ws["nl"].data_ptr(), base, base + es * Lc, base + es * 2 * Lc,
            base + es * (2 * Lc + 1), base + es * (2 * Lc + 1 + N), base + es * (2 * Lc + 2 + N),
            ws["out"].data_ptr(), ws["out"].data_ptr() + es * N, ws["out"].data_ptr() + es * (N + 1),
            torch.cuda.current_stream().cuda_stream))
        ws["out_host"].copy_(ws["out"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        o = ws["out_host"].numpy()
        self._predictedData[:] = o[:N]
        ws["chi2"], ws["logL"] = float(o[N]), float(o[N + 1])
        ws["stamp"] = self._stamp()

    def sensitivity(self, mod, **kwargs):
        """J[2F, L] = d predictedData / d ln(sigma) (FdemDataPoint.py:530-559)."""
        assert isinstance(mod, Model), TypeError("Invalid model class for sensitivity matrix [1D]")
        if self.engine is not None:
            self._sensitivity_matrix = np.asarray(self.engine.sensitivity(*self._engine_model(mod), **self._engine_height()))
            return self._sensitivity_matrix
        import torch
        from . import _lib
        from .system import DEFAULT_HANKEL_EPS_PPM
        ws = self._workspace()
        L = self._pack(ws, mod)
        if L is None:                       # deeper than the packed workspace: generic path
            self._sensitivity_matrix = self._batch(mod).sensitivity().cpu().numpy()[0][:, : int(mod.mesh.nCells)]
            return self._sensitivity_matrix
        # the entry, handle and arguments FdemBatch.sensitivity uses for this sounding (abscissa window of its altitude), on the packed
        # workspace: one copy in, one launch, one copy out
        N, Lc = self.nChannels, self._LCAP
        if "J" not in ws:
            ws["J"] = torch.empty((N, Lc), dtype=torch.float64, device="cuda")
            ws["J_host"] = torch.empty((N, Lc), dtype=torch.float64).pin_memory()
        handle = self._system[0].handle_binned(DEFAULT_HANKEL_EPS_PPM, float(self.z[0]), float(self.z[0]))
        base, es = ws["dev"].data_ptr(), 8
        _lib.check(_lib.load().gbp_fdem_sensitivity_ex(handle.ptr, 1, Lc, ws["nl"].data_ptr(), base, base + es * Lc, base + es * 2 * Lc,
                                                       ws["J"].data_ptr(), L, 0, torch.cuda.current_stream().cuda_stream))
        ws["J_host"].copy_(ws["J"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self._sensitivity_matrix = ws["J_host"].numpy()[:, :L].copy()
        return self._sensitivity_matrix

    def fm_dlogc(self, mod):
        """Forward model and Jacobian (FdemDataPoint.py:547-551): the prediction of the fused forward kernel (so that
        data_misfit / likelihood are ready) and the Jacobian."""
        self.forward(mod)
        self.sensitivity(mod)

    def _loglike(self):
        if self.engine is not None:
            from . import rjmcmc
            return rjmcmc.gauss_loglike(self._predictedData, self._data, self.std)
        import torch
        from . import _lib
        ws = self._workspace()
        if ws["stamp"] is not None and ws["stamp"] == self._stamp():
            return ws["chi2"], ws["logL"]          # computed by the fused kernel of the last forward()
        N, Lc = self.nChannels, self._LCAP
        self._pack(ws, None)
        ws["out_host"][:N] = torch.from_numpy(self._predictedData)
        ws["out"].copy_(ws["out_host"], non_blocking=True)
        base, es, ob = ws["dev"].data_ptr(), 8, ws["out"].data_ptr()
        _lib.check(_lib.load().gbp_gauss_loglike(1, N, ob, base + es * (2 * Lc + 1), base + es * (2 * Lc + 1 + N),
                                                 base + es * (2 * Lc + 2 + N), ob + es * N, ob + es * (N + 1),
                                                 torch.cuda.current_stream().cuda_stream))
        ws["out_host"].copy_(ws["out"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        o = ws["out_host"].numpy()
        ws["chi2"], ws["logL"] = float(o[N]), float(o[N + 1])
        ws["stamp"] = self._stamp()
        return ws["chi2"], ws["logL"]

    def find_best_halfspace(self, minConductivity=1e-4, maxConductivity=1e4, nSamples=100):
        """Half-space Model that best fits the data (EmDataPoint.py:148-186)."""
        from .model import RectilinearMesh1D
        b = FdemBatch(self._system[0], np.array([1]), np.ones((1, 1)), np.zeros((1, 1)), self.z[:1],
                      data=self._data[None, :], relative_error=self._relative_error[:1],
                      additive_error=self._additive_error[:1])
        sig, _ = b.find_best_halfspace(minConductivity, maxConductivity, nSamples)
        return Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, np.inf]), values=sig.cpu().numpy())

    def data_misfit(self):
        """|| W_d (d_obs - d_pre) ||_2^2 over the active channels (DataPoint.py:502-525)."""
        return np.float64(self._loglike()[0])

    def likelihood(self, log):
        """Gaussian likelihood of the predicted data (DataPoint.py:491-500, MvNormalDistribution.py:201-216)."""
        ll = self._loglike()[1]
        return np.float64(ll) if log else np.float64(np.exp(ll))

    # -- rjMCMC members (DataPoint.py:454-489, 531-644): error-level priors, proposals and moves --------------------------
    def set_priors(self, relative_error_prior=None, additive_error_prior=None, data_prior=None, **kwargs):
        """DataPoint.set_priors (:575-595): log-uniform priors [minimum, maximum] on the error levels that are solved for
        (options-file keys solve_relative_error, minimum_relative_error, maximum_relative_error and the additive twins);
        ``prng`` is remembered for perturb().  The data prior of the reference is the Gaussian likelihood itself
        (observed data, diagonal covariance std^2) and needs no object here."""
        from . import rjmcmc
        self._prng = kwargs.get("prng", self._prng)
        if relative_error_prior is None and kwargs.get("solve_relative_error", False):
            relative_error_prior = (kwargs["minimum_relative_error"], kwargs["maximum_relative_error"])
        if additive_error_prior is None and kwargs.get("solve_additive_error", False):
            additive_error_prior = (kwargs["minimum_additive_error"], kwargs["maximum_additive_error"])
        if kwargs.get("solve_z", False):                      # Point.set_priors (pointcloud/Point.py:949-965): uniform z +- maximum_z_change
            self._z_move = rjmcmc.HeightMove(float(self.z[0]), kwargs["maximum_z_change"], self._z_move.scale if self._z_move is not None else 0.0)
        var = lambda p: p.var if p is not None else 0.0
        if relative_error_prior is not None:
            lo, hi = (np.atleast_1d(v).astype(np.float64)[0] for v in relative_error_prior)
            self._rel_prior = rjmcmc.ErrorPrior(lo, hi, var(self._rel_prior))
        if additive_error_prior is not None:
            lo, hi = (np.atleast_1d(v).astype(np.float64)[0] for v in additive_error_prior)
            self._add_prior = rjmcmc.ErrorPrior(lo, hi, var(self._add_prior))

    def set_proposals(self, relative_error_proposal=None, additive_error_proposal=None, **kwargs):
        """DataPoint.set_proposals (:597-644): log-normal random walks with the options file's proposal variances."""
        self._prng = kwargs.get("prng", self._prng)
        if kwargs.get("solve_z", False):                      # Point.set_proposals (:967-983): Normal(z, z_proposal_variance)
            assert self._z_move is not None, ValueError("set_priors must come before set_proposals")
            self._z_move.scale = float(kwargs["z_proposal_variance"])
        if relative_error_proposal is None and kwargs.get("solve_relative_error", False):
            relative_error_proposal = kwargs["relative_error_proposal_variance"]
        if additive_error_proposal is None and kwargs.get("solve_additive_error", False):
            additive_error_proposal = kwargs["additive_error_proposal_variance"]
        for prior, v in ((self._rel_prior, relative_error_proposal), (self._add_prior, additive_error_proposal)):
            if v is not None:
                assert prior is not None, ValueError("set_priors must come before set_proposals")
                prior.var = float(np.atleast_1d(v)[0])

    def perturb(self):
        """DataPoint.perturb (:531-573): relative then additive error, each redrawn while outside its prior (the current
        value is kept at the 10th redraw, StatArray.propose :620-638) -- after the height, when it is sampled (Point.perturb,
        pointcloud/Point.py:614-621, reached first through super().perturb())."""
        if self._z_move is not None and self._z_move.scale > 0.0:
            self.z[0] = self._z_move.propose(self._prng, float(self.z[0]))
        if self._rel_prior is not None and self._rel_prior.var > 0.0:
            self._relative_error = np.atleast_1d(self._rel_prior.propose(self._prng, float(self._relative_error[0])))
        if self._add_prior is not None and self._add_prior.var > 0.0:
            self._additive_error = np.atleast_1d(self._add_prior.propose(self._prng, float(self._additive_error[0])))

    @property
    def probability(self):
        """DataPoint.probability (:454-489): sum of the log priors of the error levels that have one."""
        p = np.float64(0.0)
        if self._z_move is not None:                          # Point.probability (pointcloud/Point.py:159-197)
            p += self._z_move.log_prior(float(self.z[0]))
        if self._rel_prior is not None:
            p += self._rel_prior.log_prior(float(self._relative_error[0]))
        if self._add_prior is not None:
            p += self._add_prior.log_prior(float(self._additive_error[0]))
        return p
