"""Per-sounding rjMCMC driver on top of the GPU hot path.

``Inference1D`` mirrors how the reference's harness uses its class of the same name
(inversion/Inference3D.py:617-620: ``Inference1D(prng=..., **options).initialize(datapoint)`` then ``accept_reject()``
/ ``update()`` per iteration, inversion/Inference1D.py:353-464, 537-631, 705-790) for FDEM data with the
Resolve-style option set (solve_gradient, solve relative / additive error, no height move).  The sampler logic is
the host-side restatement in ``rjmcmc.py``; every forward solve, Jacobian and likelihood goes through the HIP
kernels.  With the reference's seed it reproduces the reference's chain decision by decision
(tests/test_rjmcmc.py).  ``BatchedInference`` runs many soundings in lockstep, each with its own generator, and
batches their kernel calls: one forward launch and one Jacobian launch per phase instead of one per sounding.
Posterior hit-maps / HDF output of the reference (row f-4) are not part of this module.
"""
import numpy as np

from . import rjmcmc
from .batch import FdemBatch


class GpuEngine:
    """forward / sensitivity of single soundings or lists of soundings through FdemBatch (one launch per call)."""

    def __init__(self, system, z, lmax=32, exact_jacobian=False):
        # exact_jacobian=False reproduces the reference's Jacobian expression, which is what a
        # chain needs to follow the reference decision by decision; True uses the true derivative in the proposals
        self.system, self.z, self.lmax, self.exact = system, float(z), int(lmax), bool(exact_jacobian)

    def _batch(self, models, heights=None):
        n = len(models)
        nl = np.array([v.size for _, v in models], dtype=np.int32)
        width = max(self.lmax, int(nl.max()) if n else 1)          # (a model deeper than lmax widens the batch instead of failing)
        sig = np.ones((n, width))
        thk = np.zeros((n, width))
        for i, (e, v) in enumerate(models):
            sig[i, : v.size] = v
            thk[i, : v.size - 1] = np.diff(np.r_[0.0, e])
        h = np.full(n, self.z) if heights is None else np.asarray(heights, dtype=np.float64)
        return FdemBatch(self.system, nl, sig, thk, h), nl

    def forward_many(self, models, heights=None):
        b, _ = self._batch(models, heights)
        return b.forward().cpu().numpy()

    def sensitivity_many(self, models, heights=None):
        b, nl = self._batch(models, heights)
        J = b.sensitivity(max_layers=int(nl.max()), exact=self.exact).cpu().numpy()
        return [J[i][:, : nl[i]] for i in range(len(models))]

    # One sounding at a time -- what the serial, reference-signature loop asks for thousands of times: a persistent B = 1 workspace
    # (one pinned staging buffer -> ONE copy in, the same entries FdemBatch calls with the same arguments, one copy out) instead of
    # a fresh FdemBatch with four small uploads per request.  Same kernels, same values.
    def _single(self, edges, values, z):
        import torch
        from . import _lib
        from .system import DEFAULT_HANKEL_EPS_PPM
        L, Lc = int(np.size(values)), self.lmax
        if L > Lc:
            return None
        ws = getattr(self, "_ws", None)
        if ws is None:
            if not torch.cuda.is_available():
                raise _lib.NativeLibraryError("GpuEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
            _lib.load()
            N = 2 * self.system.nFrequencies
            ws = self._ws = dict(
                host=torch.zeros(2 * Lc + 1, dtype=torch.float64).pin_memory(), dev=torch.zeros(2 * Lc + 1, dtype=torch.float64, device="cuda"),
                nl_host=torch.ones(1, dtype=torch.int32).pin_memory(), nl=torch.ones(1, dtype=torch.int32, device="cuda"),
                pred=torch.empty(N, dtype=torch.float64, device="cuda"), pred_host=torch.empty(N, dtype=torch.float64).pin_memory(),
                J=torch.empty((N, Lc), dtype=torch.float64, device="cuda"), J_host=torch.empty((N, Lc), dtype=torch.float64).pin_memory(), N=N)
        h = ws["host"].numpy()
        h[:L] = values
        h[Lc:Lc + L - 1] = np.diff(np.r_[0.0, edges])
        h[Lc + L - 1:2 * Lc] = 0.0
        h[2 * Lc] = self.z if z is None else float(z)
        ws["nl_host"][0] = L
        ws["nl"].copy_(ws["nl_host"], non_blocking=True)
        ws["dev"].copy_(ws["host"], non_blocking=True)
        ws["handle"] = self.system.handle_binned(DEFAULT_HANKEL_EPS_PPM, float(h[2 * Lc]), float(h[2 * Lc]))   # (cached by the system)
        return ws, L

    def forward(self, edges, values, z=None):
        one = self._single(edges, values, z)
        if one is None:                      # deeper than the workspace: the batch path
            return self.forward_many([(edges, values)], None if z is None else [z])[0]
        import torch
        from . import _lib
        ws, L = one
        base, Lc = ws["dev"].data_ptr(), self.lmax
        _lib.check(_lib.load().gbp_fdem_forward_ex(ws["handle"].ptr, 1, Lc, ws["nl"].data_ptr(), base, base + 8 * Lc, base + 16 * Lc,
                                                   ws["pred"].data_ptr(), 0, torch.cuda.current_stream().cuda_stream))
        ws["pred_host"].copy_(ws["pred"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return ws["pred_host"].numpy().copy()

    def sensitivity(self, edges, values, z=None):
        one = self._single(edges, values, z)
        if one is None:
            return self.sensitivity_many([(edges, values)], None if z is None else [z])[0]
        import torch
        from . import _lib
        ws, L = one
        base, Lc = ws["dev"].data_ptr(), self.lmax
        _lib.check(_lib.load().gbp_fdem_sensitivity_ex(ws["handle"].ptr, 1, Lc, ws["nl"].data_ptr(), base, base + 8 * Lc, base + 16 * Lc,
                                                       ws["J"].data_ptr(), L, 1 if self.exact else 0, torch.cuda.current_stream().cuda_stream))
        ws["J_host"].copy_(ws["J"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return ws["J_host"].numpy()[:, :L].copy()


OPTION_DEFAULTS = dict(covariance_scaling=1.0, gradient_standard_deviation=1.5, factor=10.0, minimum_thickness=1.0,
                       solve_gradient=True)


def _priors_from_options(o, value_mean):
    sp = rjmcmc.StructurePrior(o["maximum_number_of_layers"], o["minimum_depth"], o["maximum_depth"],
                               o["minimum_thickness"],
                               [o["probability_of_birth"], o["probability_of_death"], o["probability_of_perturb"],
                                o["probability_of_no_change"]])
    vp = rjmcmc.ValuePrior(value_mean, o["factor"], o["gradient_standard_deviation"], o["solve_gradient"],
                           bool(o.get("solve_parameter", False)), o.get("parameter_limits"))
    # error levels: scalars (frequency-domain data, one level each) or lists per level (time-domain options files:
    # one relative level per system x component, one additive level per system)
    lv = lambda key: o[key] if np.ndim(o[key]) == 0 else np.asarray(o[key], dtype=np.float64)
    rp = rjmcmc.ErrorPrior(lv("minimum_relative_error"), lv("maximum_relative_error"), lv("relative_error_proposal_variance"))
    ap = rjmcmc.ErrorPrior(lv("minimum_additive_error"), lv("maximum_additive_error"), lv("additive_error_proposal_variance"))
    return sp, vp, rp, ap


def height_move_from_options(o, z0):
    """HeightMove when the options carry the keys the reference's data point reads (``solve_z``, ``maximum_z_change``,
    ``z_proposal_variance``: pointcloud/Point.py:949-983), else None."""
    if not o.get("solve_z", False):
        return None
    return rjmcmc.HeightMove(z0, o["maximum_z_change"], o["z_proposal_variance"])


def initial_state(engine, data, o, error_model=None, z_move=None, geom_moves=None):
    """Inference1D.initialize (inversion/Inference1D.py:353-464, 485-535): best half-space out of 100 log-spaced
    conductivities (EmDataPoint.find_best_halfspace), its forward / Jacobian, prior and likelihood."""
    rel, add = o["initial_relative_error"], o["initial_additive_error"]
    if error_model is None:
        std = np.sqrt((rel * data) ** 2.0 + add ** 2.0)
    else:
        rel, add = np.atleast_1d(np.asarray(rel, dtype=np.float64)), np.atleast_1d(np.asarray(add, dtype=np.float64))
        std = error_model.std(data, rel, add)
    grid = np.logspace(-4.0, 4.0, 100)
    none = np.zeros(0)
    models = [(none, np.array([c])) for c in grid]
    preds = engine.forward_many(models) if hasattr(engine, "forward_many") else np.stack([engine.forward(e, v) for e, v in models])
    phi = [rjmcmc.gauss_loglike(p, data, std)[0] for p in preds]
    sigma = np.array([grid[int(np.argmin(phi))]])
    sp, vp, rp, ap = _priors_from_options(o, sigma.item())
    pred, J = engine.forward(none, sigma), engine.sensitivity(none, sigma)
    misfit, like = rjmcmc.gauss_loglike(pred, data, std)
    prior = rjmcmc.model_log_prior(sp, vp, none, sigma) + rp.log_prior(rel)
    if not (error_model is not None and error_model.tempest):           # (Tempest's multipliers: no prior term, rjmcmc.ErrorModel)
        prior += ap.log_prior(add)
    if z_move is not None:
        prior += z_move.log_prior(z_move.z0)
    geom = None
    if geom_moves:
        prior += sum(m.log_prior(m.z0) for m in geom_moves)
        geom = {m.name: m.z0 for m in geom_moves}
    return (sp, vp, rp, ap), rjmcmc.ChainState(none, sigma, rel, add, pred, J, prior, like, misfit, None if z_move is None else z_move.z0, geom)


class Posteriors:
    """The posteriors Inference1D.update accumulates every iteration (Model.update_posteriors, model/Model.py:810-847;
    RectilinearMesh1D.update_posteriors, mesh/RectilinearMesh1D.py:1595-1610), on the reference's grids
    (RectilinearMesh1D.set_posteriors :1438-1455, Model.set_posteriors): layer-count histogram, interface-depth histogram
    (interfaces across which the conductivity changes by more than ``ratio``) and the conductivity-depth hit map
    ``values[n_value_bins, n_depth_bins]`` on log10(sigma / prior mean) in +-4 prior standard deviations.  Host-side
    (numpy) twin of the accumulators of the device sampler; the reference's own counts are reproduced bit for bit
    (tests/test_rjmcmc.py)."""

    def __init__(self, max_cells, max_edge, min_width, value_mean, factor=10.0, n_value_bins=250, ratio=0.5,
                 relative_error_bounds=None, additive_error_bounds=None, n_error_bins=99, height_edges=None, geometry_edges=None):
        self.ratio = ratio
        # height (Point.set_z_posterior :1010-1017): the cells of the uniform prior, when the height is sampled
        self.height_edges = None if height_edges is None else np.asarray(height_edges, dtype=np.float64)
        self.height = np.zeros(0 if height_edges is None else self.height_edges.size - 1, dtype=np.int64)
        # sampled scalars of the loop pair (EmLoop.set_pitch_posterior ...: the cells of their uniform priors)
        self.geometry_edges = {n_: np.asarray(e_, dtype=np.float64) for n_, e_ in (geometry_edges or {}).items()}
        self.geometry = {n_: np.zeros(e_.size - 1, dtype=np.int64) for n_, e_ in self.geometry_edges.items()}
        # error levels (DataPoint.set_posteriors :651-694): n_error_bins cells, uniform in log10 between the prior bounds
        # (several levels -- time-domain data: bounds are arrays, one histogram per level: edges [G, n + 1], counts [G, n])
        grid = lambda b: None if b is None else np.linspace(*np.log10(np.asarray(b, dtype=np.float64)), n_error_bins + 1).T
        self.rel_edges, self.add_edges = grid(relative_error_bounds), grid(additive_error_bounds)
        shape = lambda e: (n_error_bins,) if e is None or e.ndim == 1 else (e.shape[0], n_error_bins)
        self.relative_error = np.zeros(shape(self.rel_edges), dtype=np.int64)
        self.additive_error = np.zeros(shape(self.add_edges), dtype=np.int64)
        self.depth_edges = np.arange(0.0, 1.1 * max_edge, 0.5 * min_width)
        self.depth_centres = 0.5 * (self.depth_edges[:-1] + self.depth_edges[1:])
        half = 4.0 * np.sqrt(np.log(1.0 + factor) ** 2.0)           # MvNormal.bins(nBins, nStd=4): +- 4 std of ln sigma ...
        self.value_edges = np.linspace(-half, half, n_value_bins + 1) / np.log(10.0)   # ... on a log10 axis
        self.relative_to = np.log10(value_mean)
        self.n_cells = np.zeros(int(max_cells) + 1, dtype=np.int64)
        self.edges = np.zeros(self.depth_centres.size, dtype=np.int64)
        self.values = np.zeros((n_value_bins, self.depth_centres.size), dtype=np.int64)

    def reset(self):
        for a in (self.n_cells, self.edges, self.values, self.relative_error, self.additive_error, self.height) + tuple(self.geometry.values()):
            a[:] = 0

    def update(self, edges, values, rel=None, add=None, z=None, geom=None):
        """``edges``: interior interface depths; ``values``: layer conductivities; ``rel`` / ``add``: error levels; ``z``: height;
        ``geom``: sampled scalars of the loop pair."""
        for n_, v_ in (geom or {}).items():
            e_ = self.geometry_edges.get(n_)
            if e_ is not None and e_[0] <= v_ <= e_[-1]:
                self.geometry[n_][min(np.searchsorted(e_, v_, side="right") - 1, e_.size - 2)] += 1
        if z is not None and self.height_edges is not None and self.height_edges[0] <= z <= self.height_edges[-1]:
            self.height[min(np.searchsorted(self.height_edges, z, side="right") - 1, self.height.size - 1)] += 1
        for x, grid, hist in ((rel, self.rel_edges, self.relative_error), (add, self.add_edges, self.additive_error)):
            if x is not None and grid is not None:
                if grid.ndim == 1:
                    hist[np.clip(np.searchsorted(grid, np.log10(x), side="right") - 1, 0, hist.size - 1)] += 1
                else:
                    for g_, xv in enumerate(np.atleast_1d(x)):
                        hist[g_, np.clip(np.searchsorted(grid[g_], np.log10(xv), side="right") - 1, 0, hist.shape[1] - 1)] += 1
        k = values.size
        self.n_cells[k] += 1
        if k > 1:
            r = np.exp(np.diff(np.log(values)))
            d = edges[(r <= 1.0 - self.ratio) | (r >= 1.0 + self.ratio)]
            d = d[(d >= self.depth_edges[0]) & (d < self.depth_edges[-1])]           # Histogram.update(trim=True)
            np.add.at(self.edges, np.searchsorted(self.depth_edges, d, side="right") - 1, 1)
            # the reference's piecewise-constant interpolation: a ramp of relative width 1e-6 below every interface
            full = np.r_[0.0, edges, self.depth_edges[-1]]
            xp = np.kron(full, [1.0, 1.000001])[1:-1]
            at_centres = np.interp(self.depth_centres, xp, np.kron(values, [1.0, 1.0]))
        else:
            at_centres = np.full(self.depth_centres.size, values[0])
        i0 = np.searchsorted(self.value_edges, np.log10(at_centres) - self.relative_to, side="right") - 1
        self.values[np.clip(i0, 0, self.values.shape[0] - 1), np.arange(self.depth_centres.size)] += 1


class Inference1D:
    """rjMCMC for one FDEM sounding -- the object the reference's harness creates per data point
    (inversion/Inference3D.py:617-620: ``Inference1D(prng=self.prng, world=self.world, **options)``, ``.initialize(datapoint)``,
    ``failed = .infer(hdf_file_handle=...)``).  The keyword list is the reference's (inversion/Inference1D.py:78-96); further
    keys of the options file (documentation_source/source/supplementary/options_files/resolve_options) arrive through
    ``**kwargs`` exactly as there.  ``engine``: test hook (an object with forward / sensitivity; default: the GPU kernels)."""

    def __init__(self, covariance_scaling=None, high_variance=np.inf, ignore_likelihood=False, interactive_plot=False,
                 low_variance=-np.inf, multiplier=1.0, n_markov_chains=100000, parameter_limits=None, prng=None,
                 reciprocate_parameters=False, reset_limit=1, save_hdf5=False, save_png=False, solve_gradient=True,
                 solve_parameter=False, update_plot_every=5000, world=None, engine=None, **kwargs):
        assert isinstance(prng, np.random.Generator), TypeError("prng must have type np.random.Generator")
        # ignore_likelihood: the prior alone (Inference1D.py:394, 519, 551, 596).  The reference's own run of it ends at the first birth or
        # death (Model.proposal_probabilities calls observation.sensitivity on None, model/Model.py:619); here the observation is left out
        # of the forward AND the reverse proposal (Model.local_precision / local_gradient with observation None): once the starting
        # half-space is chosen from the measured data, the sampler sees a sounding without an active channel -- weights, residual terms,
        # chi^2 and log-likelihood are exact zeros -- and starts burned in (Inference1D.py:388-389).  Same as DeviceChains(ignore_likelihood).
        self.ignore_likelihood = bool(ignore_likelihood)
        self.prng, self.engine, self.world = prng, engine, world
        self.options = dict(OPTION_DEFAULTS)
        self.options.update({k: v for k, v in kwargs.items() if v is not None})
        # covariance_scaling: the harness always passes the options file's value or user_parameters' default 1.0
        # (inversion/user_parameters.py:44); the class default 0.75 of the reference only applies to bare construction
        named = dict(covariance_scaling=covariance_scaling, n_markov_chains=n_markov_chains, parameter_limits=parameter_limits,
                     reset_limit=reset_limit, solve_gradient=solve_gradient, solve_parameter=solve_parameter,
                     update_plot_every=update_plot_every)
        self.options.update({k: v for k, v in named.items() if v is not None})
        self.low_variance, self.high_variance, self.multiplier = low_variance, high_variance, multiplier
        self.save_hdf5, self.save_png, self.interactive_plot = bool(save_hdf5), bool(save_png), bool(interactive_plot)
        self.reciprocate_parameter = reciprocate_parameters
        self.n_markov_chains = int(self.options.get("n_markov_chains", 100000))
        self.iteration, self.accepted = 0, False
        self.on_update = None                      # optional callback(self) after every update of infer()'s schedule

    def initialize(self, datapoint):
        """``datapoint``: geobipy_amd.FdemDataPoint or TdemDataPoint (its data, altitude, geometry and systems are used).  A
        time-domain data point brings its own engine (TdemEngine: the frequency-domain kernels on the spline nodes + the
        window operator) and error model (a relative level per system x component, an additive level per system scaled by
        sqrt(1e-3 / t) per gate: TdemDataPoint.py:361-365); the options then carry lists per level, as the reference's
        skytem_options / tempest_options do."""
        self.datapoint = datapoint
        self.data = np.asarray(datapoint.data, dtype=np.float64)
        self.error_model = datapoint.error_model() if hasattr(datapoint, "error_model") else None
        if self.engine is None:
            self.engine = getattr(datapoint, "engine", None)
        if self.engine is None:
            if hasattr(datapoint, "make_engine"):
                self.engine = datapoint.make_engine(lmax=int(self.options["maximum_number_of_layers"]) + 2)
            else:
                self.engine = GpuEngine(datapoint.system[0], datapoint.z[0])
        self.z_move = height_move_from_options(self.options, float(np.atleast_1d(datapoint.z)[0]))
        # sampled scalars of a time-domain loop pair (solve_transmitter_* / solve_receiver_*: all False in the reference's options files)
        self.geom_moves = datapoint.geometry_moves(**self.options) if hasattr(datapoint, "geometry_moves") else []
        self.priors, self.state = initial_state(self.engine, self.data, self.options, self.error_model, self.z_move, self.geom_moves)
        if self.ignore_likelihood:
            self.observed = self.data
            self.data = np.zeros_like(self.data)               # no active channel from here on (data > 0 is the flag, EmDataPoint.py:54-56)
            self.state.like, self.state.misfit = 0.0, 0.0
        self.halfspace = self.state.values.copy()
        self.iteration = 0
        self.data_misfit_v = np.zeros(2 * self.n_markov_chains + 2)
        self.data_misfit_v[0] = self.state.misfit
        self.acceptance_v = np.zeros(2 * self.n_markov_chains + 2, dtype=np.uint8)
        self.best_state, self.best_posterior, self.best_iteration = self.state, self.posterior, 0
        o = self.options
        self.posteriors = Posteriors(o["maximum_number_of_layers"], o["maximum_depth"], o["minimum_thickness"],
                                     float(self.halfspace[0]), o["factor"],
                                     relative_error_bounds=(o["minimum_relative_error"], o["maximum_relative_error"]),
                                     additive_error_bounds=(o["minimum_additive_error"], o["maximum_additive_error"]),
                                     height_edges=None if self.z_move is None else self.z_move.edges,
                                     geometry_edges={m.name: m.edges for m in self.geom_moves})

    # the quantities the reference exposes on its Inference1D
    @property
    def data_misfit(self):
        return self.state.misfit

    @property
    def prior(self):
        return self.state.prior

    @property
    def likelihood(self):
        return self.state.like

    @property
    def posterior(self):
        return self.state.like + self.state.prior

    @property
    def model(self):
        from .model import Model, RectilinearMesh1D
        return Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, self.state.edges, np.inf]), values=self.state.values)

    def accept_reject(self):
        sp, vp, rp, ap = self.priors
        self.accepted, self.state = rjmcmc.accept_reject(self.prng, self.state, self.data, self.engine, sp, vp, rp, ap,
                                                         self.options["covariance_scaling"], getattr(self, "error_model", None),
                                                         getattr(self, "z_move", None), getattr(self, "geom_moves", None))
        return False

    def update(self):
        """Bookkeeping of Inference1D.update (:705-790): misfit trace, best model, acceptance, posterior histograms (the
        burn-in reset of the reference's schedule lives in the device sampler, rjmcmc_gpu.DeviceChains)."""
        self.iteration += 1
        self.data_misfit_v[self.iteration - 1] = self.state.misfit
        if self.posterior > self.best_posterior:
            self.best_state, self.best_posterior, self.best_iteration = self.state, self.posterior, self.iteration
        self.acceptance_v[self.iteration] = self.accepted
        self.posteriors.update(self.state.edges, self.state.values, self.state.rel, self.state.add, self.state.z, self.state.geom)

    def infer(self, hdf_file_handle=None, n_iterations=None, burn_in_min_iterations=5000):
        """``failed = infer(hdf_file_handle)`` as the harness calls it (Inference3D.py:620, Inference1D.infer :633-688); with
        ``save_hdf5`` and a handle the results are written by ``writeHdf`` (geobipy_amd/hdf.py) when the chain stops.
        ``n_iterations`` given: that many iterations, every state accumulated.  Otherwise the reference's schedule
        (Inference1D.infer :633-688 with update :713-781): the chain burns in at the first iteration >
        ``burn_in_min_iterations`` whose misfit is below the number of active channels -- posteriors and best model restart
        there --, runs ``n_markov_chains`` more iterations, and fails (returns True, like the reference) if it has not
        burned in after ``n_markov_chains`` iterations.  A chain that has not burned in and accepted nothing over
        ``reset_limit`` consecutive windows of ``update_plot_every`` iterations starts over from its initial state
        (the random stream goes on); the third restart arms the reference's (inert) variance limiters and restarts once
        more, and the third restart after that gives the sounding up (returns True)."""
        if isinstance(hdf_file_handle, (int, np.integer)) and n_iterations is None:      # round-1 callers: infer(n_iterations)
            hdf_file_handle, n_iterations = None, int(hdf_file_handle)
        failed = self._infer(n_iterations, burn_in_min_iterations)
        if self.save_hdf5 and hdf_file_handle is not None:
            self.writeHdf(hdf_file_handle)
        return failed

    def writeHdf(self, parent, **kwargs):
        """Inference1D.writeHdf (:1039-1090): this sounding's row of the per-line results container."""
        from . import hdf
        hdf.write_inference1d(parent, self, **kwargs)

    def _infer(self, n_iterations, burn_in_min_iterations):
        if n_iterations is not None:
            for _ in range(n_iterations):
                self.accept_reject()
                self.update()
            return False
        n_active = int((self.data > 0.0).sum())
        window = int(self.options.get("update_plot_every") or 5000)
        reset_limit = int(self.options.get("reset_limit") or 1)
        self.burned_in, self.burned_in_iteration = False, 0
        if self.ignore_likelihood:                             # Inference1D.py:388-389: burned in from the start, 2 n_markov_chains + 1 updates
            self.burned_in, self.burned_in_iteration = True, self.n_markov_chains
        self.n_resets, self.n_zero_acceptance, limited = 0, 0, False
        while True:
            self.accept_reject()
            self.update()
            if not self.burned_in and self.iteration > burn_in_min_iterations and self.data_misfit < n_active:
                self.burned_in, self.burned_in_iteration = True, self.iteration
                self.best_state, self.best_posterior, self.best_iteration = self.state, self.posterior, self.iteration
                self.posteriors.reset()
                self.posteriors.update(self.state.edges, self.state.values, self.state.rel, self.state.add, self.state.z, self.state.geom)
            if window > 1 and self.iteration % window == 0 and not self.burned_in:                    # update :764-776
                # the reference's window: the flags it stored at indices [iteration - window, iteration), i.e. without this step's
                if self.acceptance_v[max(self.iteration - window, 0):self.iteration].sum() == 0:
                    self.n_zero_acceptance += 1
                    if self.n_zero_acceptance == reset_limit:
                        self.reset()
                        self.n_zero_acceptance = 0
                else:
                    self.n_zero_acceptance = 0
            if self.on_update is not None:         # (where the reference's update() returns)
                self.on_update(self)
            if self.burned_in and self.iteration > self.n_markov_chains + self.burned_in_iteration:
                return False
            if not self.burned_in and self.iteration >= self.n_markov_chains:
                return True
            if self.n_resets == 3 and not self.burned_in:                                             # infer :665-678
                if limited:
                    return True
                limited, self.n_resets = True, 0
                self.reset()

    def reset(self):
        """Inference1D.reset (:984-999): back to the initial state of this sounding; the random stream is not rewound."""
        self.n_resets += 1
        self.priors, self.state = initial_state(self.engine, self.data, self.options, getattr(self, "error_model", None),
                                                getattr(self, "z_move", None), getattr(self, "geom_moves", None))
        self.iteration = 0
        self.data_misfit_v[:] = 0.0
        self.data_misfit_v[0] = self.state.misfit
        self.acceptance_v[:] = 0
        self.best_state, self.best_posterior, self.best_iteration = self.state, self.posterior, 0
        self.posteriors.reset()


class BatchedInference:
    """Many soundings in lockstep: same sampler, same per-sounding random streams as independent ``Inference1D`` runs
    (so each chain is the chain the reference would produce from that seed), but the kernel calls of one phase of
    the step are issued as ONE batched launch over all chains:

        phase A  forward + Jacobian at the remapped models of the chains whose structure changed
        phase B  forward at every proposed model
        phase C  Jacobian at the proposed model of the chains that inserted / deleted a layer

    The host logic between the phases is the per-chain code of rjmcmc.accept_reject, re-entered with the phase
    results (a generator-based coroutine per chain keeps the random stream order intact)."""

    def __init__(self, system, heights, data, prngs, **options):
        self.options = dict(OPTION_DEFAULTS)
        self.options.update({k: v for k, v in options.items() if v is not None})
        # what Inference1D takes and this batched twin does not: refused, not silently ignored
        if self.options.get("solve_z", False):
            raise NotImplementedError("BatchedInference: solve_z (the height move) -- use Inference1D or the device sampler (DeviceChains)")
        for key in ("initial_relative_error", "initial_additive_error", "minimum_relative_error", "maximum_relative_error",
                    "minimum_additive_error", "maximum_additive_error", "relative_error_proposal_variance", "additive_error_proposal_variance"):
            if np.ndim(self.options.get(key)) > 0:
                raise NotImplementedError("BatchedInference: per-level error options (" + key + " is a list: time-domain data) -- use Inference1D "
                                          "or the device sampler (TdemDeviceChains)")
        if any((k_.startswith("solve_transmitter_") or k_.startswith("solve_receiver_")) and v_ for k_, v_ in self.options.items()):
            raise NotImplementedError("BatchedInference: loop-pair moves -- use Inference1D or the device sampler")
        self.data = np.asarray(data, dtype=np.float64)
        self.heights = np.asarray(heights, dtype=np.float64)
        self.prngs = list(prngs)
        self.B = self.data.shape[0]
        self.engine = GpuEngine(system, 0.0, lmax=int(self.options["maximum_number_of_layers"]) + 2,
                                exact_jacobian=bool(self.options.pop("exact_jacobian", False)))
        self.states, self.priors = [], []
        o = self.options
        grid = np.logspace(-4.0, 4.0, 100)
        none = np.zeros(0)
        # initialisation: B x 100 half-space forwards in one launch, then B forwards + B Jacobians
        preds = self.engine.forward_many([(none, np.array([c])) for _ in range(self.B) for c in grid],
                                         np.repeat(self.heights, 100)).reshape(self.B, 100, -1)
        sig0 = []
        for b in range(self.B):
            std = np.sqrt((o["initial_relative_error"] * self.data[b]) ** 2.0 + o["initial_additive_error"] ** 2.0)
            phi = [rjmcmc.gauss_loglike(preds[b, i], self.data[b], std)[0] for i in range(100)]
            sig0.append(np.array([grid[int(np.argmin(phi))]]))
        models = [(none, s) for s in sig0]
        P = self.engine.forward_many(models, self.heights)
        J = self.engine.sensitivity_many(models, self.heights)
        for b in range(self.B):
            pr = _priors_from_options(o, sig0[b].item())
            rel, add = o["initial_relative_error"], o["initial_additive_error"]
            std = np.sqrt((rel * self.data[b]) ** 2.0 + add ** 2.0)
            misfit, like = rjmcmc.gauss_loglike(P[b], self.data[b], std)
            prior = rjmcmc.model_log_prior(pr[0], pr[1], none, sig0[b]) + pr[2].log_prior(rel) + pr[3].log_prior(add)
            self.priors.append(pr)
            self.states.append(rjmcmc.ChainState(none, sig0[b], rel, add, P[b], J[b], prior, like, misfit))
        self.accepted = np.zeros(self.B, dtype=bool)
        self.iteration = 0

    def step(self):
        """One rjMCMC iteration of every chain with three batched kernel phases."""
        alpha = self.options["covariance_scaling"]
        gens = [rjmcmc.accept_reject_phases(self.prngs[b], self.states[b], self.data[b], *self.priors[b], alpha)
                for b in range(self.B)]
        requests = [next(g) for g in gens]                      # up to phase A
        for phase in range(3):
            idx = [b for b in range(self.B) if requests[b] is not None and requests[b][0] == phase]
            if idx:
                models = [requests[b][1] for b in idx]
                h = self.heights[idx]
                want_f, want_j = phase in (0, 1), phase in (0, 2)
                F = self.engine.forward_many(models, h) if want_f else [None] * len(idx)
                Jm = self.engine.sensitivity_many(models, h) if want_j else [None] * len(idx)
                for q, b in enumerate(idx):
                    requests[b] = gens[b].send((F[q], Jm[q]))
        for b in range(self.B):                                  # every coroutine has now returned its verdict
            self.accepted[b], self.states[b] = requests[b][1], requests[b][2]
        self.iteration += 1
        return self.accepted
