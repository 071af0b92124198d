"""Reversible-jump moves of the 1-D model -- host-side restatement (work in progress).

So far: the structural move ``RectilinearMesh1D.perturb`` (mesh/RectilinearMesh1D.py:993-1120) with the value
remapping of ``insert_edge`` (:805-839) and ``delete_edge`` (:643-689), consuming a ``numpy.random.Generator`` in
exactly the reference's order, so that a seeded reference run can be replayed
decision by decision (tests/test_rjmcmc.py against tests/golden/mcmc_detail.npz).  The stochastic-Newton value
proposal, the priors and the acceptance ratio are the next pieces; the Jacobian / forward values they consume come from the GPU kernels (or, in the CPU tests, from the oracle).
"""
import numpy as np

NONE, INSERT, DELETE, PERTURB = 0, 1, 2, 3


class StructurePrior:
    """The knobs of the structural move (options file keys of the reference: maximum_number_of_layers,
    minimum_depth, maximum_depth, minimum_thickness, probability_of_birth / death / perturb / no_change)."""

    def __init__(self, max_cells, min_edge, max_edge, min_width, probabilities):
        self.max_cells, self.max_edge, self.min_width = int(max_cells), max_edge, min_width
        # the reference raises the minimum depth to the minimum thickness (RectilinearMesh1D.py:358-360)
        self.min_edge = max(min_edge, min_width)
        p = np.asarray(probabilities, dtype=np.float64)
        self.pmf = np.cumsum(p / p.sum())          # CategoricalDistribution: searchsorted(cumsum(p), u)


def perturb_structure(prng, prior, edges, values, n_tries=10):
    """One structural move.  ``edges``: interior interface depths (k - 1 values, increasing); ``values``: k layer values.

    Returns (action, index, value, new_edges, remapped_values) with the reference's ``mesh.action`` triple:
    ('insert', i, depth) | ('delete', i, depth) | ('perturb', i, dz) | ('none', 0, 0.0), i indexing the full edge
    array [0, e_1 .. e_{k-1}, inf] like the reference.
    """
    edges = np.asarray(edges, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    full = np.r_[0.0, edges, np.inf]
    k = values.size
    while True:
        while True:                                  # RectilinearMesh1D.py:1041-1049
            event = int(np.searchsorted(prior.pmf, prng.uniform(size=1))[0])
            if not ((k == 1 and event in (1, 2)) or (k == prior.max_cells and event == 0)):
                break
        if event == 3:
            return NONE, 0, 0.0, edges.copy(), values.copy()
        if event == 0:                               # birth, :1061-1081
            ok = False
            for tries in range(1, n_tries + 1):
                new_edge = np.exp(prng.uniform(low=np.log(prior.min_edge), high=np.log(prior.max_edge), size=1))[0]
                i = int(np.searchsorted(full, new_edge))
                z = np.insert(full, i, new_edge)
                ok = np.min(np.diff(z)) > prior.min_width and tries < n_tries
                if ok or tries == n_tries:
                    break
            if ok:
                return INSERT, i, float(new_edge), z[1:-1], np.insert(values, i, values[i - 1])
            continue                                 # 10 failed tries: draw a new event
        if event == 1:                               # death, :1083-1087
            i = int(np.int64(prng.uniform(low=0, high=k - 1, size=1)[0])) + 1
            merged = 0.5 * (values[i - 1] + values[i])
            v = np.delete(values, i)
            v[i - 1] = merged
            return DELETE, i, float(full[i]), np.delete(full, i)[1:-1], v
        ok = False                                   # perturb, :1089-1118
        for tries in range(1, n_tries + 1):
            z = full.copy()
            i = int(np.int32(prng.uniform(low=1, high=full.size - 1, size=1)[0]))
            dz = np.sign(prng.normal()) * prior.min_width * prng.uniform()
            z[i] += dz
            ok = (np.min(np.diff(z)) > prior.min_width and z[1] > prior.min_edge and z[-2] < prior.max_edge
                  and tries < n_tries)
            if ok or tries == n_tries:
                break
        if ok:
            return PERTURB, i, float(dz), z[1:-1], values.copy()


class ValuePrior:
    """Priors on the layer conductivities as Inference1D.initialize_model sets them (inversion/Inference1D.py:497-509,
    model/Model.py:727-746): log-normal on the values, mean = best half-space, variance = ln(1 + factor)^2; and a
    normal prior on the vertical gradient of ln(sigma) with standard deviation gradient_standard_deviation."""

    def __init__(self, value_mean, factor=10.0, gradient_std=1.5, solve_gradient=True, solve_value=False, limits=None):
        self.log_mean = np.log(value_mean)
        self.value_precision = 1.0 / np.log(1.0 + factor) ** 2.0
        self.gradient_precision = 1.0 / gradient_std ** 2.0
        self.solve_gradient = solve_gradient
        # solve_parameter of the options file: the log-normal prior on the values also enters the model probability
        # (it always shapes the stochastic-Newton step: Inference1D.py:503 sets it regardless)
        self.solve_value = solve_value
        # parameter_limits of the options file: proposals with a conductivity outside [lo, hi] have zero prior probability
        # (Model.probability :555-558, a log-uniform Distribution used only as a bound)
        self.limits = None if limits is None else (float(limits[0]), float(limits[1]))


def gradient_operator(edges):
    """RectilinearMesh1D.gradient_operator (mesh/RectilinearMesh1D.py:747-786) for a mesh [0, edges..., inf]."""
    k = edges.size + 1
    if k == 1:
        return np.ones((1, 1))
    full = np.r_[0.0, edges, np.inf]
    x = np.abs(np.diff(full))
    e2e = full[-2] - full[0]
    x[-1] = x[0] if k == 2 else x[-2] + e2e
    tmp = 1.0 / (0.5 * (x[:-1] + x[1:]) * (k - 1))
    out = np.zeros((k - 1, k))
    idx = np.arange(k - 1)
    out[idx, idx] = -tmp
    out[idx, idx + 1] = tmp
    return out


def model_prior_derivative(vp, edges, values, order):
    """Model.prior_derivative (model/Model.py:421-430): Wm'Wm (order 2) or Wm'Wm (ln sigma - ln sigma_ref) (order 1)."""
    k = values.size
    op = np.eye(k) * vp.value_precision
    if vp.solve_gradient:
        Wz = gradient_operator(edges)
        op = op + Wz.T @ (np.eye(max(1, k - 1)) * vp.gradient_precision) @ Wz
    return op if order == 2 else op @ (np.log(values) - vp.log_mean)


def stochastic_newton(vp, edges, values, J, predicted, data, std, alpha=1.0):
    """Mean and covariance of the log-normal value proposal (Model.stochastic_newton_perturbation,
    model/Model.py:368-419; DataPoint.prior_derivative, data/datapoint/DataPoint.py:340-349):
        H = inv(J' Wd'Wd J + Wm'Wm),  g = J' Wd'Wd (pred - obs) + Wm'Wm (ln sigma - ln sigma_ref),
        mean = exp(ln sigma - alpha H g)
    J: d pred / d ln sigma [N, k] at the remapped model (or the stale one the datapoint carries when the
    structure did not change), active channels only = data > 0."""
    active = data > 0.0
    Ja = J[active]
    P = 1.0 / std[active] ** 2.0
    op = model_prior_derivative(vp, edges, values, 2)
    hess = op + Ja.T @ (P[:, None] * Ja)
    grad = op @ (np.log(values) - vp.log_mean) + Ja.T @ (P * (predicted[active] - data[active]))
    H = np.linalg.inv(hess)
    pk = -(H @ grad)
    return np.exp(np.log(values) + alpha * pk), H


def propose_values(prng, mean, H):
    """MvLogNormal(mean, H, linearSpace=True).rng (statistics/MvNormalDistribution.py:179-181)."""
    return np.exp(np.atleast_1d(np.squeeze(prng.multivariate_normal(np.log(mean), H, size=1))))


# ------------------------------------------------------------------------------------------------------------
# data-error moves, priors, proposal densities, and one full accept/reject step
# ------------------------------------------------------------------------------------------------------------
class ErrorPrior:
    """Log-uniform prior [lo, hi] and log-normal random-walk proposal of one error level
    (DataPoint.set_priors / set_proposals, data/datapoint/DataPoint.py:575-644) -- or, with array arguments, of the error
    levels of a time-domain data point (one relative level per system x component, one additive level per system:
    TdemDataPoint.py:361-365), which the reference proposes and redraws JOINTLY (one multivariate draw, StatArray.propose)."""

    def __init__(self, lo, hi, proposal_variance):
        self.lo, self.hi = np.log(lo), np.log(hi)
        self.var = proposal_variance if np.ndim(lo) == 0 else np.broadcast_to(np.asarray(proposal_variance, dtype=np.float64), np.shape(lo)).copy()

    def log_prior(self, x):
        lx = np.log(x)
        if np.ndim(lx) > 0:
            return -np.sum(np.log(self.hi - self.lo)) if np.all((self.lo <= lx) & (lx <= self.hi)) else -np.inf
        return -np.log(self.hi - self.lo) if (self.lo <= lx <= self.hi) else -np.inf

    def propose(self, prng, current, redraw=True, centre=None):
        """StatArray.propose with imposePrior=True, log=True (statistics/StatArray.py:578-638).  The reference draws
        with Generator.multivariate_normal on a 1 x 1 covariance, which numpy evaluates as
        mean + standard_normal() * sqrt(var) (SVD of a positive 1 x 1 matrix: s = var, vh = 1).
        ``redraw=False``: a single draw whatever the prior says (StatArray.perturb() without imposePrior -- the additive-error
        multiplier of Tempest_datapoint.perturb :339-341); ``centre``: the proposal's mean when it is not the current value."""
        if np.ndim(current) > 0:                 # several levels: the reference's own call, one joint draw per try
            cur = np.asarray(current, dtype=np.float64)
            mean = cur if centre is None else np.asarray(centre, dtype=np.float64)
            draw_n = lambda: np.exp(prng.multivariate_normal(np.log(mean), np.diag(np.broadcast_to(self.var, cur.shape)), size=1)[0])
            x = draw_n()
            if not redraw:
                return x
            tries = 0
            while self.log_prior(x) == -np.inf:
                x = draw_n()
                tries += 1
                if tries == 10:
                    return cur.copy()
            return x
        lc, sd = np.log(current), np.sqrt(self.var)
        draw = lambda: float(np.exp(lc + prng.standard_normal() * sd))
        x = draw()
        tries = 0
        while self.log_prior(x) == -np.inf:
            x = draw()
            tries += 1
            if tries == 10:
                return current
        return x


class ScalarMove:
    """One sampled scalar of the acquisition geometry with the reference's uniform prior ``centre +- max_change`` and random-walk
    proposal (Point / EmLoop.set_priors / set_proposals / perturb: pointcloud/Point.py:614-621, 949-983, system/EmLoop.py:222-305):
    the data point's height (``HeightMove``) or an attitude angle / offset of the loop pair.  See HeightMove for the proposal's
    scale quirk; ``n_bins``: cells of the posterior histogram on the prior's support (99; EmLoop's pitch: 199, EmLoop.py:326)."""

    def __init__(self, name, centre, max_change, proposal_variance, n_bins=99):
        self.name = name
        self.z0, self.lo, self.hi = float(centre), float(centre) - float(max_change), float(centre) + float(max_change)
        self.scale = float(proposal_variance)
        self.edges = np.linspace(self.lo, self.hi, n_bins + 1)

    def log_prior(self, z):
        return -np.log(self.hi - self.lo) if (self.lo <= z <= self.hi) else -np.inf      # scipy uniform.logpdf: closed support

    def propose(self, prng, current):
        draw = lambda: float(prng.normal(current, self.scale))
        x = draw()
        tries = 0
        while self.log_prior(x) == -np.inf:
            x = draw()
            tries += 1
            if tries == 10:
                return float(current)
        return x


class HeightMove(ScalarMove):
    """The height move of a data point (``solve_z``; Point.set_priors / set_proposals / perturb, pointcloud/Point.py:614-621,
    949-983): uniform prior ``z0 +- maximum_z_change``, random-walk proposal ``Normal(z, z_proposal_variance)`` redrawn up to 10
    times while the prior gives -inf, then the current height is kept (StatArray.propose, statistics/StatArray.py:578-638).
    One quirk is reproduced: NormalDistribution.rng hands the VARIANCE to numpy as the scale
    (statistics/NormalDistribution.py:111: ``prng.normal(mean, variance, size)``), so the step's standard deviation is
    ``z_proposal_variance`` itself, not its square root.  (No options file of the reference sets these keys -- its ``solve_height`` /
    ``maximum_height_change`` / ``height_proposal_variance`` are never read -- so the move is off unless a user adds them.)"""

    def __init__(self, z0, max_change, proposal_variance, n_bins=99):
        super().__init__("z", z0, max_change, proposal_variance, n_bins)          # Point.set_z_posterior: Uniform.bins() = 99 cells


def mvn_logpdf(x, mean, cov):
    """MvNormal.probability(log=True) (statistics/MvNormalDistribution.py:201-216)."""
    d = x - mean
    sign, logdet = np.linalg.slogdet(cov)
    return -(0.5 * x.size) * np.log(2.0 * np.pi) - 0.5 * sign * logdet - 0.5 * d @ (np.linalg.inv(cov) @ d)


def model_log_prior(sp, vp, edges, values):
    """Model.probability(solve_value, solve_gradient) (model/Model.py:533-575): uniform prior on the
    number of layers (mesh/RectilinearMesh1D.py:1351-1382; the order-statistics prior on the interfaces is
    commented out in the reference) + normal prior on the vertical gradient of ln sigma (Model.py:213-234)."""
    k = values.size
    if vp.limits is not None and (np.any(values < vp.limits[0]) or np.any(values > vp.limits[1])):
        return -np.inf
    lp = -np.log(sp.max_cells - 1.0)
    if vp.solve_value:
        lp += mvn_logpdf(np.log(values), np.full(k, vp.log_mean), np.eye(k) / vp.value_precision)
    if not vp.solve_gradient:
        return lp
    if k == 1:   # the reference evaluates a two-layer copy of the half-space: zero gradient
        g = np.zeros(1)
    else:
        widths = np.diff(np.r_[0.0, edges])
        g = np.diff(np.log(values)) / np.log(widths)
    return lp + mvn_logpdf(g, np.zeros(g.size), np.eye(g.size) / vp.gradient_precision)


def gauss_loglike(pred, data, std):
    """chi^2 and log-likelihood over the active (data > 0) channels -- host-side twin of gbp_gauss_loglike for the
    engines that do not return them (DataPoint.py:491-525)."""
    a = data > 0.0
    r = (pred[a] - data[a]) / std[a]
    chi2 = float(np.sum(r * r))
    return chi2, float(-(0.5 * a.sum()) * np.log(2.0 * np.pi) - np.sum(np.log(std[a])) - 0.5 * chi2)


class ErrorModel:
    """Standard deviation of every channel from the error levels: DataPoint.std (data/datapoint/DataPoint.py:268-282: one
    relative and one additive level) or, with channel -> level maps, TdemDataPoint.std (TdemDataPoint.py:361-365:
    sigma_i^2 = (rel_{system, component} d_i)^2 + (add_system sqrt(1e-3 / t_i))^2)."""

    def __init__(self, rel_group=None, add_group=None, add_scale=None, stale_prediction=False, tempest=False, add_centre=None):
        # stale_prediction: the reference's TdemDataPoint.fm_dlogc (TdemDataPoint.py:1031-1055) stores the Jacobian of the
        # remapped model but NOT its prediction (the assignment is commented out there), so the stochastic-Newton gradient of a
        # time-domain chain is formed with the prediction of the CURRENT model (Model.py:383-399) -- reproduced on request
        self.stale_prediction = bool(stale_prediction)
        # tempest: the additive levels are Tempest_datapoint's MULTIPLIERS of per-channel additive errors (``add_scale``), which
        # the reference treats unlike every other level (data/datapoint/Tempest_datapoint.py:339-341, 475-487): proposed last --
        # after the loop pair --, with a single draw (no redraw against the prior), and their prior never enters
        # Tempest_datapoint.probability (set_priors hands DataPoint.set_priors solve_additive_error = False); and their proposal
        # stays centred on the values it was created with (``add_centre``): Tempest_datapoint.perturb does not move the proposal's
        # mean to the new state as DataPoint.perturb does for the other levels -- an independence sampler around the initial
        # multipliers, not a random walk.  All four reproduced (tests/golden/mcmc_trace_tempest.npz)
        self.tempest = bool(tempest)
        self.add_centre = None if add_centre is None else np.asarray(add_centre, dtype=np.float64).copy()
        self.rel_group = None if rel_group is None else np.asarray(rel_group, dtype=np.int64)
        self.add_group = None if add_group is None else np.asarray(add_group, dtype=np.int64)
        self.add_scale = None if add_scale is None else np.asarray(add_scale, dtype=np.float64)

    def std(self, data, rel, add):
        if self.rel_group is None:
            return np.sqrt((rel * data) ** 2.0 + add ** 2.0)
        rel, add = np.atleast_1d(rel), np.atleast_1d(add)
        return np.sqrt((rel[self.rel_group] * data) ** 2.0 + (add[self.add_group] * self.add_scale) ** 2.0)


_PLAIN_ERRORS = ErrorModel()


class ChainState:
    """What Inference1D carries between iterations for one sounding."""

    def __init__(self, edges, values, rel, add, pred, J, prior, like, misfit, z=None, geom=None):
        self.z = None if z is None else float(z)               # the data point's height, when it is sampled (HeightMove)
        self.geom = None if geom is None else dict(geom)       # sampled scalars of the loop pair by name (ScalarMove)
        self.edges, self.values = np.array(edges, dtype=np.float64), np.array(values, dtype=np.float64)
        self.rel = float(rel) if np.ndim(rel) == 0 else np.array(rel, dtype=np.float64)
        self.add = float(add) if np.ndim(add) == 0 else np.array(add, dtype=np.float64)
        self.pred, self.J = np.array(pred, dtype=np.float64), np.array(J, dtype=np.float64)
        self.prior, self.like, self.misfit = float(prior), float(like), float(misfit)

    @property
    def k(self):
        return self.values.size


def accept_reject_phases(prng, state, data, sp, vp, rel_prior, add_prior, alpha=1.0, error_model=None, z_move=None, geom_moves=None):
    """One iteration of Inference1D.accept_reject (inversion/Inference1D.py:537-631) for the Resolve-style option
    set (solve_gradient, solve relative / additive error, no height move), written as a coroutine around the hot
    path: it yields ``(phase, (edges, values))`` whenever it needs the kernels --
        phase 0: forward + Jacobian at the remapped model (fm_dlogc, Model.py:383-384)
        phase 1: forward at the proposed model (Inference1D.py:572)
        phase 2: Jacobian at the proposed model (reversible-jump step, Model.py:612)
    -- is sent back ``(predicted, J)``, and finally yields ``(3, accepted, state)``.  The random stream is consumed
    in exactly the reference's order, so a driver may interleave the coroutines of many chains and batch each
    phase into one launch.  With ``z_move`` (HeightMove) and / or ``geom_moves`` (ScalarMove list: sampled angles / offsets of a
    time-domain loop pair, proposed after the error levels as TdemDataPoint.perturb does, TdemDataPoint.py:681-683) the geometry is
    sampled too and the requests carry it as a third entry, ``(phase, (edges, values), {"z": z, "geometry": {name: value}})``:
    the current geometry for phase 0 -- Model.perturb works on the unperturbed copy of the data point, Inference1D.py:547-560 --,
    the proposed one for phases 1 and 2."""
    prng.random()                                               # Inference1D.py:542
    action, _, _, edges, rem = perturb_structure(prng, sp, state.edges, state.values)
    moving = z_move is not None or bool(geom_moves)

    def ask(ph, e, v, z, geom):
        if not moving:
            return (ph, (e, v))
        kw = {}
        if z_move is not None:
            kw["z"] = z
        if geom_moves:
            kw["geometry"] = dict(geom)
        return (ph, (e, v), kw)
    if action != NONE:
        pred_rem, J = yield ask(0, edges, rem, state.z, state.geom)
    else:
        pred_rem, J = state.pred, state.J
    em = _PLAIN_ERRORS if error_model is None else error_model
    if em.stale_prediction:
        pred_rem = state.pred
    std = em.std(data, state.rel, state.add)
    mean, H = stochastic_newton(vp, edges, rem, J, pred_rem, data, std, alpha)
    prop = propose_values(prng, mean, H)
    z = state.z if z_move is None else z_move.propose(prng, state.z)     # Point.perturb first (DataPoint.perturb :561) ...
    rel = rel_prior.propose(prng, state.rel)                    # ... then the error levels, DataPoint.py:531-573
    add = state.add if em.tempest else add_prior.propose(prng, state.add)
    geom = state.geom
    if geom_moves:                                              # Loop_pair.perturb (system/Loop_pair.py:161-164), after the error levels
        geom = {m.name: m.propose(prng, state.geom[m.name]) for m in geom_moves}
    if em.tempest:                                              # Tempest_datapoint.perturb :339-341: the multipliers, last, one draw
        add = add_prior.propose(prng, state.add, redraw=False, centre=em.add_centre)
    pred, _ = yield ask(1, edges, prop, z, geom)
    std_t = em.std(data, rel, add)
    misfit, like = gauss_loglike(pred, data, std_t)
    prior = rel_prior.log_prior(rel) + (0.0 if em.tempest else add_prior.log_prior(add))
    if z_move is not None:
        prior += z_move.log_prior(z)                            # Point.probability, pointcloud/Point.py:159-197
    if geom_moves:
        prior += sum(m.log_prior(geom[m.name]) for m in geom_moves)          # Loop_pair.probability, system/Loop_pair.py:294-295
    if prior == -np.inf:
        yield (3, False, state)
        return
    prior += model_log_prior(sp, vp, edges, prop)
    if prior == -np.inf:
        yield (3, False, state)
        return
    q_fwd = q_rev = 1.0
    if action in (INSERT, DELETE):                              # Model.proposal_probabilities, Model.py:577-659
        _, J = yield ask(2, edges, prop, z, geom)
        a = data > 0.0
        grad = model_prior_derivative(vp, edges, prop, 1) + J[a].T @ ((pred[a] - data[a]) / std_t[a] ** 2.0)
        # ln sigma' - alpha * pk with pk = -H g; the reference exponentiates in long double (expReal,
        # base/utilities.py:827-856), so the reverse mean only degenerates beyond exp(+-11356)
        mean_r = np.exp(np.longdouble(1.0) * (np.log(prop) + alpha * (H @ grad)))
        if np.any(np.isinf(mean_r)) or np.any(mean_r == 0.0):
            q_fwd = q_rev = -np.inf
        else:
            q_fwd = mvn_logpdf(np.log(rem), np.log(mean_r).astype(np.float64), H)
            q_rev = mvn_logpdf(np.log(prop), np.log(rem), H)
    log_ratio = (prior - state.prior) + (like - state.like) + (q_fwd - q_rev)
    # expReal(log_ratio) > U(0,1): inf above 11356, NaN (from -inf - -inf) compares False
    with np.errstate(invalid="ignore"):
        ratio = np.inf if log_ratio > 11356.0 else np.exp(np.longdouble(log_ratio))
        accepted = bool(ratio > prng.uniform())
    if not accepted:
        yield (3, False, state)
        return
    yield (3, True, ChainState(edges, prop, rel, add, pred, J, prior, like, misfit, z, geom))


def accept_reject(prng, state, data, engine, sp, vp, rel_prior, add_prior, alpha=1.0, error_model=None, z_move=None, geom_moves=None):
    """Drive ``accept_reject_phases`` for one chain with ``engine.forward(edges, values)`` /
    ``engine.sensitivity(edges, values)`` (GPU kernels in the product; with a sampled geometry both take ``z=`` / ``geometry=``).
    Returns (accepted, state)."""
    g = accept_reject_phases(prng, state, data, sp, vp, rel_prior, add_prior, alpha, error_model, z_move, geom_moves)
    req = next(g)
    while req[0] != 3:
        phase, (e, v) = req[0], req[1]
        kw = req[2] if len(req) > 2 else {}
        F = engine.forward(e, v, **kw) if phase in (0, 1) else None
        Jm = engine.sensitivity(e, v, **kw) if phase in (0, 2) else None
        req = g.send((F, Jm))
    return req[1], req[2]
