"""Device-resident rjMCMC: every chain of a block of soundings lives in HBM and advances in lockstep.

Same statistical model as ``rjmcmc.py`` (the host restatement that reproduces the reference's chains decision by
decision): same priors, same four structural moves with the same validity rules and retry limits, same
stochastic-Newton log-normal value proposal, same error moves, same reversible-jump ratio and Metropolis test
(reference: inversion/Inference1D.py:537-631 and the methods it calls -- see include/geobipy_amd.h).  What differs
is where the random numbers come from: a counter-based generator keyed by (seed, chain, iteration) on the device
instead of one numpy generator per sounding consumed in the reference's call order.  A chain here is therefore a
draw from the same Markov kernel but not the *same* draw as the reference's chain for a numpy seed -- that is what
``inference.Inference1D`` / ``BatchedInference`` are for; tests/test_rjmcmc_gpu.py checks the stages against a host
emulation with the same draws, and the ensembles against the host sampler.

``DeviceChains`` owns the buffers (torch tensors) and hands their addresses to ``gbp_rj_run``, which issues the ten
launches of one iteration (propose | 2 x prediction + Jacobian of the remapped models | newton (packed, general) | fused forward +
likelihood of the proposals that keep their dimension | 2 x prediction + Jacobian of those that change it | accept (packed, general)) on the caller's stream without
synchronising; nothing crosses PCIe between iterations.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

NONE, INSERT, DELETE, PERTURB = 0, 1, 2, 3
LOG_2PI = math.log(2.0 * math.pi)


def layer_widths(edges, k):
    """Finite layer thicknesses [B, K]: w_j = e_j - e_{j-1} for j < k - 1, 0 elsewhere.  edges: [B, K] (+inf padded)."""
    B, K = edges.shape
    top = torch.cat([torch.zeros(B, 1, dtype=edges.dtype, device=edges.device), edges[:, :-1]], dim=1)
    j = torch.arange(K, device=edges.device)[None, :]
    return torch.where(j < (k[:, None] - 1), edges - top, torch.zeros_like(edges))


def model_log_prior(edges, sigma, k, max_cells, gradient_precision, solve_gradient=True, value_precision=None, log_mean=None):
    """Batched rjmcmc.model_log_prior (Model.probability, model/Model.py:533-575); the prior on the values is included
    when ``value_precision`` / ``log_mean`` are given (solve_parameter)."""
    lp = torch.full((sigma.shape[0],), -math.log(max_cells - 1.0), dtype=sigma.dtype, device=sigma.device)
    if value_precision is not None:
        j = torch.arange(sigma.shape[1], device=sigma.device)[None, :]
        d = torch.where(j < k[:, None], torch.log(sigma) - log_mean[:, None], torch.zeros_like(sigma))
        kk = k.to(sigma.dtype)
        lp = lp - 0.5 * kk * LOG_2PI + 0.5 * kk * math.log(value_precision) - 0.5 * value_precision * (d * d).sum(dim=1)
    if not solve_gradient:
        return lp
    w = layer_widths(edges, k)[:, :-1]
    je = torch.arange(w.shape[1], device=w.device)[None, :]
    live = je < k[:, None] - 1
    g = torch.where(live, (torch.log(sigma[:, 1:]) - torch.log(sigma[:, :-1])) / torch.log(torch.where(live, w, torch.full_like(w, math.e))),
                    torch.zeros_like(w))
    n = (k - 1).clamp(min=1).to(sigma.dtype)
    return lp - 0.5 * n * LOG_2PI + 0.5 * n * math.log(gradient_precision) - 0.5 * gradient_precision * (g * g).sum(dim=1)


def log_uniform_prior(x, lo, hi):
    lx = torch.log(x)
    inside = (lx >= math.log(lo)) & (lx <= math.log(hi))
    return torch.where(inside, torch.full_like(x, -math.log(math.log(hi) - math.log(lo))), torch.full_like(x, float("-inf")))


class DeviceChains:
    """B rjMCMC chains (one per sounding) resident on one GPU.  ``options``: keys of the reference's options file
    (documentation_source/source/supplementary/options_files/resolve_options).

    ``first_chain``: global index of the block's first sounding -- the random streams are keyed by the global chain
    index, so a survey produces the same chains on 1 GPU or sharded over 8, bit for bit.  ``forward_waves``: the forward
    kernels' waves per workgroup, a performance hint only (the Hankel sums are reduced per 64-point pass and added in pass
    order, whatever wave ran the pass); the default, 2, is the best single choice from 1k to 64k chains per GPU, 0 lets the
    library choose from the block size.
    ``reference_schedule``: per-sounding burn-in / stop rule of the reference (Inference1D.update :713-737, infer
    :641-688) evaluated on the device: a chain burns in at the first iteration > ``burn_in_min_iterations`` whose misfit
    is below the number of active channels (its posteriors and best model restart there), is done ``n_markov_chains``
    iterations later and has failed if it has not burned in after ``n_markov_chains`` iterations; see ``infer``.
    ``hankel_eps_ppm``: every chain evaluates only the filter abscissae whose terms can add up to more than that many ppm
    at its own sounding's altitude (1 m altitude bins, so the result does not depend on the sharding; default 1e-10 keeps
    about 600 of the 1200 points of the 10-frequency system; predictions and true-derivative Jacobians move by less than
    that bound); 0 evaluates all 120 / 140 abscissae.  ``min_altitude`` is accepted and ignored.
    ``chain_id`` (int64[B], optional): the key of every chain's random streams (default ``first_chain + row``) -- pass the
    soundings' global indices when a block holds a non-contiguous selection of a survey.
    hitmap=True also accumulates the conductivity-depth hit map, int32[B, n_value_bins, n_depth_bins] (440 KB per
    sounding with the default grids: 29 GB for 65536 soundings -- sized for 288 GB of HBM)."""

    def __init__(self, system, heights, data, seed=0, exact_jacobian=False, device=None, hitmap=False, n_value_bins=250,
                 first_chain=0, forward_waves=2, reference_schedule=False, burn_in_min_iterations=5000, hankel_eps_ppm=None,
                 min_altitude=None, add_scale=None, rel_group=None, add_group=None, chain_id=None, extra_log_prior=0.0,
                 additive_independent=False, trace_every=0, trace_length=None, ignore_likelihood=False, **options):
        """``ignore_likelihood``: sample the PRIOR alone (the reference's option of that name, Inference1D.py:394, 519, 551, 596: no data
        term in the stochastic-Newton step, likelihood constant).  The reference's own run of it ends at the first birth or death, where
        Model.proposal_probabilities calls ``observation.sensitivity`` on None (model/Model.py:619); here the observation is left out
        consistently -- Model.local_precision / local_gradient with ``observation is None`` for the forward AND the reverse proposal -- by
        handing the sampler a block whose channels are all inactive once the starting half-space has been chosen from the measured data:
        weights, residual terms, chi^2 and log-likelihood are then exact zeros in every stage (the same kernels, nothing skipped).  The
        chains start burned in with ``burned_in_iteration = n_markov_chains`` (Inference1D.py:388-389): 2 n_markov_chains + 1 updates,
        posteriors from the first.  The measured data stay available as ``observed``.

        ``trace_every`` > 0: keep every ``trace_every``-th entry of the reference's per-iteration arrays ``data_misfit_v`` /
        ``acceptance_v`` (Inference1D.py:408, 414) on the device -- ``trace_misfit`` [B, trace_length] (NaN = not reached),
        ``trace_accept`` uint8 [B, trace_length]; ``trace_length`` defaults to the reference's 2 n_markov_chains / trace_every (needs
        n_markov_chains).  1 = the reference's arrays in full."""
        from .inference import OPTION_DEFAULTS
        o = dict(OPTION_DEFAULTS)
        o.update({k: v for k, v in options.items() if v is not None})
        self.o = o
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.system = system
        f64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).to(self.device).contiguous()
        data, heights = f64(data), f64(heights)
        with torch.cuda.device(self.device):
            from .system import DEFAULT_HANKEL_EPS_PPM
            self.hankel_eps_ppm = DEFAULT_HANKEL_EPS_PPM if hankel_eps_ppm is None else float(hankel_eps_ppm)
            # the height move (keys the reference's data point reads: solve_z, maximum_z_change, z_proposal_variance --
            # pointcloud/Point.py:949-983; geobipy_amd.rjmcmc.HeightMove is the host twin)
            self.solve_height = bool(o.get("solve_z", False))
            self.height_half_width = float(o["maximum_z_change"]) if self.solve_height else 0.0
            if self.hankel_eps_ppm > 0.0 and heights.numel() > 0 and hasattr(system, "handle_binned"):   # per-chain abscissa window, as FdemBatch
                self._h = system.handle_binned(self.hankel_eps_ppm, float(heights.min()) - self.height_half_width,
                                               float(heights.max()) + self.height_half_width)
            else:
                self._h = system.handle()
        self.B, self.N = data.shape
        self.K = K = int(o["maximum_number_of_layers"])
        self.min_width = float(o["minimum_thickness"])
        self.min_edge = max(float(o["minimum_depth"]), self.min_width)      # RectilinearMesh1D.py:358-360
        self.max_edge = float(o["maximum_depth"])
        assert self.min_width * self.K < self.max_edge, ValueError(          # RectilinearMesh1D.set_priors :1523
            "minimum_thickness * maximum_number_of_layers = {} is bigger than maximum_depth {}".format(self.min_width * self.K,
                                                                                                       self.max_edge))
        self.gradient_precision = 1.0 / o["gradient_standard_deviation"] ** 2
        p = np.array([o["probability_of_birth"], o["probability_of_death"], o["probability_of_perturb"],
                      o["probability_of_no_change"]], dtype=np.float64)
        p = p / p.sum()
        # posterior grids: the reference's interface-depth grid (RectilinearMesh1D.set_posteriors :1438-1455) and a
        # log10 conductivity axis of +-4 prior standard deviations about the prior mean (Model.set_posteriors)
        self.depth_bin_width = 0.5 * self.min_width
        self.n_depth_bins = max(1, np.arange(0.0, 1.1 * self.max_edge, self.depth_bin_width).size - 1)   # cells between the reference's edges
        self.n_value_bins = int(n_value_bins)
        self.value_half_width = 4.0 * math.log(1.0 + o["factor"]) / math.log(10.0)
        ro = _lib.RjOptions()
        ro.max_layers, ro.n_channels = K, self.N
        ro.solve_gradient = int(bool(o["solve_gradient"]))
        ro.solve_value = int(bool(o.get("solve_parameter", False)))
        ro.solve_relative_error = int(bool(o.get("solve_relative_error", True)))
        ro.solve_additive_error = int(bool(o.get("solve_additive_error", True)))
        ro.exact_jacobian = int(bool(exact_jacobian))
        ro.n_depth_bins, ro.n_value_bins = self.n_depth_bins, self.n_value_bins
        ro.n_error_bins = self.n_error_bins = 99                    # DataPoint.set_posteriors: Uniform.bins(nBins=99)
        ro.min_edge, ro.max_edge, ro.min_width = self.min_edge, self.max_edge, self.min_width
        ro.p_birth, ro.p_death, ro.p_perturb, ro.p_none = p
        ro.value_precision = 1.0 / math.log(1.0 + o["factor"]) ** 2
        ro.gradient_precision = self.gradient_precision
        lim = o.get("parameter_limits")
        ro.value_min, ro.value_max = (0.0, 0.0) if lim is None else (float(lim[0]), float(lim[1]))
        ro.alpha = float(o["covariance_scaling"])
        # error levels: one of each for frequency-domain data; time-domain data have one relative level per system x
        # component and one additive level per system (options given as lists, as in the reference's skytem / tempest files)
        def vec(key, n):
            if o.get(key) is None:
                raise KeyError("option '{}' is required (error-level bounds and proposal variances are read even when the "
                               "level is not solved for: the prior of the fixed level uses them)".format(key))
            return np.broadcast_to(np.atleast_1d(np.asarray(o[key], dtype=np.float64)), (n,)).copy()
        self.n_rel_groups = Gr = 1 if rel_group is None else int(np.max(rel_group)) + 1
        self.n_add_groups = Ga = 1 if add_group is None else int(np.max(add_group)) + 1
        ro.n_rel_groups, ro.n_add_groups = Gr, Ga
        for name, key, n, f in (("rel_min", "minimum_relative_error", Gr, None), ("rel_max", "maximum_relative_error", Gr, None),
                                ("rel_sd", "relative_error_proposal_variance", Gr, np.sqrt), ("add_min", "minimum_additive_error", Ga, None),
                                ("add_max", "maximum_additive_error", Ga, None), ("add_sd", "additive_error_proposal_variance", Ga, np.sqrt)):
            v = vec(key, n)
            setattr(ro, name, (ctypes.c_double * 4)(*(list(f(v) if f else v) + [1.0] * (4 - n))))
        self._rel0, self._add0 = vec("initial_relative_error", Gr), vec("initial_additive_error", Ga)
        self._bounds = dict(rel=(vec("minimum_relative_error", Gr), vec("maximum_relative_error", Gr)),
                            add=(vec("minimum_additive_error", Ga), vec("maximum_additive_error", Ga)))
        ro.depth_bin_width, ro.value_half_width = self.depth_bin_width, self.value_half_width
        ro.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        ro.first_chain = int(first_chain)          # global index of this block's first sounding (sharded surveys)
        ro.forward_waves = int(forward_waves)
        ro.schedule = int(bool(reference_schedule))
        ro.burn_in_min_iterations = int(burn_in_min_iterations)
        ro.n_markov_chains = int(o.get("n_markov_chains", 0))
        assert not reference_schedule or ro.n_markov_chains > 0, ValueError("reference_schedule needs n_markov_chains")
        ro.solve_height = int(self.solve_height)
        ro.height_half_width = self.height_half_width
        # (NormalDistribution.rng hands the variance to numpy as the scale, statistics/NormalDistribution.py:111: reproduced)
        ro.height_scale = float(o["z_proposal_variance"]) if self.solve_height else 0.0
        ro.extra_log_prior = float(extra_log_prior)     # (priors of sampled scalars that live outside gbp_rj_chains: gbp_td_moves)
        # Tempest's additive-error multipliers as the reference samples them (gbp_rj_options.additive_independent)
        ro.additive_independent = int(bool(additive_independent))
        ro.add_centre = (ctypes.c_double * 4)(*(list(self._add0) + [1.0] * (4 - Ga)))
        self.trace_every = int(trace_every or 0)
        if self.trace_every > 0:
            if trace_length is None:
                assert ro.n_markov_chains > 0, ValueError("trace_every needs trace_length or n_markov_chains (the reference's arrays hold 2 n_markov_chains entries)")
                trace_length = -(-2 * ro.n_markov_chains // self.trace_every)
            ro.trace_every, ro.trace_length = self.trace_every, int(trace_length)
        self.trace_length = int(ro.trace_length)
        self._o = ro
        B, N, dev = self.B, self.N, self.device
        z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
        i32, i64 = torch.int32, torch.int64
        i32v = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev).contiguous()
        self.t = t = dict(
            rel_group=i32v(rel_group), add_group=i32v(add_group),
            add_scale=None if add_scale is None else f64(add_scale),
            chain_id=None if chain_id is None else torch.as_tensor(np.asarray(chain_id), dtype=torch.int64).to(dev).contiguous(), data=data,
            height=heights.clone() if self.solve_height else heights, log_mean_prior=z(B), k=z(B, dt=i32), edges=z(B, K), sigma=z(B, K), rel=z(B, Gr), add=z(B, Ga),
            pred=z(B, N), J=z(B, N, K), prior=z(B), like=z(B), misfit=z(B), action=z(B, dt=i32), k_r=z(B, dt=i32),
            nl_a=z(3, B, dt=i32), nl_c=z(3, B, dt=i32), nl_b=z(B, dt=i32), edges_r=z(B, K), sigma_r=z(B, K), thk_r=z(B, K), rel_p=z(B, Gr), add_p=z(B, Ga),
            pred_r=z(B, N), J_r=z(B, N, K), chol=z(B, K, K), log_prop=z(B, K), sigma_p=z(B, K), pred_p=z(B, N), misfit_p=z(B),
            like_p=z(B), J_p=z(B, N, K), log_ratio=z(B), n_accepted=z(B, dt=i64), k_hist=z(B, K + 1, dt=i32),
            edge_hist=z(B, self.n_depth_bins, dt=i32), rel_hist=z(B, Gr, 99, dt=i32), add_hist=z(B, Ga, 99, dt=i32),
            hitmap=z(B, self.n_value_bins, self.n_depth_bins, dt=i32) if hitmap else None,
            hit_dwell=z(B, dt=i32) if hitmap else None,
            burned_in_iteration=torch.full((B,), -1, dtype=i32, device=dev), status=z(B, dt=i32),
            best_posterior=z(B), best_k=z(B, dt=i32), best_edges=z(B, K), best_sigma=z(B, K), best_rel=z(B, Gr), best_add=z(B, Ga),
            iteration0=z(B, dt=i32),
            height_p=heights.clone() if self.solve_height else None, height0=heights if self.solve_height else None,
            height_hist=z(B, 99, dt=i32) if self.solve_height else None, best_height=heights.clone() if self.solve_height else None,
            step_flags=z(B, dt=i32),
            trace_misfit=torch.full((B, self.trace_length), float("nan"), dtype=torch.float64, device=dev) if self.trace_every > 0 else None,
            trace_accept=z(B, self.trace_length, dt=torch.uint8) if self.trace_every > 0 else None,
            best_iteration=z(B, dt=i32))
        self._bind()
        self.iteration = 0
        self.forward_waves = int(forward_waves)      # also passed explicitly to the forward calls of the initialisation
        self.ignore_likelihood = bool(ignore_likelihood or o.get("ignore_likelihood", False))
        self._initialize()
        if self.ignore_likelihood:
            self._drop_observation()

    @property
    def observed(self):
        """The measured data [B, N] (``t['data']`` unless the chains sample the prior alone: then the sampler's block is all-inactive)."""
        return self.t["observed"] if self.t.get("observed") is not None else self.t["data"]

    def _drop_observation(self):
        """ignore_likelihood: from here on no channel is active (data <= 0 is the reference's own flag for an inactive channel,
        EmDataPoint.py:54-56), the likelihood of every state is the constant 0 and the schedule starts burned in."""
        t = self.t
        t["observed"] = t["data"].clone()
        t["data"].zero_()
        for name in ("like", "misfit", "init_like", "init_misfit"):
            t[name].zero_()
        t["best_posterior"].copy_(t["prior"])
        if self._o.schedule == 1:
            t["burned_in_iteration"].fill_(int(self._o.n_markov_chains))

    def _bind(self):
        """(Re)build the gbp_rj_chains struct from the tensors in self.t."""
        rc = _lib.RjChains()
        rc.B = int(self.t["k"].shape[0])
        for name in _lib.RJ_CHAIN_FIELDS:
            setattr(rc, name, None if self.t[name] is None else self.t[name].data_ptr())
        self._c = rc
        self._set_row_map(None)

    def _set_row_map(self, index):
        """Hook for samplers whose system handle holds per-row state (time-domain chains with per-sounding geometry): the
        launches that follow evaluate the block's rows ``index`` (int64 tensor), or -- None -- the block's rows in order."""

    def __getattr__(self, name):              # chain state by the names of gbp_rj_chains
        t = self.__dict__.get("t")
        if t is not None and name in t:
            if name == "hitmap" and t["hitmap"] is not None:
                # a model enters the hit map with its dwell time when it is replaced; settle the current models first
                with torch.cuda.device(self.device):
                    _lib.check(_lib.load().gbp_rj_flush_posteriors(self._o, self._c, self._stream()))
            return t[name]
        raise AttributeError(name)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def channel_std(self, data, rel, add):
        """[rows, N] standard deviation of every channel from error levels [rows, groups] (DataPoint.std / TdemDataPoint.std)."""
        t, N = self.t, data.shape[1]
        zero = torch.zeros(N, dtype=torch.long, device=self.device)
        rg = t["rel_group"].long() if t["rel_group"] is not None else zero
        ag = t["add_group"].long() if t["add_group"] is not None else zero
        sc = t["add_scale"][None, :] if t["add_scale"] is not None else 1.0
        return torch.sqrt((rel[:, rg] * data) ** 2 + (add[:, ag] * sc) ** 2).contiguous()

    # -- Inference1D.initialize (:353-464): best half-space, its forward / Jacobian, prior and likelihood ----------------
    def _initialize(self):
        o, B, K, t = self.o, self.B, self.K, self.t
        t["rel"].copy_(torch.as_tensor(self._rel0, device=self.device)[None, :].expand_as(t["rel"]))
        t["add"].copy_(torch.as_tensor(self._add0, device=self.device)[None, :].expand_as(t["add"]))
        grid = torch.logspace(-4.0, 4.0, 100, dtype=torch.float64, device=self.device)
        best = torch.zeros(B, dtype=torch.int64, device=self.device)
        chunk = max(1, min(B, (1 << 22) // 100))                      # B x 100 half-space forwards, in slabs
        for s0 in range(0, B, chunk):
            nb = min(chunk, B - s0)
            n = nb * 100
            k1 = torch.ones(n, dtype=torch.int32, device=self.device)
            sig = torch.ones((n, K), dtype=torch.float64, device=self.device)
            sig[:, 0] = grid.repeat(nb)
            rep = lambda x: x[s0:s0 + nb].repeat_interleave(100, dim=0).contiguous()
            chi2 = torch.empty(n, dtype=torch.float64, device=self.device)
            logl = torch.empty_like(chi2)
            hh, dd, rr, aa, thk = rep(t["height"]), rep(t["data"]), rep(t["rel"]), rep(t["add"]), torch.zeros_like(sig)
            self._set_row_map(torch.arange(s0, s0 + nb, device=self.device).repeat_interleave(100))
            self._eval_loglike(k1, sig, thk, hh, dd, rr, aa, None, chi2, logl)
            best[s0:s0 + nb] = torch.argmin(chi2.view(nb, 100), dim=1)
        self._set_row_map(None)
        t["k"].fill_(1)
        t["sigma"].fill_(1.0)
        t["sigma"][:, 0] = grid[best]
        t["edges"].fill_(float("inf"))
        t["log_mean_prior"].copy_(torch.log(t["sigma"][:, 0]))
        thk = torch.zeros_like(t["sigma"])
        self._eval_loglike(t["k"], t["sigma"], thk, t["height"], t["data"], t["rel"], t["add"], t["pred"], t["misfit"], t["like"])
        self._eval_jacobian(t["k"], t["sigma"], thk, t["height"], t["J"], 1)
        prior = model_log_prior(t["edges"], t["sigma"], t["k"].to(torch.int64), K, self.gradient_precision, o["solve_gradient"],
                                self._o.value_precision if self._o.solve_value else None, t["log_mean_prior"])
        if self._o.solve_relative_error:
            prior = prior + sum(log_uniform_prior(t["rel"][:, g], self._bounds["rel"][0][g], self._bounds["rel"][1][g])
                                for g in range(self.n_rel_groups))
        if self._o.solve_additive_error and not self._o.additive_independent:
            prior = prior + sum(log_uniform_prior(t["add"][:, g], self._bounds["add"][0][g], self._bounds["add"][1][g])
                                for g in range(self.n_add_groups))
        if self.solve_height:
            prior = prior - math.log(2.0 * self.height_half_width)
        prior = prior + self._o.extra_log_prior
        t["prior"].copy_(prior)
        t["best_posterior"].copy_(t["like"] + t["prior"])
        t["best_sigma"].copy_(t["sigma"])
        t["best_edges"].copy_(t["edges"])
        t["best_k"].copy_(t["k"])
        t["best_rel"].copy_(t["rel"])
        t["best_add"].copy_(t["add"])
        # host-side bookkeeping of infer()'s restarts (not part of gbp_rj_chains; re-packed with the rest): the state the
        # chains start from and the reference's counters _n_zero_acceptance / _n_resets / "limiters armed"
        zi = lambda dt: torch.zeros(B, dtype=dt, device=self.device)
        t.update(init_sigma=t["sigma"][:, 0].clone(), init_pred=t["pred"].clone(), init_J0=t["J"][:, :, 0].clone(),
                 init_prior=t["prior"].clone(), init_like=t["like"].clone(), init_misfit=t["misfit"].clone(),
                 acc_mark=zi(torch.int64), n_zero=zi(torch.int32), n_resets=zi(torch.int32), limited=zi(torch.int32))

    # the two evaluations the initialisation needs, through the same entries the sampler uses (overridden for time-domain data)
    def _eval_loglike(self, k, sigma, thk, height, data, rel, add, pred, chi2, logl):
        _lib.check(_lib.load().gbp_fdem_forward_loglike_ex(
            self._h.ptr, k.numel(), self.K, k.data_ptr(), sigma.data_ptr(), thk.data_ptr(), height.data_ptr(), data.data_ptr(),
            rel.data_ptr(), add.data_ptr(), None if pred is None else pred.data_ptr(), chi2.data_ptr(), logl.data_ptr(),
            self.forward_waves, self._stream()))

    def _eval_jacobian(self, k, sigma, thk, height, J, max_layers):
        _lib.check(_lib.load().gbp_fdem_sensitivity_ex(self._h.ptr, k.numel(), self.K, k.data_ptr(), sigma.data_ptr(), thk.data_ptr(),
                                                       height.data_ptr(), J.data_ptr(), int(max_layers), self._o.exact_jacobian,
                                                       self._stream()))

    def _launch(self, n, accumulate):
        # run_mode: 0 = the library's choice, 1 = lock-step driver, 2 = persistent per-chain kernel (gbp_rj_run_mode); same chains
        _lib.check(_lib.load().gbp_rj_run_mode(self._h.ptr, self._o, self._c, self.iteration, int(n), int(bool(accumulate)),
                                               int(getattr(self, "run_mode", 0)), self._stream()))

    # -- sampling ------------------------------------------------------------------------------------------------------
    def run(self, n, accumulate=True):
        """n iterations of every chain (asynchronous: returns once the launches are queued)."""
        if n > 0:
            with torch.cuda.device(self.device):
                self._launch(n, accumulate)
            self.iteration += int(n)
        return self

    def step(self, accumulate=True):
        return self.run(1, accumulate)

    def infer(self, check_every=1000, compact_below=0.5, min_rows=64):
        """Run under the reference's schedule until every chain is done or has failed; the host looks at the status flags
        every ``check_every`` iterations.  When fewer than ``compact_below`` of the rows are still running, the finished
        chains are set aside and the block is re-packed (finished chains would otherwise idle through every kernel until
        the slowest one stops); the chains do not notice -- their random streams are keyed by ``chain_id``, not by the
        row -- and all rows are back in place on return.

        Restarts (Inference1D.update :764-776, infer :665-678, reset :984-999): a chain that has not burned in and accepted
        nothing over ``reset_limit`` consecutive windows of ``update_plot_every`` iterations goes back to its initial state
        and its schedule starts over (the random streams go on: they are indexed by the lockstep iteration); the third
        restart arms the reference's (inert) variance limiters and restarts once more, the third after that gives the
        sounding up.  Windows are the reference's: the decisions of updates (m-1) W ... m W - 1, looked at after update m W.
        Returns the number of chains that failed."""
        assert self._o.schedule == 1, "infer() needs reference_schedule=True"
        window = int(self.o.get("update_plot_every") or 5000)
        reset_limit = int(self.o.get("reset_limit") or 1)
        restarts = window > 1
        limit = 8 * self._o.n_markov_chains + 2 if restarts else 2 * self._o.n_markov_chains + 2
        full, rows = None, None                   # the set-aside full-size tensors and the global row of each working row
        while self.iteration < limit:
            running = self.t["status"] == 0
            n_run = int(running.sum())
            if n_run == 0:
                break
            n_rows = int(running.numel())
            if n_rows > min_rows and n_run < compact_below * n_rows:
                keep = torch.nonzero(running).flatten()
                if full is None:
                    full = dict(self.t)
                    rows = torch.arange(self.B, device=self.device)
                    if self.t["chain_id"] is None:
                        full["chain_id"] = torch.arange(self._o.first_chain, self._o.first_chain + self.B, dtype=torch.int64, device=self.device)
                        self.t["chain_id"] = full["chain_id"]
                else:
                    self._scatter(full, rows)
                rows = rows[keep]
                self.t = {n: (v if v is None or n in ("add_scale", "rel_group", "add_group") else (v[:, keep] if n in ("nl_a", "nl_c") else v[keep]).contiguous())
                          for n, v in self.t.items()}
                self._bind()
                running = self.t["status"] == 0
            n = min(check_every, limit - self.iteration)
            if restarts:
                pos = self.iteration % window
                if pos == window - 1:             # the update that closes the window: decide on the counts before it
                    snap = self.t["n_accepted"].clone()
                    self.run(1)
                    self._restart_stuck_chains(running, snap, reset_limit)
                    continue
                n = min(n, window - 1 - pos)
            self.run(n)
        if full is not None:
            self._scatter(full, rows)
            self.t = full
            self._bind()
        return int((self.t["status"] == 2).sum())

    def _restart_stuck_chains(self, was_running, snap, reset_limit):
        """The reference's end-of-window test for the chains that were running before this iteration (see infer)."""
        t = self.t
        cand = was_running & (t["burned_in_iteration"] < 0)
        zero = cand & (snap == t["acc_mark"])
        t["acc_mark"].copy_(snap)
        t["n_zero"].copy_(torch.where(zero, t["n_zero"] + 1, torch.where(cand, torch.zeros_like(t["n_zero"]), t["n_zero"])))
        reset = zero & (t["n_zero"] >= reset_limit)
        if not bool(reset.any()):
            return
        n_res = t["n_resets"] + reset.to(torch.int32)
        third = reset & (n_res == 3)
        give_up = third & (t["limited"] != 0)
        rearm = third & (t["limited"] == 0)                       # limiters armed: _n_resets = 0, reset() again -> 1
        t["n_resets"].copy_(torch.where(rearm, torch.ones_like(n_res), n_res))
        t["limited"].copy_(torch.where(rearm, torch.ones_like(t["limited"]), t["limited"]))
        r = torch.nonzero(reset).flatten()
        rel0 = torch.as_tensor(self._rel0, device=self.device)
        add0 = torch.as_tensor(self._add0, device=self.device)
        t["k"][r] = 1
        t["sigma"][r] = 1.0
        t["sigma"][r, 0] = t["init_sigma"][r]
        t["edges"][r] = float("inf")
        t["rel"][r] = rel0
        t["add"][r] = add0
        t["pred"][r] = t["init_pred"][r]
        t["J"][r] = 0.0
        t["J"][r, :, 0] = t["init_J0"][r]
        for name in ("prior", "like", "misfit"):
            t[name][r] = t["init_" + name][r]
        for name in ("n_accepted", "acc_mark", "n_zero", "k_hist", "edge_hist", "rel_hist", "add_hist", "hitmap", "hit_dwell", "height_hist"):
            if t.get(name) is not None:
                t[name][r] = 0
        if self.solve_height:
            t["height"][r] = t["height0"][r]
            t["best_height"][r] = t["height0"][r]
        self._restart_more(r)
        t["burned_in_iteration"][r] = -1
        t["iteration0"][r] = self.iteration
        t["best_posterior"][r] = t["init_like"][r] + t["init_prior"][r]
        t["best_k"][r] = 1
        t["best_rel"][r] = rel0
        t["best_add"][r] = add0
        t["best_sigma"][r] = t["sigma"][r]
        t["best_edges"][r] = float("inf")
        t["best_iteration"][r] = 0
        if t.get("trace_misfit") is not None:                    # (Inference1D.reset :984-999 starts the two arrays over)
            t["trace_misfit"][r] = float("nan")
            t["trace_accept"][r] = 0
        t["status"].copy_(torch.where(give_up, torch.full_like(t["status"], 2), torch.where(reset, torch.zeros_like(t["status"]), t["status"])))

    def _restart_more(self, r):
        """Hook: further per-chain state of a subclass going back to its initial values for the rows ``r``."""

    def _scatter(self, full, rows):
        """Working rows -> their places in the full-size tensors."""
        for n, v in self.t.items():
            if v is None or full[n] is v:
                continue
            if n in ("nl_a", "nl_c"):
                full[n][:, rows] = v
            else:
                full[n][rows] = v

    def summaries(self):
        """[B, 6] per-sounding summary for the gather: misfit, logL, prior, k, acceptance rate, best posterior."""
        t = self.t
        return torch.stack([t["misfit"], t["like"], t["prior"], t["k"].to(torch.float64),
                            t["n_accepted"].to(torch.float64) / max(1, self.iteration), t["best_posterior"]], dim=1)
