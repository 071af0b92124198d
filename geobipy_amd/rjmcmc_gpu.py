"""Device-resident rjMCMC: all chains of a batch of soundings advance in lockstep as tensor programs.

Same statistical model as ``rjmcmc.py`` (the host restatement that reproduces the reference's chains decision by
decision): same priors, same four structural moves with the same validity rules and retry limits, same
stochastic-Newton log-normal value proposal, same error moves, same reversible-jump ratio and Metropolis test.
What differs is only where the random numbers come from -- one counter-based torch generator for the whole batch
instead of one numpy generator per sounding consumed in the reference's call order -- so a chain here is a valid
draw from the same Markov kernel but not the *same* draw as the reference's chain for a seed; the deterministic
pieces are checked against ``rjmcmc.py`` function by function (tests/test_rjmcmc_gpu.py).

Everything per-chain is a padded tensor: k[B], edges[B, K-1] (+inf padded), sigma[B, K] (1 padded), J[B, N, K];
the k x k algebra (Hessian, inverse, Cholesky sampling) is batched [B, K, K] with an identity block on the padded
dimensions.  The hot path is entered three times per iteration through the C ABI: forward + Jacobian of the
remapped models whose structure changed, the fused forward + chi^2 + logL of every proposal, and the Jacobian of the
proposals that inserted / deleted a layer.
"""
import math

import numpy as np
import torch

from . import _lib

NONE, INSERT, DELETE, PERTURB = 0, 1, 2, 3
LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------------------------
# pure tensor pieces (run on any device; tested on the CPU against rjmcmc.py)
# ----------------------------------------------------------------------------------------------------------------
def layer_widths(edges, k):
    """Finite layer thicknesses [B, K]: w_j = e_j - e_{j-1} for j < k - 1, 0 elsewhere."""
    B, Km1 = edges.shape
    K = Km1 + 1
    top = torch.cat([torch.zeros(B, 1, dtype=edges.dtype, device=edges.device), edges], dim=1)[:, :K - 1]
    w = edges - top
    j = torch.arange(K - 1, device=edges.device)[None, :]
    w = torch.where(j < (k[:, None] - 1), w, torch.zeros_like(w))
    return torch.cat([w, torch.zeros(B, 1, dtype=edges.dtype, device=edges.device)], dim=1)


def remap(action, index, value, edges, sigma, k):
    """Apply the structural moves (rjmcmc.perturb_structure's remapping) to padded tensors.
    index: the reference's full-edge index i (1..k-1 for delete / perturb, 1..k for insert); value: new depth
    (insert) or dz (perturb)."""
    B, K = sigma.shape
    dev = sigma.device
    j = torch.arange(K, device=dev)[None, :]
    je = torch.arange(K - 1, device=dev)[None, :]
    i = index[:, None]
    ins, dele, per = (action == INSERT)[:, None], (action == DELETE)[:, None], (action == PERTURB)[:, None]
    # conductivities
    s_prev = torch.cat([sigma[:, :1], sigma[:, :-1]], dim=1)                    # sigma[j - 1]
    s_next = torch.cat([sigma[:, 1:], sigma[:, -1:]], dim=1)                    # sigma[j + 1]
    s_ins = torch.where(j < i, sigma, s_prev)                                   # new layer copies the one above
    merged = 0.5 * (sigma + s_next)
    s_del = torch.where(j < i - 1, sigma, torch.where(j == i - 1, merged, s_next))
    new_sigma = torch.where(ins, s_ins, torch.where(dele, s_del, sigma))
    # interior edges: full index i <-> interior index i - 1
    e_prev = torch.cat([edges[:, :1], edges[:, :-1]], dim=1)
    e_next = torch.cat([edges[:, 1:], torch.full_like(edges[:, :1], float("inf"))], dim=1)
    e_ins = torch.where(je < i - 1, edges, torch.where(je == i - 1, value[:, None].expand_as(edges), e_prev))
    e_del = torch.where(je < i - 1, edges, e_next)
    e_per = torch.where(je == i - 1, edges + value[:, None], edges)
    new_edges = torch.where(ins, e_ins, torch.where(dele, e_del, torch.where(per, e_per, edges)))
    new_k = k + (action == INSERT).to(k.dtype) - (action == DELETE).to(k.dtype)
    new_sigma = torch.where(j < new_k[:, None], new_sigma, torch.ones_like(new_sigma))
    new_edges = torch.where(je < new_k[:, None] - 1, new_edges, torch.full_like(new_edges, float("inf")))
    return new_edges, new_sigma, new_k


def min_gap(edges, k):
    """Smallest finite layer thickness (inf for a half-space)."""
    w = layer_widths(edges, k)
    j = torch.arange(w.shape[1], device=w.device)[None, :]
    w = torch.where(j < k[:, None] - 1, w, torch.full_like(w, float("inf")))
    return w.min(dim=1).values


def prior_operator(edges, k, value_precision, gradient_precision, solve_gradient=True):
    """Wm'Wm [B, K, K] (rjmcmc.model_prior_derivative, order 2), identity on the padded block."""
    B, Km1 = edges.shape
    K = Km1 + 1
    dev, dt = edges.device, edges.dtype
    j = torch.arange(K, device=dev)[None, :]
    live = j < k[:, None]
    op = torch.diag_embed(torch.where(live, torch.full((B, K), value_precision, dtype=dt, device=dev),
                                      torch.ones((B, K), dtype=dt, device=dev)))
    if not solve_gradient:
        return op
    x = layer_widths(edges, k)
    kk = k[:, None]
    deepest = torch.gather(torch.cat([torch.zeros(B, 1, dtype=dt, device=dev), edges], dim=1), 1,
                           (k[:, None] - 1).clamp(min=0))                           # e_{k-1} (0 for k = 1)
    x_last = torch.where(kk == 2, x[:, :1], torch.gather(x, 1, (kk - 2).clamp(min=0)) + deepest)
    x = torch.where(j == kk - 1, x_last.expand_as(x), x)
    c2c = 0.5 * (x[:, :-1] + x[:, 1:])
    je = torch.arange(K - 1, device=dev)[None, :]
    t2 = torch.where(je < kk - 1, 1.0 / (c2c * (kk - 1).clamp(min=1).to(dt)) ** 2, torch.zeros_like(c2c)) * gradient_precision
    G = torch.zeros((B, K, K), dtype=dt, device=dev)
    idx = torch.arange(K - 1, device=dev)
    G[:, idx, idx] += t2
    G[:, idx + 1, idx + 1] += t2
    G[:, idx, idx + 1] -= t2
    G[:, idx + 1, idx] -= t2
    single = (k == 1)
    G[single, 0, 0] += gradient_precision                                             # Wz = [[1]] for a half-space
    return op + G


def stochastic_newton(edges, sigma, k, J, pred, data, std, log_mean_prior, value_precision, gradient_precision, alpha,
                      solve_gradient=True):
    """Batched rjmcmc.stochastic_newton: returns (mean of ln sigma' [B, K], hess = H^-1 [B, K, K], H [B, K, K])."""
    B, K = sigma.shape
    j = torch.arange(K, device=sigma.device)[None, :]
    live = (j < k[:, None]).to(sigma.dtype)
    active = (data > 0.0).to(sigma.dtype)
    P = active / (std * std)
    Jm = J * live[:, None, :]
    op = prior_operator(edges, k, value_precision, gradient_precision, solve_gradient)
    hess = op + torch.einsum("bnk,bn,bnl->bkl", Jm, P, Jm)
    dev_log = (torch.log(sigma) - log_mean_prior[:, None]) * live
    resid = torch.where(data > 0.0, pred - data, torch.zeros_like(pred))
    grad = torch.einsum("bkl,bl->bk", op, dev_log) * live + torch.einsum("bnk,bn->bk", Jm, P * resid)
    H = torch.linalg.inv(hess)
    mean_log = torch.log(sigma) - alpha * torch.einsum("bkl,bl->bk", H, grad)
    return mean_log, hess, H


def model_log_prior(edges, sigma, k, max_cells, gradient_precision, solve_gradient=True):
    """Batched rjmcmc.model_log_prior."""
    lp = torch.full((sigma.shape[0],), -math.log(max_cells - 1.0), dtype=sigma.dtype, device=sigma.device)
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


# ----------------------------------------------------------------------------------------------------------------
# the sampler
# ----------------------------------------------------------------------------------------------------------------
class DeviceChains:
    """B rjMCMC chains (one per sounding) resident on one GPU.  ``options``: keys of the reference's options file."""

    def __init__(self, system, heights, data, seed=0, exact_jacobian=False, device=None, **options):
        from .inference import OPTION_DEFAULTS
        o = dict(OPTION_DEFAULTS)
        o.update({k: v for k, v in options.items() if v is not None})
        self.o = o
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.system, self.exact = system, bool(exact_jacobian)
        with torch.cuda.device(self.device):
            self._h = system.handle()
        f64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).to(self.device).contiguous()
        self.data, self.height = f64(data), f64(heights)
        self.B, self.N = self.data.shape
        self.K = int(o["maximum_number_of_layers"])
        self.min_width = float(o["minimum_thickness"])
        self.min_edge = max(float(o["minimum_depth"]), self.min_width)
        self.max_edge = float(o["maximum_depth"])
        p = torch.tensor([o["probability_of_birth"], o["probability_of_death"], o["probability_of_perturb"],
                          o["probability_of_no_change"]], dtype=torch.float64, device=self.device)
        self.p_event = p / p.sum()
        self.value_precision = 1.0 / math.log(1.0 + o["factor"]) ** 2
        self.gradient_precision = 1.0 / o["gradient_standard_deviation"] ** 2
        self.alpha = float(o["covariance_scaling"])
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self._initialize()

    # -- kernels ---------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _forward_loglike(self, k, sigma, thk, height, data, rel, add):
        n = k.numel()
        pred = torch.empty((n, self.N), dtype=torch.float64, device=self.device)
        chi2 = torch.empty(n, dtype=torch.float64, device=self.device)
        logl = torch.empty(n, dtype=torch.float64, device=self.device)
        if n:
            _lib.check(_lib.load().gbp_fdem_forward_loglike(
                self._h.ptr, n, self.K, k.data_ptr(), sigma.data_ptr(), thk.data_ptr(), height.data_ptr(), data.data_ptr(),
                rel.data_ptr(), add.data_ptr(), pred.data_ptr(), chi2.data_ptr(), logl.data_ptr(), self._stream()))
        return pred, chi2, logl

    def _sensitivity(self, k, sigma, thk, height):
        """J[n, N, K] at n models; large batches are launched per layer-count bucket so that shallow models do not pay
        the LDS footprint (and lost occupancy) of the deepest one."""
        n = k.numel()
        J = torch.zeros((n, self.N, self.K), dtype=torch.float64, device=self.device)
        if n == 0:
            return J
        kmax = int(k.max().item())

        def launch(kk, ss, tt, hh, out, cap):
            _lib.check(_lib.load().gbp_fdem_sensitivity_ex(
                self._h.ptr, kk.numel(), self.K, kk.data_ptr(), ss.data_ptr(), tt.data_ptr(), hh.data_ptr(), out.data_ptr(),
                int(cap), 1 if self.exact else 0, self._stream()))

        if n < 512 or kmax <= 8:
            launch(k, sigma, thk, height, J, kmax)
            return J
        lo = 0
        for hi in (8, 16, self.K):
            idx = torch.nonzero((k > lo) & (k <= hi)).flatten()
            if idx.numel():
                out = torch.empty((idx.numel(), self.N, self.K), dtype=torch.float64, device=self.device)
                launch(k[idx].contiguous(), sigma[idx].contiguous(), thk[idx].contiguous(), height[idx].contiguous(), out,
                       min(hi, kmax))
                J[idx] = out
            lo = hi
        return J

    def _thk(self, edges, k):
        return layer_widths(edges, k).contiguous()

    def _std(self, rel, add):
        return torch.sqrt((rel[:, None] * self.data) ** 2 + add[:, None] ** 2)

    # -- initialisation (Inference1D.initialize) -------------------------------------------------------------------
    def _initialize(self):
        o, B, K, dev = self.o, self.B, self.K, self.device
        self.rel = torch.full((B,), float(o["initial_relative_error"]), dtype=torch.float64, device=dev)
        self.add = torch.full((B,), float(o["initial_additive_error"]), dtype=torch.float64, device=dev)
        grid = torch.logspace(-4.0, 4.0, 100, dtype=torch.float64, device=dev)
        n = B * 100
        k1 = torch.ones(n, dtype=torch.int32, device=dev)
        sig = torch.ones((n, K), dtype=torch.float64, device=dev)
        sig[:, 0] = grid.repeat(B)
        rep = lambda t: t.repeat_interleave(100, dim=0).contiguous()
        _, chi2, _ = self._forward_loglike(k1, sig, torch.zeros_like(sig), rep(self.height), rep(self.data), rep(self.rel),
                                           rep(self.add))
        best = torch.argmin(chi2.view(B, 100), dim=1)
        self.k = torch.ones(B, dtype=torch.int32, device=dev)
        self.sigma = torch.ones((B, K), dtype=torch.float64, device=dev)
        self.sigma[:, 0] = grid[best]
        self.edges = torch.full((B, K - 1), float("inf"), dtype=torch.float64, device=dev)
        self.log_mean_prior = torch.log(self.sigma[:, 0]).clone()
        thk = self._thk(self.edges, self.k)
        self.pred, self.misfit, self.like = self._forward_loglike(self.k, self.sigma, thk, self.height, self.data, self.rel,
                                                                  self.add)
        self.J = self._sensitivity(self.k, self.sigma, thk, self.height)
        self.prior = (model_log_prior(self.edges, self.sigma, self.k, K, self.gradient_precision, o["solve_gradient"])
                      + log_uniform_prior(self.rel, o["minimum_relative_error"], o["maximum_relative_error"])
                      + log_uniform_prior(self.add, o["minimum_additive_error"], o["maximum_additive_error"]))
        self.iteration = 0
        self.n_accepted = torch.zeros(B, dtype=torch.int64, device=dev)
        self.k_hist = torch.zeros((B, K + 1), dtype=torch.int32, device=dev)           # posterior on the layer count
        self.best_posterior = self.like + self.prior
        self.best_sigma, self.best_edges, self.best_k = self.sigma.clone(), self.edges.clone(), self.k.clone()

    # -- random draws ------------------------------------------------------------------------------------------------
    def _u(self, *shape):
        return torch.rand(shape, dtype=torch.float64, device=self.device, generator=self.gen)

    def _n(self, *shape):
        return torch.randn(shape, dtype=torch.float64, device=self.device, generator=self.gen)

    # -- structural move (RectilinearMesh1D.perturb) ------------------------------------------------------------------
    def _propose_structure(self):
        B, K, dev = self.B, self.K, self.device
        k = self.k.to(torch.int64)
        action = torch.zeros(B, dtype=torch.int64, device=dev)
        index = torch.ones(B, dtype=torch.int64, device=dev)
        value = torch.zeros(B, dtype=torch.float64, device=dev)
        pending = torch.ones(B, dtype=torch.bool, device=dev)
        lo, hi = math.log(self.min_edge), math.log(self.max_edge)
        for _ in range(8):                                   # re-draw the event when 9 birth / perturb tries fail
            if not bool(pending.any()):
                break
            p = self.p_event[None, :].repeat(B, 1)
            p[:, 0] = torch.where(k == K, torch.zeros_like(p[:, 0]), p[:, 0])
            p[:, 1] = torch.where(k == 1, torch.zeros_like(p[:, 1]), p[:, 1])
            p[:, 2] = torch.where(k == 1, torch.zeros_like(p[:, 2]), p[:, 2])
            event = torch.searchsorted(torch.cumsum(p / p.sum(dim=1, keepdim=True), dim=1), self._u(B, 1).clamp(max=1 - 1e-16)).flatten().clamp(max=3)
            is_none, is_birth, is_death, is_pert = event == 3, event == 0, event == 1, event == 2
            done = pending & (is_none | is_death)
            action = torch.where(pending & is_death, torch.full_like(action, DELETE), action)
            d_idx = (self._u(B) * (k - 1).clamp(min=1).to(torch.float64)).floor().to(torch.int64).clamp(max=(k - 2).clamp(min=0)) + 1
            index = torch.where(pending & is_death, d_idx, index)
            action = torch.where(pending & is_none, torch.full_like(action, NONE), action)
            pending = pending & ~done
            need_b, need_p = pending & is_birth, pending & is_pert
            for _try in range(9):                            # the reference's 10th try always fails
                if not bool((need_b | need_p).any()):
                    break
                depth = torch.exp(self._u(B) * (hi - lo) + lo)
                pos = torch.searchsorted(self.edges, depth[:, None]).flatten() + 1           # full-edge insert index
                e_try, _, k_try = remap(torch.where(need_b, INSERT, NONE), pos, depth, self.edges, self.sigma, k)
                ok_b = need_b & (min_gap(e_try, k_try) > self.min_width)
                p_idx = (1.0 + self._u(B) * (k - 1).clamp(min=1).to(torch.float64)).floor().to(torch.int64).clamp(max=(k - 1).clamp(min=1))
                dz = torch.sign(self._n(B)) * self.min_width * self._u(B)
                e_try, _, _ = remap(torch.where(need_p, PERTURB, NONE), p_idx, dz, self.edges, self.sigma, k)
                first = e_try[:, 0]
                last = torch.gather(e_try, 1, (k[:, None] - 2).clamp(min=0)).flatten()
                ok_p = need_p & (min_gap(e_try, k) > self.min_width) & (first > self.min_edge) & (last < self.max_edge)
                action = torch.where(ok_b, torch.full_like(action, INSERT), torch.where(ok_p, torch.full_like(action, PERTURB), action))
                index = torch.where(ok_b, pos, torch.where(ok_p, p_idx, index))
                value = torch.where(ok_b, depth, torch.where(ok_p, dz, value))
                pending = pending & ~(ok_b | ok_p)
                need_b, need_p = need_b & ~ok_b, need_p & ~ok_p
        return action, index, value            # chains still pending after 8 rounds (never observed) keep NONE

    # -- one iteration (Inference1D.accept_reject) ----------------------------------------------------------------------
    @torch.no_grad()
    def step(self):
        o, B, K, dev = self.o, self.B, self.K, self.device
        action, index, value = self._propose_structure()
        k64 = self.k.to(torch.int64)
        edges_r, sigma_r, k_r = remap(action, index, value, self.edges, self.sigma, k64)
        k_r32 = k_r.to(torch.int32)
        changed = action != NONE
        pred_r, J = self.pred, self.J
        idx = torch.nonzero(changed).flatten()
        if idx.numel():                                     # fm_dlogc at the remapped models whose structure changed
            kk, ss = k_r32[idx].contiguous(), sigma_r[idx].contiguous()
            tt, hh = self._thk(edges_r[idx], k_r[idx]), self.height[idx].contiguous()
            p_sub, _, _ = self._forward_loglike(kk, ss, tt, hh, self.data[idx].contiguous(), self.rel[idx].contiguous(),
                                                self.add[idx].contiguous())
            pred_r, J = self.pred.clone(), self.J.clone()
            pred_r[idx] = p_sub
            J[idx] = self._sensitivity(kk, ss, tt, hh)
        mean_log, hess, H = stochastic_newton(edges_r, sigma_r, k_r, J, pred_r, self.data, self._std(self.rel, self.add),
                                              self.log_mean_prior, self.value_precision, self.gradient_precision, self.alpha,
                                              o["solve_gradient"])
        L = torch.linalg.cholesky(0.5 * (H + H.transpose(1, 2)))
        j = torch.arange(K, device=dev)[None, :]
        live = j < k_r[:, None]
        log_prop = torch.where(live, mean_log + torch.einsum("bkl,bl->bk", L, self._n(B, K) * live), torch.zeros_like(mean_log))
        sigma_p = torch.exp(log_prop)
        rel_p = self._propose_error(self.rel, o["relative_error_proposal_variance"], o["minimum_relative_error"],
                                    o["maximum_relative_error"])
        add_p = self._propose_error(self.add, o["additive_error_proposal_variance"], o["minimum_additive_error"],
                                    o["maximum_additive_error"])
        thk_p = self._thk(edges_r, k_r)
        pred_p, misfit_p, like_p = self._forward_loglike(k_r32.contiguous(), sigma_p.contiguous(), thk_p, self.height, self.data,
                                                         rel_p, add_p)
        prior_p = (log_uniform_prior(rel_p, o["minimum_relative_error"], o["maximum_relative_error"])
                   + log_uniform_prior(add_p, o["minimum_additive_error"], o["maximum_additive_error"])
                   + model_log_prior(edges_r, sigma_p, k_r, K, self.gradient_precision, o["solve_gradient"]))
        dq = torch.zeros(B, dtype=torch.float64, device=dev)
        jump = (action == INSERT) | (action == DELETE)
        jdx = torch.nonzero(jump).flatten()
        J_new = J
        if jdx.numel():                                     # reversible-jump proposal ratio (Model.proposal_probabilities)
            Jp = self._sensitivity(k_r32[jdx].contiguous(), sigma_p[jdx].contiguous(), thk_p[jdx].contiguous(),
                                   self.height[jdx].contiguous())
            J_new = J.clone()
            J_new[jdx] = Jp
            lv = live[jdx].to(torch.float64)
            std_t = self._std(rel_p, add_p)[jdx]
            dat = self.data[jdx]
            P = (dat > 0.0).to(torch.float64) / (std_t * std_t)
            resid = torch.where(dat > 0.0, pred_p[jdx] - dat, torch.zeros_like(dat))
            op = prior_operator(edges_r[jdx], k_r[jdx], self.value_precision, self.gradient_precision, o["solve_gradient"])
            grad = (torch.einsum("bkl,bl->bk", op, (log_prop[jdx] - self.log_mean_prior[jdx, None]) * lv) * lv
                    + torch.einsum("bnk,bn->bk", Jp * lv[:, None, :], P * resid))
            mean_r = log_prop[jdx] + self.alpha * torch.einsum("bkl,bl->bk", H[jdx], grad)
            log_rem = torch.log(sigma_r[jdx]) * lv
            d1 = (log_rem - mean_r) * lv
            d2 = (log_prop[jdx] - log_rem) * lv
            hq = hess[jdx]
            dq[jdx] = -0.5 * torch.einsum("bk,bkl,bl->b", d1, hq, d1) + 0.5 * torch.einsum("bk,bkl,bl->b", d2, hq, d2)
        log_ratio = (prior_p - self.prior) + (like_p - self.like) + dq
        accept = torch.log(self._u(B)) < log_ratio            # NaN compares False, -inf priors reject
        a1, a2, a3 = accept[:, None], accept[:, None, None], accept
        self.edges = torch.where(a1, edges_r, self.edges)
        self.sigma = torch.where(a1, sigma_p, self.sigma)
        self.k = torch.where(a3, k_r32, self.k)
        self.rel, self.add = torch.where(a3, rel_p, self.rel), torch.where(a3, add_p, self.add)
        self.pred = torch.where(a1, pred_p, self.pred)
        self.J = torch.where(a2, J_new, self.J)
        self.prior, self.like = torch.where(a3, prior_p, self.prior), torch.where(a3, like_p, self.like)
        self.misfit = torch.where(a3, misfit_p, self.misfit)
        self.n_accepted += accept.to(torch.int64)
        self.iteration += 1
        self.k_hist.scatter_add_(1, self.k.to(torch.int64)[:, None], torch.ones((B, 1), dtype=torch.int32, device=dev))
        post = self.like + self.prior
        better = post > self.best_posterior
        self.best_posterior = torch.where(better, post, self.best_posterior)
        self.best_sigma = torch.where(better[:, None], self.sigma, self.best_sigma)
        self.best_edges = torch.where(better[:, None], self.edges, self.best_edges)
        self.best_k = torch.where(better, self.k, self.best_k)
        return accept

    def _propose_error(self, cur, var, lo, hi):
        sd, llo, lhi = math.sqrt(var), math.log(lo), math.log(hi)
        lc = torch.log(cur)
        x = lc + self._n(self.B) * sd
        bad = (x < llo) | (x > lhi)
        for _ in range(10):
            if not bool(bad.any()):
                break
            x = torch.where(bad, lc + self._n(self.B) * sd, x)
            bad = (x < llo) | (x > lhi)
        return torch.exp(torch.where(bad, lc, x))

    def run(self, n):
        for _ in range(n):
            self.step()
        return self

    def summaries(self):
        """[B, 6] per-sounding summary for the gather: misfit, logL, prior, k, acceptance rate, best posterior."""
        return torch.stack([self.misfit, self.like, self.prior, self.k.to(torch.float64),
                            self.n_accepted.to(torch.float64) / max(1, self.iteration), self.best_posterior], dim=1)
