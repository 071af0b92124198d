"""Host emulation of the three stage kernels of the device-resident rjMCMC step (geobipy_amd/csrc/gbp_rjmcmc.h), drawing
from the same counter-based streams, built on the host restatement rjmcmc.py (which is pinned to the reference's
chains).  Test infrastructure only."""
import math

import numpy as np

from geobipy_amd import rjmcmc

M32 = 0xFFFFFFFF


def philox(seed, c0, c1, c2, c3):
    """Philox4x32-10 (Salmon et al. 2011)."""
    k0, k1 = seed & M32, (seed >> 32) & M32
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def u53(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


class Rng:
    def __init__(self, seed, chain, it, stream):
        self.a, self.n, self.buf = (seed, chain, it, stream), 0, None

    def uniform(self):
        if self.buf is not None:
            u, self.buf = self.buf, None
            return u
        r = philox(*self.a, self.n)
        self.n += 1
        self.buf = u53(r[2], r[3])
        return u53(r[0], r[1])

    def normal(self):
        u1, u2 = self.uniform(), self.uniform()
        return math.sqrt(-2.0 * math.log(1.0 - u1)) * math.cos(2.0 * math.pi * u2)


def normal_pair(seed, chain, it, stream, j):
    r = philox(seed, chain, it, stream, j)
    rad, ang = math.sqrt(-2.0 * math.log(1.0 - u53(r[0], r[1]))), 2.0 * math.pi * u53(r[2], r[3])
    return rad * math.cos(ang), rad * math.sin(ang)


def propose_error(r, cur, sd, lo, hi):
    lc, llo, lhi = math.log(cur), math.log(lo), math.log(hi)
    x = lc + sd * r.normal()
    tries = 0
    while not (llo <= x <= lhi):
        x = lc + sd * r.normal()
        tries += 1
        if tries == 10:
            return cur
    return math.exp(x)


def propose_levels(r, cur, sd, lo, hi):
    """Joint proposal of the levels of one kind (scalars or vectors): all are redrawn while any is outside its prior."""
    scalar = np.ndim(cur) == 0
    cur, sd, lo, hi = (np.atleast_1d(np.asarray(v, dtype=float)) for v in (cur, sd, lo, hi))
    draw = lambda: np.array([math.log(c) + s_ * r.normal() for c, s_ in zip(cur, sd)])
    ok = lambda x: bool(np.all((x >= np.log(lo)) & (x <= np.log(hi))))
    x = draw()
    tries = 0
    while not ok(x):
        x = draw()
        tries += 1
        if tries == 10:
            return cur.item() if scalar else cur.copy()
    out = np.array([math.exp(v) for v in x])
    return out.item() if scalar else out


def channel_std(data, rel, add, add_scale=1.0, groups=None):
    """sqrt((rel_g d)^2 + (add_g' add_scale)^2) per channel; groups = (rel_group, add_group) channel -> level maps."""
    if groups is not None:
        rel, add = np.asarray(rel)[groups[0]], np.asarray(add)[groups[1]]
    return np.sqrt((rel * data) ** 2 + (add * add_scale) ** 2)


def levels_log_prior(x, lo, hi):
    x, lo, hi = (np.atleast_1d(np.asarray(v, dtype=float)) for v in (x, lo, hi))
    return float(sum(rjmcmc.ErrorPrior(l, h, 1.0).log_prior(v) for v, l, h in zip(x, lo, hi)))


def propose(o, seed, b, it, edges, sigma, rel, add, height=None, height0=None):
    """k_rj_propose for one chain.  edges: k - 1 interior depths.  Returns (action, idx, val, edges_r, sigma_r, rel_p, add_p) -- and,
    with a sampled height (o["height_scale"], o["height_half_width"]; drawn before the error levels), the proposed height."""
    r = Rng(seed, b, it, 0)
    k, K, mw = sigma.size, o["K"], o["min_width"]
    lo, hi = math.log(o["min_edge"]), math.log(o["max_edge"])
    action, idx, val = rjmcmc.NONE, 0, 0.0
    done = False
    for _ in range(8):
        if done:
            break
        pb = 0.0 if k == K else o["p"][0]
        pd, pp = (0.0, 0.0) if k == 1 else (o["p"][1], o["p"][2])
        u = r.uniform() * (pb + pd + pp + o["p"][3])
        if u < pb:
            for _t in range(9):
                depth = math.exp(lo + r.uniform() * (hi - lo))
                pos = int(np.searchsorted(edges, depth))
                prev = edges[pos - 1] if pos > 0 else 0.0
                nxt = edges[pos] if pos < k - 1 else math.inf
                if depth - prev > mw and nxt - depth > mw:
                    action, idx, val, done = rjmcmc.INSERT, pos + 1, depth, True
                    break
        elif u < pb + pd:
            idx = min(int(math.floor(r.uniform() * (k - 1))), k - 2) + 1
            action, done = rjmcmc.DELETE, True
        elif u < pb + pd + pp:
            for _t in range(9):
                i = min(int(math.floor(1.0 + r.uniform() * (k - 1))), k - 1)
                n = r.normal()
                dz = (1.0 if n > 0 else (-1.0 if n < 0 else 0.0)) * mw * r.uniform()
                z = edges.copy()
                z[i - 1] += dz
                full = np.r_[0.0, z, np.inf]
                if np.min(np.diff(full)) > mw and z[0] > o["min_edge"] and z[-1] < o["max_edge"]:
                    action, idx, val, done = rjmcmc.PERTURB, i, dz, True
                    break
        else:
            done = True
    full = np.r_[0.0, edges, np.inf]
    if action == rjmcmc.INSERT:
        e_r, s_r = np.insert(full, idx, val)[1:-1], np.insert(sigma, idx, sigma[idx - 1])
    elif action == rjmcmc.DELETE:
        s_r = np.delete(sigma, idx)
        s_r[idx - 1] = 0.5 * (sigma[idx - 1] + sigma[idx])
        e_r = np.delete(full, idx)[1:-1]
    elif action == rjmcmc.PERTURB:
        e_r, s_r = edges.copy(), sigma.copy()
        e_r[idx - 1] += val
    else:
        e_r, s_r = edges.copy(), sigma.copy()
    if height is not None:
        lo_h, hi_h = height0 - o["height_half_width"], height0 + o["height_half_width"]
        height_p = height + o["height_scale"] * r.normal()
        tries = 0
        while not (lo_h <= height_p <= hi_h):
            height_p = height + o["height_scale"] * r.normal()
            tries += 1
            if tries == 10:
                height_p = height
                break
    rel_p = propose_levels(r, rel, o["rel_sd"], o["rel_min"], o["rel_max"])
    if o.get("add_independent"):          # Tempest's multipliers: one draw per level about a fixed centre, no redraw (gbp_rj_options)
        add_p = np.array([math.exp(math.log(c_) + sd_ * r.normal()) for c_, sd_ in zip(np.atleast_1d(o["add_centre"]), np.atleast_1d(o["add_sd"]))])
    else:
        add_p = propose_levels(r, add, o["add_sd"], o["add_min"], o["add_max"])
    if height is not None:
        return action, idx, val, e_r, s_r, rel_p, add_p, height_p
    return action, idx, val, e_r, s_r, rel_p, add_p


def propose_angles(moves, seed, b, it, current, centre):
    """k_td_moves_propose for one chain: moves = [(name, half_width, scale)], drawn in that order from stream 3; random walk redrawn
    up to 10 times while outside the uniform prior centre +- half_width, then the current value is kept."""
    r = Rng(seed, b, it, 3)
    out = {}
    for name, hw, sc in moves:
        cur, lo, hi = current[name], centre[name] - hw, centre[name] + hw
        x = cur + sc * r.normal()
        tries = 0
        while not (lo <= x <= hi):
            x = cur + sc * r.normal()
            tries += 1
            if tries == 10:
                x = cur
                break
        out[name] = x
    return out


def newton(o, seed, b, it, vp, edges_r, sigma_r, J, pred, data, rel, add, add_scale=1.0, groups=None):
    """k_rj_newton for one chain: (log_prop, C) with precision = C C'."""
    k = sigma_r.size
    std = channel_std(data, rel, add, add_scale, groups)
    a = data > 0.0
    Ja, P = J[a][:, :k], 1.0 / std[a] ** 2
    op = rjmcmc.model_prior_derivative(vp, edges_r, sigma_r, 2)
    hess = op + Ja.T @ (P[:, None] * Ja)
    grad = op @ (np.log(sigma_r) - vp.log_mean) + Ja.T @ (P * (pred[a] - data[a]))
    C = np.linalg.cholesky(hess)
    z = np.array([v for j in range((k + 1) // 2) for v in normal_pair(seed, b, it, 1, j)])[:k]      # counter j -> normals 2j, 2j + 1
    log_prop = np.log(sigma_r) - o["alpha"] * np.linalg.solve(hess, grad) + np.linalg.solve(C.T, z)
    return log_prop, C


def accept(o, seed, b, it, sp, vp, action, edges_r, sigma_r, log_prop, C, J_p, pred_p, data, rel_p, add_p, like_p, prior, like,
           add_scale=1.0, groups=None, prior_const=0.0):
    """k_rj_accept for one chain: (log_ratio, accepted, prior_p).  prior_const: the density of the uniform height prior, when the
    height is sampled (its proposals are inside the prior by construction)."""
    prop = np.exp(log_prop)
    prior_p = rjmcmc.model_log_prior(sp, vp, edges_r, prop) + levels_log_prior(rel_p, o["rel_min"], o["rel_max"])
    if not o.get("add_independent"):
        prior_p = prior_p + levels_log_prior(add_p, o["add_min"], o["add_max"])
    prior_p = prior_p + prior_const
    dq = 0.0
    if action in (rjmcmc.INSERT, rjmcmc.DELETE):
        k = prop.size
        std = channel_std(data, rel_p, add_p, add_scale, groups)
        a = data > 0.0
        grad = rjmcmc.model_prior_derivative(vp, edges_r, prop, 1) + J_p[a][:, :k].T @ ((pred_p[a] - data[a]) / std[a] ** 2)
        hess = C @ C.T
        mean_r = log_prop + o["alpha"] * np.linalg.solve(hess, grad)
        d1, d2 = np.log(sigma_r) - mean_r, log_prop - np.log(sigma_r)
        dq = -0.5 * d1 @ hess @ d1 + 0.5 * d2 @ hess @ d2
        if not np.all(np.abs(mean_r) < 11356.0):        # the reference's long-double exp of the reverse mean over/underflows
            dq = np.nan
    log_ratio = (prior_p - prior) + (like_p - like) + dq
    r = philox(seed, b, it, 2, 0)
    u = u53(r[0], r[1])
    with np.errstate(divide="ignore"):
        return log_ratio, bool(np.log(u) < log_ratio), prior_p


class Chain:
    """A complete CPU chain with the device sampler's random streams: the three stage emulations above around an engine
    with forward(edges, values) / sensitivity(edges, values) (the C oracle in the tests).  Carries the same state and
    posterior accumulators as one row of gbp_rj_chains."""

    def __init__(self, o, seed, b, engine, sp, vp, data, sigma0, rel, add, n_depth_bins, depth_bin_width, add_scale=1.0, groups=None,
                 height=None, angle_moves=None, angles=None):
        self.o, self.seed, self.b, self.engine, self.sp, self.vp, self.data = o, seed, b, engine, sp, vp, data
        self.add_scale, self.groups = add_scale, groups
        self.edges, self.sigma, self.rel, self.add = np.zeros(0), np.array([sigma0]), rel, add
        self.height, self.height0 = height, height          # a sampled height (engine.forward / sensitivity then take z=)
        self.height_hist = np.zeros(99, dtype=int)
        # sampled attitude angles of a time-domain loop pair (engine.forward / sensitivity then take geometry={name: value})
        self.angle_moves = angle_moves or []
        self.angles, self.angles0 = (dict(angles) if angles else None), (dict(angles) if angles else None)
        self.angle_hist = {m[0]: np.zeros(199, dtype=int) for m in self.angle_moves}
        self.pred, self.J = self._fwd(self.edges, self.sigma, height), self._sen(self.edges, self.sigma, height)
        std = channel_std(data, rel, add, add_scale, groups)
        self.misfit, self.like = rjmcmc.gauss_loglike(self.pred, data, std)
        self.prior = rjmcmc.model_log_prior(sp, vp, self.edges, self.sigma) + levels_log_prior(rel, o["rel_min"], o["rel_max"])
        if not o.get("add_independent"):
            self.prior = self.prior + levels_log_prior(add, o["add_min"], o["add_max"])
        if height is not None:
            self.prior -= math.log(2.0 * o["height_half_width"])
        self.prior_const_angles = -sum(math.log(2.0 * m[1]) for m in self.angle_moves)
        self.prior += self.prior_const_angles
        self.k_hist = np.zeros(o["K"] + 1, dtype=int)
        self.edge_hist = np.zeros(n_depth_bins, dtype=int)
        self.w = depth_bin_width
        self.n_accepted = 0
        self.trace = []

    def _fwd(self, e, s, z, geometry=None):
        kw = ({} if z is None else {"z": z})
        if self.angle_moves:
            kw["geometry"] = self.angles if geometry is None else geometry
        return self.engine.forward(e, s, **kw)

    def _sen(self, e, s, z, geometry=None):
        kw = ({} if z is None else {"z": z})
        if self.angle_moves:
            kw["geometry"] = self.angles if geometry is None else geometry
        return self.engine.sensitivity(e, s, **kw)

    def step(self, it):
        o, d = self.o, self.data
        height_p = None
        if self.height is None:
            action, idx, val, e_r, s_r, rel_p, add_p = propose(o, self.seed, self.b, it, self.edges, self.sigma, self.rel, self.add)
        else:
            action, idx, val, e_r, s_r, rel_p, add_p, height_p = propose(o, self.seed, self.b, it, self.edges, self.sigma, self.rel, self.add,
                                                                         self.height, self.height0)
        angles_p = propose_angles(self.angle_moves, self.seed, self.b, it, self.angles, self.angles0) if self.angle_moves else None
        if action != rjmcmc.NONE:
            pred_r, J_r = self._fwd(e_r, s_r, self.height), self._sen(e_r, s_r, self.height)
        else:
            pred_r, J_r = self.pred, self.J
        log_prop, C = newton(o, self.seed, self.b, it, self.vp, e_r, s_r, J_r, pred_r, d, self.rel, self.add, self.add_scale, self.groups)
        prop = np.exp(log_prop)
        pred_p = self._fwd(e_r, prop, height_p, angles_p)
        misfit_p, like_p = rjmcmc.gauss_loglike(pred_p, d, channel_std(d, rel_p, add_p, self.add_scale, self.groups))
        J_p = self._sen(e_r, prop, height_p, angles_p) if action in (rjmcmc.INSERT, rjmcmc.DELETE) else None
        log_ratio, acc, prior_p = accept(o, self.seed, self.b, it, self.sp, self.vp, action, e_r, s_r, log_prop, C, J_p, pred_p, d,
                                         rel_p, add_p, like_p, self.prior, self.like, self.add_scale, self.groups,
                                         (0.0 if self.height is None else -math.log(2.0 * o["height_half_width"])) + self.prior_const_angles)
        if acc and self.angle_moves:
            self.angles = angles_p
        if acc and self.height is not None:
            self.height = height_p
        if acc:
            self.edges, self.sigma, self.rel, self.add, self.pred = e_r, prop, rel_p, add_p, pred_p
            self.prior, self.like, self.misfit = prior_p, like_p, misfit_p
            self.J = J_p if J_p is not None else J_r
            self.n_accepted += 1
        self.trace.append((action, acc, self.sigma.size))
        self.k_hist[self.sigma.size] += 1
        for name, hw, _ in self.angle_moves:
            u = (self.angles[name] - (self.angles0[name] - hw)) / (2.0 * hw)
            nb = 199 if name.endswith("pitch") else 99
            if 0.0 <= u <= 1.0:
                self.angle_hist[name][min(int(math.floor(u * nb)), nb - 1)] += 1
        if self.height is not None:
            u = (self.height - (self.height0 - o["height_half_width"])) / (2.0 * o["height_half_width"])
            if 0.0 <= u <= 1.0:
                self.height_hist[min(int(math.floor(u * 99)), 98)] += 1
        ratio = self.sigma[1:] / self.sigma[:-1]
        for depth in self.edges[(ratio <= 0.5) | (ratio >= 1.5)]:
            self.edge_hist[min(int(depth // self.w), self.edge_hist.size - 1)] += 1
        return log_ratio
