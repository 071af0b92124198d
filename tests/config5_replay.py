"""CPU replay of device chains for BASELINE config 5 (test / bench CHECKER only; imports the oracle, never torch).

A device chain is keyed by (seed, global chain index): the host emulation of the three stage kernels (tests/rj_emul.py, built
on geobipy_amd/rjmcmc.py, which is pinned to the reference's own chains) around the C oracle's forward and Jacobian walks
the same chain from the same initial half-space.  ``replay`` runs complete chains in worker processes and returns, per
chain, the state at every checkpoint (layer count, accepted steps so far, misfit) and the final posterior histograms."""
import math
import multiprocessing as mp
import os

import numpy as np


def _one(spec):
    import rj_emul
    from geobipy_amd import rjmcmc
    from conftest import oracle_system
    from oracle import fdem_oracle as fo
    (o, seed, chain, system_name, z, data, sigma0, rel0, add0, log_mean, n_depth_bins, depth_bin_width, n_it, every, prior_kw) = spec[:15]
    fo.set_exact_jacobian(bool(spec[15]) if len(spec) > 15 else False)      # (worker processes: one chain at a time)

    class Engine:
        def __init__(self):
            if isinstance(system_name, str):
                self.sys = oracle_system(system_name)
            else:                                                   # (frequencies, tx orientation, tx moment, tx xyz, rx ...) arrays
                self.sys = fo.OracleSystem(*system_name)

        @staticmethod
        def _thk(edges):
            return np.r_[np.diff(np.r_[0.0, edges]), np.inf]

        def forward(self, edges, values):
            return fo.predicted_data(self.sys, values, self._thk(edges), z)

        def sensitivity(self, edges, values):
            J = fo.sensitivity(self.sys, values, self._thk(edges), z)
            return np.vstack([J.real, J.imag])

    sp = rjmcmc.StructurePrior(o["K"], prior_kw["minimum_depth"], prior_kw["maximum_depth"], prior_kw["minimum_thickness"], o["p"])
    vp = rjmcmc.ValuePrior(math.exp(log_mean), prior_kw["factor"], prior_kw["gradient_standard_deviation"], prior_kw["solve_gradient"],
                           prior_kw["solve_parameter"], prior_kw["parameter_limits"])
    c = rj_emul.Chain(o, seed, chain, Engine(), sp, vp, data, sigma0, rel0, add0, n_depth_bins, depth_bin_width)
    marks = []
    for it in range(n_it):
        c.step(it)
        if (it + 1) % every == 0:
            marks.append((c.sigma.size, c.n_accepted, c.misfit))
    return dict(chain=chain, marks=np.array(marks), k_hist=c.k_hist, edge_hist=c.edge_hist, sigma=c.sigma, misfit=c.misfit)


def specs_from_device(dc, rows, system_name, n_it, every, data, heights, system=None, exact=False):
    """Build the worker arguments for the sampled ``rows`` of a DeviceChains block right after its initialisation.
    ``system_name``: a system file of tests/golden, or None with ``system`` = a geobipy_amd.FdemSystem.  ``exact``: the CPU chain
    uses the true-derivative Jacobian (oracle test switch) -- for device chains run with exact_jacobian=True."""
    if system_name is None:
        s = system
        system_name = (np.asarray(s.frequencies), list(s.transmitter.orientation), np.asarray(s.transmitter.moment),
                       np.c_[s.transmitter.x, s.transmitter.y, s.transmitter.z], list(s.receiver.orientation),
                       np.asarray(s.receiver.moment), np.c_[s.receiver.x, s.receiver.y, s.receiver.z])
    o = dc._o
    eo = dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge, p=[o.p_birth, o.p_death, o.p_perturb, o.p_none],
              rel_sd=o.rel_sd[0], rel_min=o.rel_min[0], rel_max=o.rel_max[0], add_sd=o.add_sd[0], add_min=o.add_min[0],
              add_max=o.add_max[0], alpha=o.alpha)
    oo = dc.o
    prior_kw = dict(minimum_depth=oo["minimum_depth"], maximum_depth=oo["maximum_depth"], minimum_thickness=oo["minimum_thickness"],
                    factor=oo["factor"], gradient_standard_deviation=oo["gradient_standard_deviation"], solve_gradient=oo["solve_gradient"],
                    solve_parameter=bool(oo.get("solve_parameter", False)), parameter_limits=oo.get("parameter_limits"))
    sig0 = dc.sigma[:, 0].cpu().numpy()
    lm = dc.log_mean_prior.cpu().numpy()
    rel0, add0 = dc.rel[:, 0].cpu().numpy(), dc.add[:, 0].cpu().numpy()
    first = int(o.first_chain)
    return [(eo, int(o.seed), first + int(b), system_name, float(heights[b]), np.asarray(data[b], dtype=np.float64), float(sig0[b]),
             float(rel0[b]), float(add0[b]), float(lm[b]), dc.n_depth_bins, dc.depth_bin_width, int(n_it), int(every), prior_kw, bool(exact))
            for b in rows]


def start(specs, processes=None):
    """Launch the replays in `spawn`ed workers (the parent may hold a HIP context); returns (pool, async result)."""
    n = processes or max(1, min(len(specs), len(os.sched_getaffinity(0))))
    pool = mp.get_context("spawn").Pool(n)
    return pool, pool.map_async(_one, specs, chunksize=1)


def compare(results, dev_marks, dev_k_hist, dev_edge_hist, rows):
    """Per sampled chain: the first checkpoint at which (layers, accepted steps) differ from the device chain, or -1; and
    whether the final layer-count / interface-depth histograms are identical."""
    out = []
    for j, (r, b) in enumerate(zip(results, rows)):
        cpu = r["marks"]
        same = (cpu[:, 0] == dev_marks[:, j, 0]) & (cpu[:, 1] == dev_marks[:, j, 1])
        first = -1 if same.all() else int(np.argmin(same))
        hist = bool(np.array_equal(r["k_hist"], dev_k_hist[j]) and np.array_equal(r["edge_hist"], dev_edge_hist[j]))
        out.append(dict(row=int(b), first_divergent_checkpoint=first, histograms_equal=hist,
                        max_rel_misfit_diff=float(np.max(np.abs(cpu[:, 2] - dev_marks[:, j, 2]) / np.abs(dev_marks[:, j, 2])))))
    return out
