#!/usr/bin/env python
"""Joint proposals of SEVERAL error levels (build container only) -> tests/golden/error_levels.npz.

A time-domain data point carries one relative error per system x component and one additive error per system
(data/datapoint/TdemDataPoint.py:361-365); DataPoint.perturb (data/datapoint/DataPoint.py:531-573) moves each set with ONE
joint draw -- StatArray.perturb(imposePrior=True, log=True) -> StatArray.propose (statistics/StatArray.py:578-638): a
multivariate log-normal random walk, redrawn while the uniform(log) prior of the set gives -inf, the current values kept at the
10th redraw -- and DataPoint.probability (:454-489) sums the sets' log priors.  The reference's TdemDataPoint itself cannot be
built here (gatdaem1d is absent), but the StatArray / distribution objects it delegates to can: this script runs the
reference's own StatArray with the priors and proposals DataPoint.set_priors / set_proposals attach (:575-644: Uniform(min, max,
log=True), MvLogNormal(values, variance, linearSpace=True)) and records, for seeded generators,
  start [n], lo [n], hi [n], var [n]     the set-up
  draws [steps, n]                       the values after each perturb()
  log_prior [steps]                      prior.probability(values, log=True) after each step
Only numbers are stored (data), none of the reference's code.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference   # noqa: E402


def main():
    import numpy as np
    import_reference()
    from geobipy import StatArray, Distribution
    from numpy.random import Generator, PCG64DXSM
    out = {}
    cases = {
        "rel2": ([0.05, 0.05], [0.005, 0.005], [0.5, 0.5], [1e-6, 1e-6], 300),          # skytem_options
        "add2": ([2e-14, 2e-13], [1e-16, 1e-16], [1e-10, 1e-10], [1e-5, 1e-5], 300),
        "tight3": ([0.05, 0.02, 0.1], [0.045, 0.019, 0.09], [0.055, 0.021, 0.11], [4e-3, 1e-3, 2e-3], 400),   # redraws and "keep" paths
        "one": ([0.05], [0.001], [0.5], [1e-6], 50),
    }
    for name, (start, lo, hi, var, steps) in cases.items():
        prng = Generator(PCG64DXSM(20260928))
        x = StatArray(np.asarray(start, dtype=np.float64), "level")
        x.prior = Distribution("Uniform", np.asarray(lo), np.asarray(hi), log=True, prng=prng)
        x.proposal = Distribution("MvLogNormal", x, np.asarray(var), linearSpace=True, prng=prng)
        draws, lp = [], []
        for _ in range(steps):
            x.perturb(imposePrior=True, log=True)
            x.proposal.mean = x                      # DataPoint.perturb :565-566: the walk re-centres on the new value
            draws.append(np.asarray(x).copy())
            lp.append(float(x.probability(log=True)))
        out[name + "_start"], out[name + "_lo"], out[name + "_hi"], out[name + "_var"] = map(np.asarray, (start, lo, hi, var))
        out[name + "_draws"], out[name + "_log_prior"] = np.asarray(draws), np.asarray(lp)
    np.savez_compressed(os.path.join(HERE, "error_levels.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
