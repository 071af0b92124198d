#!/usr/bin/env python
"""The reference's restart / give-up bookkeeping (build container only) -> tests/golden/mcmc_reset.npz.

Runs the reference's own Inference1D.infer (inversion/Inference1D.py:633-688, with update :705-790 and reset :984-999) on
chains that cannot move -- parameter_limits pinned to the starting half-space value, so every proposal has zero prior
probability -- and on a chain that moves freely, with a short acceptance window (update_plot_every), and records after every
update() call what a restatement of that bookkeeping has to reproduce:

  iteration      the reference's (restarting) iteration counter
  accepted       the step's decision
  n_resets, n_zero   its _n_resets / _n_zero_acceptance counters
  burned_in
and, per run, the value infer() returned (failed) and the number of update() calls it made.
Only inputs / outputs are stored (data), none of the reference's code.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402


def main():
    import numpy as np
    import_reference()
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    out = {}
    # (name, window, reset_limit, n_markov_chains, pinned)
    runs = [("stuck1", 20, 1, 400, True), ("stuck2", 15, 2, 400, True), ("free", 20, 1, 150, False), ("short", 20, 1, 70, True)]
    for name, window, reset_limit, n_mc, pinned in runs:
        options = up.user_parameters.read(opt_file, data_directory=SUP)
        options["system_filename"] = SUP + "/resolve.stm"
        options.update(n_markov_chains=n_mc, save_hdf5=False, interactive_plot=True, update_plot_every=window, reset_limit=reset_limit)
        dp = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"]).datapoint(30)
        if pinned:
            probe = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
            probe.initialize(dp)
            s0 = float(probe.model.values[0])
            options["parameter_limits"] = np.r_[s0 / 1.0001, s0 * 1.0001]
            dp = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"]).datapoint(30)
        inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
        inf.initialize(dp)
        inf.plot_posteriors = lambda *a, **k: None          # infer() draws every iteration when interactive_plot is set
        inf.toPNG = lambda *a, **k: None
        inf.save_png = False
        inf.posterior_ax = []                               # reset() clears the axes of the interactive figure
        rows = []
        upd = inf.update

        def update():
            r = upd()
            rows.append((int(inf.iteration), int(bool(inf.accepted)), int(inf._n_resets), int(inf._n_zero_acceptance), int(bool(inf.burned_in))))
            return r
        inf.update = update
        failed = inf.infer(None)
        rows = np.array(rows, dtype=np.int64)
        out[name + "_trace"] = rows
        out[name + "_meta"] = np.array([window, reset_limit, n_mc, int(pinned), int(bool(failed)), rows.shape[0], int(inf._n_resets)], dtype=np.int64)
        if pinned:
            out[name + "_limits"] = np.asarray(options["parameter_limits"], dtype=np.float64)
        print(name, "failed", failed, "updates", rows.shape[0], "resets at", np.flatnonzero(np.diff(rows[:, 0]) < 0) + 1, "n_resets", inf._n_resets)
    np.savez_compressed(HERE + "/mcmc_reset.npz", **out)


if __name__ == "__main__":
    main()
