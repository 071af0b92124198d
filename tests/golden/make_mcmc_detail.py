#!/usr/bin/env python
"""Per-iteration internals of the reference's rjMCMC step (build container only) -> tests/golden/mcmc_detail.npz.

Groundwork for SURVEY row f-2 (batched rjMCMC): the same run as make_mcmc_trace.py (reference Inference1D, its
resolve_options and seed, 400 iterations), but recording for EVERY iteration the state a restatement of
Inference1D.accept_reject (inversion/Inference1D.py:537-631) has to reproduce:

  rng_state        PCG64DXSM state at the start of the iteration (so each iteration can be replayed alone)
  cur_*            current model (k, interface depths, conductivities), errors, prior / likelihood before the step,
                   and the datapoint's predicted data and (possibly stale) sensitivity matrix it carries
  action, a_index, a_value      RectilinearMesh1D.perturb outcome (mesh/RectilinearMesh1D.py:993-1120)
  rem_*            remapped model;  H, mean          stochastic-Newton proposal (model/Model.py:368-419)
  prop_*           proposed conductivities and error levels
  t_misfit, t_dp_prior, t_model_prior, t_like, q_fwd, q_rev, accepted   the terms of the acceptance ratio
Only inputs / outputs are stored (data), none of the reference's code.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402

N_ITER = 400        # iterations recorded in full detail
N_LONG = 3000       # iterations of the same run whose decisions / layer counts / misfits are recorded
LMAX = 30
ACTIONS = {"none": 0, "insert": 1, "delete": 2, "perturb": 3}


def main():
    global N_LONG
    import numpy as np
    deep = len(sys.argv) > 1 and sys.argv[1] in ("deep", "value")   # decisions-only variants on other soundings:
    value = len(sys.argv) > 1 and sys.argv[1] == "value"            # "deep": birth-heavy -> mcmc_deep.npz; "value": prior on the
                                                                     # conductivities instead of on their gradient -> mcmc_value.npz
    import_reference()
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up
    from geobipy.src.classes.model.Model import Model

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    options = up.user_parameters.read(opt_file, data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    options.update(n_markov_chains=N_LONG, save_hdf5=False, interactive_plot=True, update_plot_every=100000)
    if value:
        options.update(solve_parameter=True, solve_gradient=False)
        N_LONG = 1000
        options["n_markov_chains"] = N_LONG
    elif deep:
        options.update(probability_of_birth=0.5, probability_of_death=0.1, probability_of_perturb=0.2,
                       probability_of_no_change=0.2)
        N_LONG = 1200
        options["n_markov_chains"] = N_LONG
    dp = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"]).datapoint(45 if value else (60 if deep else 30))
    prng = get_prng(seed=options["seed"])
    inf = Inference1D(prng=prng, world=None, **options)
    inf.initialize(dp)

    cur = {}

    def pad(a, n=LMAX):
        out = np.zeros(n)
        a = np.atleast_1d(np.asarray(a, dtype=float))
        out[: a.size] = a
        return out

    def padm(a):
        out = np.zeros((LMAX, LMAX))
        a = np.atleast_2d(np.asarray(a, dtype=float))
        out[: a.shape[0], : a.shape[1]] = a
        return out

    snp0 = Model.stochastic_newton_perturbation
    pp0 = Model.proposal_probabilities
    mp0 = Model.probability

    def snp(self, *a, **k):
        rem, per = snp0(self, *a, **k)
        act = rem.mesh.action
        cur.update(action=ACTIONS[act[0]], a_index=int(act[1]), a_value=float(np.squeeze(act[2])),
                   rem_k=int(rem.nCells.item()), rem_edges=pad(np.asarray(rem.mesh.edges)[1:-1]), rem_sigma=pad(rem.values),
                   H=padm(per.values.proposal.variance), mean=pad(per.values.proposal.mean), prop_sigma=pad(per.values))
        return rem, per

    def pp(self, *a, **k):
        out = pp0(self, *a, **k)
        cur.update(q_fwd=float(out[0]), q_rev=float(out[1]))
        return out

    def mp(self, *a, **k):
        out = mp0(self, *a, **k)
        cur["t_model_prior"] = float(out)
        return out

    Model.stochastic_newton_perturbation, Model.proposal_probabilities, Model.probability = snp, pp, mp

    keys = ["cur_J", "cur_pred", "rng_state", "cur_k", "cur_edges", "cur_sigma", "cur_rel", "cur_add", "cur_prior", "cur_like", "cur_misfit",
            "action", "a_index", "a_value", "rem_k", "rem_edges", "rem_sigma", "H", "mean", "prop_sigma", "prop_rel",
            "prop_add", "t_model_prior", "q_fwd", "q_rev", "accepted", "new_k", "new_misfit", "new_prior", "new_like"]
    rec = {k: [] for k in keys}
    defaults = dict(action=-1, a_index=0, a_value=0.0, rem_k=0, rem_edges=np.zeros(LMAX), rem_sigma=np.zeros(LMAX),
                    H=np.zeros((LMAX, LMAX)), mean=np.zeros(LMAX), prop_sigma=np.zeros(LMAX), t_model_prior=np.nan,
                    q_fwd=np.nan, q_rev=np.nan)
    for it in range(N_ITER):
        st = prng.bit_generator.state
        s, inc = st["state"]["state"], st["state"]["inc"]
        m64 = (1 << 64) - 1
        rec["rng_state"].append(np.array([s >> 64, s & m64, inc >> 64, inc & m64, st["has_uint32"], st["uinteger"]],
                                         dtype=np.uint64))
        Jc = np.zeros((12, LMAX))
        Jm = np.asarray(inf.datapoint.sensitivity_matrix, dtype=float)
        Jc[:, : Jm.shape[1]] = Jm
        rec["cur_J"].append(Jc)
        rec["cur_pred"].append(np.asarray(inf.datapoint.predictedData, dtype=float).copy())
        rec["cur_k"].append(int(inf.model.nCells.item()))
        rec["cur_edges"].append(pad(np.asarray(inf.model.mesh.edges)[1:-1]))
        rec["cur_sigma"].append(pad(inf.model.values))
        rec["cur_rel"].append(float(inf.datapoint.relative_error.item()))
        rec["cur_add"].append(float(inf.datapoint.additive_error.item()))
        rec["cur_prior"].append(float(inf.prior))
        rec["cur_like"].append(float(inf.likelihood))
        rec["cur_misfit"].append(float(inf.data_misfit))
        cur.clear()
        cur.update(defaults)
        # the proposed datapoint is a deepcopy made inside accept_reject; capture its errors through perturb()
        from geobipy.src.classes.data.datapoint.DataPoint import DataPoint
        per0 = DataPoint.perturb

        def per(self):
            per0(self)
            cur.update(prop_rel=float(self.relative_error.item()), prop_add=float(self.additive_error.item()))
        DataPoint.perturb = per
        inf.accept_reject()
        DataPoint.perturb = per0
        inf.update()
        for k in ["action", "a_index", "a_value", "rem_k", "rem_edges", "rem_sigma", "H", "mean", "prop_sigma",
                  "t_model_prior", "q_fwd", "q_rev"]:
            rec[k].append(cur[k])
        rec["prop_rel"].append(cur.get("prop_rel", np.nan))
        rec["prop_add"].append(cur.get("prop_add", np.nan))
        rec["accepted"].append(bool(inf.accepted))
        rec["new_k"].append(int(inf.model.nCells.item()))
        rec["new_misfit"].append(float(inf.data_misfit))
        rec["new_prior"].append(float(inf.prior))
        rec["new_like"].append(float(inf.likelihood))
    long_acc, long_k, long_misfit = list(rec["accepted"]), list(rec["new_k"]), list(rec["new_misfit"])
    for it in range(N_ITER, N_LONG):
        inf.accept_reject()
        inf.update()
        long_acc.append(bool(inf.accepted))
        long_k.append(int(inf.model.nCells.item()))
        long_misfit.append(float(inf.data_misfit))
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["long_accepted"], out["long_k"], out["long_misfit"] = np.array(long_acc), np.array(long_k), np.array(long_misfit)
    out["data"] = np.asarray(dp.data, dtype=float)
    out["z"] = np.float64(dp.z.item())
    out["halfspace"] = np.asarray(inf.halfspace, dtype=float)
    out["options"] = np.array([options[k] for k in ["maximum_number_of_layers", "minimum_depth", "maximum_depth",
                                                     "minimum_thickness", "probability_of_birth", "probability_of_death",
                                                     "probability_of_perturb", "probability_of_no_change",
                                                     "covariance_scaling", "gradient_standard_deviation", "factor",
                                                     "minimum_relative_error", "maximum_relative_error",
                                                     "minimum_additive_error", "maximum_additive_error",
                                                     "relative_error_proposal_variance",
                                                     "additive_error_proposal_variance"]], dtype=float)
    if deep:     # decisions-only fixture: the starting state (iteration 0) and the whole decision sequence
        keep = ["rng_state", "cur_k", "cur_edges", "cur_sigma", "cur_rel", "cur_add", "cur_prior", "cur_like",
                "cur_misfit", "cur_J", "cur_pred"]
        out = {k: (v[:1] if k in keep else v) for k, v in out.items()
               if k in keep or k in ("long_accepted", "long_k", "long_misfit", "data", "z", "halfspace", "options")}
        np.savez_compressed(HERE + ("/mcmc_value.npz" if value else "/mcmc_deep.npz"), **out)
        print("long run: acceptance", out["long_accepted"].mean(), "k max", out["long_k"].max())
        return
    else:
        # the posteriors the reference accumulated over the N_LONG iterations (Inference1D.update -> Model.update_posteriors,
        # model/Model.py:819-847, mesh/RectilinearMesh1D.py:1595-1610); the chain never burns in on this sounding, so nothing
        # is reset: layer-count histogram, interface-depth histogram (0.5 m cells to 220 m), conductivity-depth hit map
        # [250 value cells (log10 relative to the prior mean) x 440 depth cells]
        m = inf.model
        hv = m.values.posterior
        out["post_ncells"] = np.asarray(m.mesh.nCells.posterior.counts, dtype=np.int64)
        out["post_edges"] = np.asarray(m.mesh.edges.posterior.counts, dtype=np.int64)
        out["post_values"] = np.asarray(hv.counts, dtype=np.int32)
        out["post_values_x_edges"] = np.asarray(hv.mesh.x.edges, dtype=float)
        out["post_values_relative_to"] = np.float64(np.asarray(hv.mesh.x.relative_to).item())
        out["post_burned_in"] = np.bool_(inf.burned_in)
        # error-level posteriors (DataPoint.set_posteriors / update_posteriors): 99 cells, uniform in log10 between the prior bounds
        for name, key in (("relative_error", "post_rel"), ("additive_error", "post_add")):
            hp = getattr(inf.datapoint, name).posterior
            hp = hp[0] if isinstance(hp, list) else hp
            out[key] = np.asarray(hp.counts, dtype=np.int64)
            out[key + "_edges"] = np.asarray(hp.mesh.edges, dtype=float) + np.asarray(hp.mesh.relative_to, dtype=float).item()
        np.savez_compressed(HERE + "/mcmc_detail.npz", **out)
    a = out["action"]
    print("long run: acceptance", out["long_accepted"].mean(), "k max", out["long_k"].max())
    print("iterations", N_ITER, "actions none/insert/delete/perturb", [(a == i).sum() for i in range(4)],
          "accepted", out["accepted"].mean(), "k max", out["new_k"].max())


if __name__ == "__main__":
    main()
