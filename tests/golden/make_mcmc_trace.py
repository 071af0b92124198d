#!/usr/bin/env python
"""Capture what the reference's rjMCMC caller actually asks of the hot path (build container only).

Runs the IMPORTED reference's ``Inference1D`` (inversion/Inference1D.py: initialize :353-464, accept_reject
:537-631, update :705-790) on one Resolve sounding with the reference's own options file and seed, and records
every call it makes into the hot path, in order:

    forward      -> model (nlayers, sigma, thk), altitude           -> predictedData[2F]
    sensitivity  -> model                                           -> J[2F, L]
    data_misfit / likelihood(log=True) -> predictedData, data, relative / additive error -> chi2, logL

The fixture (tests/golden/mcmc_trace.npz) holds only these inputs and outputs; the GPU tests replay them as
ragged batches (birth / death / perturb proposals with 1..n layers, perturbed error levels) and check every
number against the parity bar -- row 15 of SURVEY 8(a), the caller contract.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402

N_ITER = 400
LMAX = 30


def main():
    import numpy as np
    import_reference()
    import geobipy
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up
    from geobipy.src.classes.data.datapoint.FdemDataPoint import FdemDataPoint

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    options = up.user_parameters.read(opt_file, data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    options["n_markov_chains"] = N_ITER
    options["save_hdf5"] = False
    options["interactive_plot"] = True
    options["update_plot_every"] = 100000
    data = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"])
    dp = data.datapoint(30)

    rec = {k: [] for k in ["kind", "nl", "sigma", "thk", "z", "pred", "J", "data", "rel", "add", "chi2", "logL"]}

    def pad(a):
        out = np.zeros(LMAX)
        out[: len(a)] = a
        return out

    def _log(kind, **kw):
        rec["kind"].append(kind)
        for k in rec:
            if k == "kind":
                continue
            v = kw.get(k)
            if k in ("sigma", "thk"):
                rec[k].append(np.zeros(LMAX) if v is None else pad(v))
            elif k == "pred" or k == "data":
                rec[k].append(np.full(12, np.nan) if v is None else np.asarray(v, dtype=float).copy())
            elif k == "J":
                J = np.zeros((12, LMAX))
                if v is not None:
                    J[:, : v.shape[1]] = v
                rec[k].append(J)
            else:
                rec[k].append(np.nan if v is None else float(np.asarray(v).item() if np.size(v) == 1 else v))

    f0, s0, m0, l0 = (FdemDataPoint.forward, FdemDataPoint.sensitivity, FdemDataPoint.data_misfit,
                      FdemDataPoint.likelihood)

    def model_args(mod):
        w = np.asarray(mod.mesh.widths, dtype=float).copy()
        w[-1] = 0.0
        return dict(nl=mod.mesh.nCells.item(), sigma=np.asarray(mod.values, dtype=float), thk=w)

    def forward(self, mod):
        f0(self, mod)
        _log("forward", z=self.z.item(), pred=np.asarray(self.predictedData), **model_args(mod))

    def sensitivity(self, mod, **kw):
        out = s0(self, mod, **kw)
        _log("sensitivity", z=self.z.item(), J=np.asarray(out), **model_args(mod))
        return out

    def data_misfit(self):
        out = m0(self)
        _log("misfit", pred=np.asarray(self.predictedData), data=np.asarray(self.data),
            rel=self.relative_error, add=self.additive_error, chi2=out)
        return out

    def likelihood(self, log):
        log_, log = log, _log
        out = l0(self, log_)
        log("likelihood", pred=np.asarray(self.predictedData), data=np.asarray(self.data),
            rel=self.relative_error, add=self.additive_error, logL=out)
        return out

    FdemDataPoint.forward, FdemDataPoint.sensitivity = forward, sensitivity
    FdemDataPoint.data_misfit, FdemDataPoint.likelihood = data_misfit, likelihood

    prng = get_prng(seed=options["seed"])
    inf = Inference1D(prng=prng, world=None, **options)
    inf.initialize(dp)
    n_init = len(rec["kind"])
    accepted, ks = [], []
    for it in range(N_ITER):
        inf.accept_reject()
        inf.update()
        accepted.append(bool(inf.accepted))
        ks.append(int(inf.model.nCells.item()))
    kinds = np.array(rec["kind"])
    out = {k: np.asarray(v) for k, v in rec.items() if k != "kind"}
    out["kind"] = np.array([{"forward": 0, "sensitivity": 1, "misfit": 2, "likelihood": 3}[k] for k in kinds], dtype=np.int8)
    out["n_init"] = np.int64(n_init)
    out["accepted"] = np.array(accepted)
    out["k"] = np.array(ks)
    np.savez_compressed(HERE + "/mcmc_trace.npz", **out)
    print("calls:", {k: int((kinds == k).sum()) for k in ["forward", "sensitivity", "misfit", "likelihood"]},
          "init calls", n_init, "acceptance", np.mean(accepted), "k range", min(ks), max(ks),
          "final chi2", float(inf.data_misfit))


if __name__ == "__main__":
    main()
