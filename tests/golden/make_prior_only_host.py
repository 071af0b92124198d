"""Host-sampler ensemble of prior-only chains (tests/prior_only.py: options, harness, statistics) -> tests/golden/prior_only_host.npz.

    python tests/golden/make_prior_only_host.py [chains] [iterations]

Per statistic: the ensemble's cell probabilities and their standard error from the spread BETWEEN chains (the states of one chain are
correlated; the chains are independent).  ~10 minutes on 16 cores with the defaults."""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import prior_only as po  # noqa: E402

BURN, THIN = 2000, 10


def one(args):
    seed, n = args
    return po.host_chain(seed, n, BURN, THIN)


if __name__ == "__main__":
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    with Pool(os.cpu_count()) as pool:
        res = pool.map(one, [(1000 + c, n) for c in range(chains)])
    out = dict(chains=chains, iterations=n, burn=BURN, thin=THIN, acceptance=np.array([r[1] for r in res]))
    for key in ("k", "depth", "value", "rel", "add"):
        per = np.stack([r[0][key] / max(1.0, r[0][key].sum()) for r in res])          # every chain's own cell probabilities
        out[key + "_p"] = per.mean(axis=0)
        out[key + "_se"] = per.std(axis=0, ddof=1) / np.sqrt(chains)
    out["interfaces_per_state"] = np.mean([r[0]["depth"].sum() / r[0]["k"].sum() for r in res])
    np.savez(os.path.join(HERE, "prior_only_host.npz"), **out)
    for key in ("k", "depth", "value", "rel", "add"):
        print(key, np.round(out[key + "_p"], 4), "se max", out[key + "_se"].max())
    print("acceptance", out["acceptance"].mean())
