"""SURVEY row f-2 groundwork: the structural rjMCMC move replayed against the reference run, iteration by
iteration, from the recorded PCG64DXSM states (tests/golden/mcmc_detail.npz)."""
import os

import numpy as np
from numpy.random import Generator, PCG64DXSM

from conftest import GOLDEN


def generator_at(state_row):
    bg = PCG64DXSM(0)
    st = bg.state
    st["state"]["state"] = (int(state_row[0]) << 64) | int(state_row[1])
    st["state"]["inc"] = (int(state_row[2]) << 64) | int(state_row[3])
    st["has_uint32"], st["uinteger"] = int(state_row[4]), int(state_row[5])
    bg.state = st
    return Generator(bg)


def test_structural_moves_replay_the_reference_run():
    from geobipy_amd import rjmcmc
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    o = d["options"]
    prior = rjmcmc.StructurePrior(max_cells=o[0], min_edge=o[1], max_edge=o[2], min_width=o[3], probabilities=o[4:8])
    counts = np.zeros(4, dtype=int)
    for it in range(d["action"].size):
        prng = generator_at(d["rng_state"][it])
        prng.random()                                   # the draw hidden in the debug f-string (Inference1D.py:542)
        k = int(d["cur_k"][it])
        action, index, value, edges, values = rjmcmc.perturb_structure(prng, prior, d["cur_edges"][it, :k - 1],
                                                                       d["cur_sigma"][it, :k])
        assert action == d["action"][it], it
        counts[action] += 1
        rk = int(d["rem_k"][it])
        assert values.size == rk and edges.size == rk - 1
        assert np.array_equal(edges, d["rem_edges"][it, :rk - 1]) and np.array_equal(values, d["rem_sigma"][it, :rk])
        if action != rjmcmc.NONE:
            assert index == d["a_index"][it] and value == d["a_value"][it]
    assert counts.tolist() == [229, 70, 49, 52]


def test_stochastic_newton_proposal_replays_the_reference_run():
    """H, mean and the sampled conductivities of every iteration (Model.stochastic_newton_perturbation), with the
    Jacobian / forward of the remapped model supplied by the oracle (CPU tier; the GPU tier uses the kernels)."""
    from conftest import oracle_system
    from geobipy_amd import rjmcmc
    from oracle import fdem_oracle as fo
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    o = d["options"]
    sp = rjmcmc.StructurePrior(max_cells=o[0], min_edge=o[1], max_edge=o[2], min_width=o[3], probabilities=o[4:8])
    vp = rjmcmc.ValuePrior(value_mean=d["halfspace"].item(), factor=o[10], gradient_std=o[9])
    osys = oracle_system("resolve")
    data, z = d["data"], float(d["z"])
    for it in range(d["action"].size):
        prng = generator_at(d["rng_state"][it])
        prng.random()
        k = int(d["cur_k"][it])
        action, _, _, edges, values = rjmcmc.perturb_structure(prng, sp, d["cur_edges"][it, :k - 1], d["cur_sigma"][it, :k])
        rk = values.size
        if action == rjmcmc.NONE:
            J, pred = d["cur_J"][it][:, :rk], d["cur_pred"][it]
        else:
            thk = np.r_[np.diff(np.r_[0.0, edges]), np.inf]
            Jc = fo.sensitivity(osys, values, thk, z)
            J, pred = np.vstack([Jc.real, Jc.imag]), fo.predicted_data(osys, values, thk, z)
        std = np.sqrt((d["cur_rel"][it] * data) ** 2 + d["cur_add"][it] ** 2)
        mean, H = rjmcmc.stochastic_newton(vp, edges, values, J, pred, data, std, alpha=o[8])
        assert np.allclose(H, d["H"][it][:rk, :rk], rtol=1e-8, atol=1e-14), it
        assert np.allclose(mean, d["mean"][it][:rk], rtol=1e-8), it
        prop = rjmcmc.propose_values(prng, mean, H)
        assert np.allclose(prop, d["prop_sigma"][it][:rk], rtol=1e-7), it


class OracleEngine:
    """CPU-tier stand-in for the GPU kernels: forward / Jacobian of one sounding through the C oracle."""

    def __init__(self, name, z):
        from conftest import oracle_system
        from oracle import fdem_oracle as fo
        self.fo, self.sys, self.z = fo, oracle_system(name), z

    def _thk(self, edges):
        return np.r_[np.diff(np.r_[0.0, edges]), np.inf]

    def forward(self, edges, values):
        return self.fo.predicted_data(self.sys, values, self._thk(edges), self.z)

    def sensitivity(self, edges, values):
        J = self.fo.sensitivity(self.sys, values, self._thk(edges), self.z)
        return np.vstack([J.real, J.imag])


def chain_setup(d):
    from geobipy_amd import rjmcmc
    o = d["options"]
    sp = rjmcmc.StructurePrior(max_cells=o[0], min_edge=o[1], max_edge=o[2], min_width=o[3], probabilities=o[4:8])
    vp = rjmcmc.ValuePrior(value_mean=d["halfspace"].item(), factor=o[10], gradient_std=o[9])
    rp = rjmcmc.ErrorPrior(o[11], o[12], o[15])
    ap = rjmcmc.ErrorPrior(o[13], o[14], o[16])
    k = int(d["cur_k"][0])
    st = rjmcmc.ChainState(d["cur_edges"][0, :k - 1], d["cur_sigma"][0, :k], d["cur_rel"][0], d["cur_add"][0],
                           d["cur_pred"][0], d["cur_J"][0][:, :k], d["cur_prior"][0], d["cur_like"][0],
                           d["cur_misfit"][0])
    return sp, vp, rp, ap, st, o[8]


def run_chain(d, engine, n):
    from geobipy_amd import rjmcmc
    sp, vp, rp, ap, st, alpha = chain_setup(d)
    prng = generator_at(d["rng_state"][0])
    acc, ks, mis = [], [], []
    for it in range(n):
        a, st = rjmcmc.accept_reject(prng, st, d["data"], engine, sp, vp, rp, ap, alpha)
        acc.append(a)
        ks.append(st.k)
        mis.append(st.misfit)
    return np.array(acc), np.array(ks), np.array(mis), prng


def test_full_chain_reproduces_the_reference_decisions():
    """400 consecutive iterations from the reference's state and RNG stream at iteration 0, WITHOUT resynchronising:
    every accept / reject decision, the layer count and the misfit trajectory of the reference run are reproduced,
    and the generator ends in the state the reference's generator had."""
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    n = d["action"].size
    acc, ks, mis, prng = run_chain(d, OracleEngine("resolve", float(d["z"])), n)
    assert np.array_equal(acc, d["accepted"])
    assert np.array_equal(ks, d["new_k"])
    assert np.allclose(mis, d["new_misfit"], rtol=1e-9)
