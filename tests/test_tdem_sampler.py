"""Device-resident rjMCMC on time-domain data (geobipy_amd.tdem.TdemDeviceChains -> gbp_rj_run_td), one system.

No reference pin exists for the TDEM path (gatdaem1d is absent); what is checked here is that the sampler
around it is the FDEM-pinned one: a CPU chain built from rjmcmc.py with the same random streams and the TDEM forward /
Jacobian of TdemBatch walks the same chain; cached state = from-scratch evaluation; synthetic data are fitted."""
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import rj_emul
from geobipy_amd import rjmcmc
from geobipy_amd import rjmcmc_gpu as rg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OFFSET = (-13.0, 0.0, 2.0)
OPTIONS = dict(solve_gradient=True, maximum_number_of_layers=20, minimum_depth=1.0, maximum_depth=300.0, minimum_thickness=1.0,
               initial_relative_error=0.05, minimum_relative_error=0.005, maximum_relative_error=0.5,
               relative_error_proposal_variance=1e-6, additive_error_proposal_variance=1e-5, probability_of_birth=1.0 / 6.0,
               probability_of_death=1.0 / 6.0, probability_of_perturb=1.0 / 6.0, probability_of_no_change=0.5, covariance_scaling=0.5)


def _survey(B, seed=0, stm=("SkytemLM.stm",), offset=None, alt=(30.0, 40.0)):
    """Synthetic 3-layer soundings for the given systems; returns (systems, heights, data, add_scale, options, groups)."""
    from geobipy_amd.tdem import TdemBatch, TdemSystem
    systems = [TdemSystem(os.path.join(GOLDEN, f)) for f in stm]
    rng = np.random.default_rng(seed)
    K = 20
    sig, thk = np.ones((B, K)), np.zeros((B, K))
    sig[:, :3] = np.c_[10.0 ** rng.uniform(-2.5, -1.5, B), 10.0 ** rng.uniform(-1.2, -0.5, B), 10.0 ** rng.uniform(-2.5, -1.5, B)]
    thk[:, :2] = np.c_[rng.uniform(8, 25, B), rng.uniform(10, 40, B)]
    offset = OFFSET if offset is None else offset
    h = rng.uniform(alt[0], alt[1], B)
    clean = TdemBatch(systems, np.full(B, 3), sig, thk, h, offset).forward().cpu().numpy()
    scale, add0, add_group, rel_group = [], [], [], []
    col = g = 0
    for i, s in enumerate(systems):
        n = s.n_components * s.nwindows
        sc = np.sqrt(1e-3 / np.tile(s.off_time, s.n_components))
        add0.append(0.02 * np.abs(clean[:, col:col + n]).min(axis=1).mean() / sc.min() + 1e-30)   # a few % of the system's smallest gate
        scale += list(sc)
        add_group += [i] * n
        for c in range(s.n_components):
            rel_group += [g + c] * s.nwindows
        col, g = col + n, g + s.n_components
    scale, add0, groups = np.array(scale), np.array(add0), (np.array(rel_group), np.array(add_group))
    std = np.sqrt((0.05 * clean) ** 2 + (add0[groups[1]] * scale) ** 2)
    data = clean + rng.normal(size=clean.shape) * std
    one = len(systems) == 1
    opts = dict(OPTIONS, initial_additive_error=add0.item() if one else list(add0),
                minimum_additive_error=(add0 / 30.0).item() if one else list(add0 / 30.0),
                maximum_additive_error=(add0 * 30.0).item() if one else list(add0 * 30.0))
    return (systems[0] if one else systems), h, data, scale, opts, groups


@pytest.mark.parametrize("stm", [("SkytemLM.stm",), ("SkytemHM.stm", "SkytemLM.stm")])
def test_tdem_chains_equal_cpu_chains_with_the_same_seeds(stm):
    """One moment (one relative, one additive level) and the two SkyTEM moments together (merged frequency-domain handle,
    two relative and two additive levels proposed jointly)."""
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    B, n_it = 3, 250 if len(stm) == 1 else 150
    s, h, data, scale, opts, groups = _survey(B, seed=3, stm=stm)
    assert np.all(data > 0)
    dc = TdemDeviceChains(s, h, data, OFFSET, seed=77, **opts)
    assert np.allclose(dc.add_scale.cpu().numpy(), scale) and np.array_equal(dc.rel_group.cpu().numpy(), groups[0])
    assert dc.n_rel_groups == len(stm) and dc.n_add_groups == len(stm)

    class Engine:                       # TDEM forward / Jacobian of one sounding through TdemBatch (B = 1)
        def __init__(self, z):
            self.z = z

        def _batch(self, e, v):
            K = 20
            sg, th = np.ones((1, K)), np.zeros((1, K))
            sg[0, : v.size], th[0, : v.size - 1] = v, np.diff(np.r_[0.0, e])
            return TdemBatch(s, np.array([v.size]), sg, th, np.array([self.z]), OFFSET)

        def forward(self, e, v):
            return self._batch(e, v).forward().cpu().numpy()[0].copy()

        def sensitivity(self, e, v):
            return self._batch(e, v).sensitivity().cpu().numpy()[0][:, : v.size].copy()

    o, G = dc._o, len(stm)
    eo = dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge, p=[o.p_birth, o.p_death, o.p_perturb, o.p_none],
              rel_sd=np.array(o.rel_sd[:G]), rel_min=np.array(o.rel_min[:G]), rel_max=np.array(o.rel_max[:G]),
              add_sd=np.array(o.add_sd[:G]), add_min=np.array(o.add_min[:G]), add_max=np.array(o.add_max[:G]), alpha=o.alpha)
    sig0 = dc.sigma[:, 0].cpu().numpy()
    chains = []
    for b in range(B):
        sp = rjmcmc.StructurePrior(dc.K, opts["minimum_depth"], opts["maximum_depth"], opts["minimum_thickness"], eo["p"])
        vp = rjmcmc.ValuePrior(sig0[b], 10.0, 1.5, True)
        rel0 = np.broadcast_to(np.atleast_1d(opts["initial_relative_error"]), (G,)).astype(float)
        add0 = np.atleast_1d(np.asarray(opts["initial_additive_error"], dtype=float))
        chains.append(rj_emul.Chain(eo, 77, b, Engine(h[b]), sp, vp, data[b], sig0[b], rel0, add0, dc.n_depth_bins, dc.depth_bin_width,
                                    add_scale=scale, groups=groups))
        assert np.isclose(chains[b].misfit, float(dc.misfit[b]), rtol=1e-8) and np.isclose(chains[b].prior, float(dc.prior[b]), rtol=1e-12)
    acts, accs, ks = [], [], []
    prev = dc.n_accepted.cpu().numpy().copy()
    for it in range(n_it):
        dc.step()
        now = dc.n_accepted.cpu().numpy()
        acts.append(dc.action.cpu().numpy().copy()); accs.append(now - prev); ks.append(dc.k.cpu().numpy().copy())
        prev = now.copy()
        for c in chains:
            c.step(it)
    acts, accs, ks = np.array(acts), np.array(accs), np.array(ks)
    for b, c in enumerate(chains):
        tr = np.array(c.trace)
        assert np.array_equal(tr[:, 0], acts[:, b]) and np.array_equal(tr[:, 1], accs[:, b]) and np.array_equal(tr[:, 2], ks[:, b]), b
        assert np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-5)
    assert accs.sum() > 0.15 * accs.size and set(np.unique(acts)) == {0, 1, 2, 3}


def test_tempest_two_components_one_system():
    """Tempest: x and z components of one system -> two relative levels, one additive level (n_rel_groups != n_add_groups);
    the device chain equals the CPU chain with the same seeds."""
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    off = (-107.0, 0.0, -45.0)
    B, n_it = 2, 120
    s, h, data, scale, opts, groups = _survey(B, seed=9, stm=("tempest.stm",), offset=off, alt=(115.0, 125.0))
    data = np.abs(data)                                  # (inline component changes sign; keep every channel active)
    opts = dict(opts, initial_relative_error=[0.05, 0.05], minimum_relative_error=[0.005, 0.005], maximum_relative_error=[0.5, 0.5],
                relative_error_proposal_variance=[1e-6, 1e-6])
    dc = TdemDeviceChains(s, h, data, off, seed=5, **opts)
    assert dc.n_rel_groups == 2 and dc.n_add_groups == 1 and dc.rel.shape == (B, 2) and dc.add.shape == (B, 1)

    class Engine:
        def __init__(self, z):
            self.z = z

        def _batch(self, e, v):
            sg, th = np.ones((1, 20)), np.zeros((1, 20))
            sg[0, : v.size], th[0, : v.size - 1] = v, np.diff(np.r_[0.0, e])
            return TdemBatch(s, np.array([v.size]), sg, th, np.array([self.z]), off)

        def forward(self, e, v):
            return self._batch(e, v).forward().cpu().numpy()[0].copy()

        def sensitivity(self, e, v):
            return self._batch(e, v).sensitivity().cpu().numpy()[0][:, : v.size].copy()

    o = dc._o
    eo = dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge, p=[o.p_birth, o.p_death, o.p_perturb, o.p_none],
              rel_sd=np.array(o.rel_sd[:2]), rel_min=np.array(o.rel_min[:2]), rel_max=np.array(o.rel_max[:2]),
              add_sd=np.array(o.add_sd[:1]), add_min=np.array(o.add_min[:1]), add_max=np.array(o.add_max[:1]), alpha=o.alpha)
    sig0 = dc.sigma[:, 0].cpu().numpy()
    chains = []
    for b in range(B):
        sp = rjmcmc.StructurePrior(dc.K, opts["minimum_depth"], opts["maximum_depth"], opts["minimum_thickness"], eo["p"])
        vp = rjmcmc.ValuePrior(sig0[b], 10.0, 1.5, True)
        chains.append(rj_emul.Chain(eo, 5, b, Engine(h[b]), sp, vp, data[b], sig0[b], np.array([0.05, 0.05]),
                                    np.atleast_1d(np.asarray(opts["initial_additive_error"], dtype=float)), dc.n_depth_bins,
                                    dc.depth_bin_width, add_scale=scale, groups=groups))
        assert np.isclose(chains[b].misfit, float(dc.misfit[b]), rtol=1e-8)
    acc = []
    prev = dc.n_accepted.cpu().numpy().copy()
    for it in range(n_it):
        dc.step()
        now = dc.n_accepted.cpu().numpy()
        acc.append(now - prev)
        prev = now.copy()
        for c in chains:
            c.step(it)
    acc = np.array(acc)
    for b, c in enumerate(chains):
        tr = np.array(c.trace)
        assert np.array_equal(tr[:, 1], acc[:, b]) and tr[-1, 2] == int(dc.k[b])
        assert np.allclose(c.rel, dc.rel[b].cpu().numpy(), rtol=1e-9) and np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-5)
    assert acc.sum() > 10


@pytest.mark.parametrize("case", ["skytem_tx_rx", "tempest_total_field", "tempest_positions", "skytem_heights", "central_loop_heights"])
def test_sampled_attitude_angles_on_the_device_equal_cpu_chains(case):
    """gbp_td_moves -- the reference's solve_transmitter_pitch / solve_receiver_pitch / _roll (pinned to the reference on the host:
    test_tdem_object_api.py::test_host_sampler_walks_the_reference_chain_with_loop_pair_moves): CPU chains with the same
    counter-based streams, every evaluation a TdemBatch of the request's geometry (current angles for the remapped model, proposed
    angles for the proposal; Tempest: + the free-space primary field of that geometry), walk the same chains as the device, whose
    kernels never see a new table -- a rotation only changes the per-chain mixing weights (and primary-field offset) that
    k_td_moves_propose forms: decisions, layer counts, angles to 1e-9 degrees, angle posteriors, highest-posterior angles.
    POSITION moves (round 4; the reference's solve_receiver_x / _z, solve_transmitter_z -- what its Tempest gallery example puts
    priors on): the CPU chains evaluate every request with a TdemBatch built for the request's OWN offset and height (new Hankel
    tables), the device keeps each chain's table set and evaluates it with the per-chain distance scale and effective height
    (gbp_td_moves.rho_scale, gbp_fdem_*_rows_scaled) -- the same chains: "tempest_positions" (dipole source: receiver x, z and pitch,
    total-field data with the primary field of the moved geometry), "skytem_heights" (loop source: receiver z and transmitter z)."""
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    from geobipy_amd.tdem_geometry import gaaem_tuple
    if case == "skytem_tx_rx":
        off, stm, alt, n_it = OFFSET, ("SkytemLM.stm",), (30.0, 40.0), 200
        mv = dict(solve_transmitter_pitch=True, maximum_transmitter_pitch_change=4.0, transmitter_pitch_proposal_variance=0.4,
                  solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.5)
        moves = [("tx_pitch", 4.0, 0.4), ("rx_pitch", 5.0, 0.5)]
        att0 = (1.0, 2.0, 0.0, -1.5, 1.0, 0.5)           # GA-AEM convention: tx roll, pitch, yaw, rx roll, pitch, yaw
    elif case in ("skytem_heights", "central_loop_heights"):
        # ("central_loop_heights": the receiver ON the transmitter loop's axis -- the table set has no horizontal distance, rho_set = 0;
        #  the distance scale was 0 / 0 = NaN there)
        off = OFFSET if case == "skytem_heights" else (0.0, 0.0, 2.0)
        stm, alt, n_it = ("SkytemLM.stm",), (30.0, 40.0), 150
        mv = dict(solve_receiver_z=True, maximum_receiver_z_change=1.0, receiver_z_proposal_variance=0.15,
                  solve_transmitter_z=True, maximum_transmitter_z_change=2.0, transmitter_z_proposal_variance=0.3)
        moves = [("dz", 1.0, 0.15), ("tx_z", 2.0, 0.3)]
        att0 = (0.0, 1.0, 0.0, 0.0, -0.5, 0.0)
    elif case == "tempest_positions":
        off, stm, alt, n_it = (-107.0, 0.0, -45.0), ("tempest.stm",), (115.0, 125.0), 150
        mv = dict(solve_receiver_x=True, maximum_receiver_x_change=1.0, receiver_x_proposal_variance=0.1,
                  solve_receiver_z=True, maximum_receiver_z_change=1.0, receiver_z_proposal_variance=0.1,
                  solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.3)
        moves = [("dx", 1.0, 0.1), ("dz", 1.0, 0.1), ("rx_pitch", 5.0, 0.3)]
        att0 = (0.0, 0.0, 0.0, 0.0, -1.0, 0.0)
    else:
        off, stm, alt, n_it = (-107.0, 0.0, -45.0), ("tempest.stm",), (115.0, 125.0), 150
        mv = dict(solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.3,
                  solve_receiver_roll=True, maximum_receiver_roll_change=3.0, receiver_roll_proposal_variance=0.2)
        moves = [("rx_pitch", 5.0, 0.3), ("rx_roll", 3.0, 0.2)]
        att0 = (0.0, 0.0, 0.0, 0.5, -1.0, 0.0)
    total_field = case.startswith("tempest")
    B = 3
    s, h, data, scale, opts, groups = _survey(B, seed=11, stm=stm, offset=off, alt=alt)
    systems = s if isinstance(s, list) else [s]
    base = dict(dx=off[0], dy=off[1], dz=off[2], tx_x=0.0, tx_y=0.0, tx_z=0.0, tx_roll=att0[0], tx_pitch=-att0[1], tx_yaw=-att0[2],
                rx_roll=att0[3], rx_pitch=-att0[4], rx_yaw=-att0[5])          # the loops' own convention (Loop_pair.Geometry negates)
    kw = {}
    if total_field:
        opts = dict(opts, initial_relative_error=[0.05, 0.05], minimum_relative_error=[0.005, 0.005], maximum_relative_error=[0.5, 0.5],
                    relative_error_proposal_variance=[1e-6, 1e-6], initial_additive_error=[1.0, 1.0], minimum_additive_error=[0.1, 0.1],
                    maximum_additive_error=[10.0, 10.0], additive_error_proposal_variance=[1e-6, 1e-6])
        chan_add = np.full(data.shape[1], 0.02 * np.abs(data).min())
        pf = TdemBatch(systems, np.ones(B, dtype=np.int32), np.ones((B, 2)), np.zeros((B, 2)), h, off, attitude=att0).primary_field()
        data = data + np.repeat(pf, [systems[0].nwindows] * systems[0].n_components, axis=1)      # total field
        data = np.abs(data)
        kw = dict(channel_additive=chan_add, primary_field=pf)
        groups = (groups[0], groups[0])
        scale = chan_add
    dc = TdemDeviceChains(systems, h, data, off, attitude=att0, seed=41, **dict(opts, **mv), **kw)
    assert [m_[0] for m_ in dc._moves] == [m_[0] for m_ in moves]

    class Engine:
        def __init__(self, z):
            self.z = z

        def _batch(self, e, v, geometry):
            K = 20
            sg, th = np.ones((1, K)), np.zeros((1, K))
            sg[0, : v.size], th[0, : v.size - 1] = v, np.diff(np.r_[0.0, e])
            g = gaaem_tuple(dict(dict(base, tx_z=self.z), **geometry))       # (the request's own height and offset: tables of their own)
            return TdemBatch(systems, np.array([v.size]), sg, th, np.array([g[0]]), tuple(g[4:7]), attitude=tuple(np.r_[g[1:4], g[7:10]]))

        def forward(self, e, v, geometry):
            b_ = self._batch(e, v, geometry)
            p_ = b_.forward().cpu().numpy()[0].copy()
            if total_field:
                p_ = p_ + np.repeat(b_.primary_field()[0], [systems[0].nwindows] * systems[0].n_components)
            return p_

        def sensitivity(self, e, v, geometry):
            return self._batch(e, v, geometry).sensitivity().cpu().numpy()[0][:, : v.size].copy()

    o = dc._o
    Gr, Ga = dc.n_rel_groups, dc.n_add_groups
    eo = dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge, p=[o.p_birth, o.p_death, o.p_perturb, o.p_none],
              rel_sd=np.array(o.rel_sd[:Gr]), rel_min=np.array(o.rel_min[:Gr]), rel_max=np.array(o.rel_max[:Gr]),
              add_sd=np.array(o.add_sd[:Ga]), add_min=np.array(o.add_min[:Ga]), add_max=np.array(o.add_max[:Ga]), alpha=o.alpha,
              add_independent=bool(o.additive_independent), add_centre=np.array(o.add_centre[:Ga]))
    assert bool(o.additive_independent) == total_field       # (the reference's treatment of Tempest's multipliers)
    sig0 = dc.sigma[:, 0].cpu().numpy()
    chains = []
    for b in range(B):
        sp = rjmcmc.StructurePrior(dc.K, opts["minimum_depth"], opts["maximum_depth"], opts["minimum_thickness"], eo["p"])
        vp = rjmcmc.ValuePrior(sig0[b], 10.0, 1.5, True)
        rel0 = np.broadcast_to(np.atleast_1d(opts["initial_relative_error"]), (Gr,)).astype(float)
        add0 = np.broadcast_to(np.atleast_1d(np.asarray(opts["initial_additive_error"], dtype=float)), (Ga,)).astype(float) if not total_field else np.ones(Ga)
        chains.append(rj_emul.Chain(eo, 41, b, Engine(h[b]), sp, vp, data[b], sig0[b], rel0, add0, dc.n_depth_bins, dc.depth_bin_width,
                                    add_scale=scale, groups=groups, angle_moves=moves,
                                    angles={m_[0]: (h[b] if m_[0] == "tx_z" else base[m_[0]]) for m_ in moves}))
        assert np.isclose(chains[b].misfit, float(dc.misfit[b]), rtol=1e-8) and np.isclose(chains[b].prior, float(dc.prior[b]), rtol=1e-12)
    accs, ks = [], []
    prev = dc.n_accepted.cpu().numpy().copy()
    for it in range(n_it):
        dc.step()
        now = dc.n_accepted.cpu().numpy()
        accs.append(now - prev); ks.append(dc.k.cpu().numpy().copy())
        prev = now.copy()
        ang = {n_: v_.cpu().numpy() for n_, v_ in dc.sampled_angles().items()}
        for c in chains:
            c.step(it)
            for n_ in ang:
                assert abs(c.angles[n_] - ang[n_][c.b]) < 1e-9, (it, c.b, n_)
    accs, ks = np.array(accs), np.array(ks)
    hist = dc.t["geom_hist"].cpu().numpy()
    for b, c in enumerate(chains):
        tr = np.array(c.trace)
        assert np.array_equal(tr[:, 1], accs[:, b]) and np.array_equal(tr[:, 2], ks[:, b]), b
        assert np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-5)
        for q, (n_, _, _) in enumerate(moves):
            assert np.array_equal(c.angle_hist[n_], hist[b, q]) and hist[b, q].sum() == n_it, (b, n_)
    centre = lambda n_, b_: h[b_] if n_ == "tx_z" else base[n_]
    moved = np.array([[abs(c.angles[m_[0]] - centre(m_[0], c.b)) / m_[1] for m_ in moves] for c in chains])
    assert accs.sum() > 0.1 * accs.size and moved.max() > 0.1
    best = dc.sampled_angles("best_geom")
    for m_ in moves:
        c0 = torch.as_tensor([centre(m_[0], b_) for b_ in range(B)], dtype=torch.float64, device=best[m_[0]].device)
        assert torch.all(torch.abs(best[m_[0]] - c0) <= m_[1] + 1e-12)
    if dc._pos_moves:                                             # the chains' table sets never changed: one per measured (rho, dz)
        g_now = dc.t["geom"].cpu().numpy()
        rho_now, rho_set = np.hypot(g_now[:, 4], g_now[:, 5]), np.hypot(off[0], off[1])
        want = rho_set / rho_now if rho_set > 0.0 else np.ones_like(rho_now)      # (an on-axis table set has no distance to scale: 1, not 0 / 0)
        assert np.allclose(dc.t["rho_scale"].cpu().numpy(), want, rtol=1e-14) and bool(torch.isfinite(dc.misfit).all())
        assert np.allclose(dc.t["height"].cpu().numpy(), g_now[:, 0] + 0.5 * (g_now[:, 6] - off[2]), rtol=1e-14)


def test_sampled_angles_survive_the_repacking_of_a_block():
    """infer() re-packs the block when chains finish (and restarts stuck ones): the per-chain tuples, weights, proposals and
    posteriors of gbp_td_moves travel with the rows -- the same chains, row for row, as a run that never re-packs."""
    from geobipy_amd.tdem import TdemDeviceChains
    B = 24
    s, h, data, scale, opts, groups = _survey(B, seed=5)
    mv = dict(solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.4,
              solve_receiver_roll=True, maximum_receiver_roll_change=3.0, receiver_roll_proposal_variance=0.3)
    runs = []
    for compact in (0.0, 0.95):
        dc = TdemDeviceChains(s, h, data, OFFSET, seed=9, reference_schedule=True,
                              burn_in_min_iterations=250, **dict(opts, n_markov_chains=500, **mv))
        sizes = []
        run0 = dc.run
        dc.run = lambda n, accumulate=True, dc=dc, run0=run0, sizes=sizes: (sizes.append(dc._c.B), run0(n, accumulate))[1]
        dc.infer(check_every=40, compact_below=compact, min_rows=2)
        runs.append((dc, sizes))
    (a, sa), (b, sb) = runs
    assert min(sa) == B and min(sb) < B                         # (the second run really re-packed)
    assert (a.status == 1).sum() >= 3 and (a.status == 2).sum() >= 3          # (both ways out of the schedule occur)
    for n in ("k", "sigma", "edges", "status", "burned_in_iteration", "n_accepted", "misfit", "geom", "best_geom", "geom_hist", "mix_w", "k_hist"):
        assert torch.equal(a.t[n], b.t[n]), n
    done = a.status == 1
    assert torch.all(a.t["geom_hist"][done].sum(dim=2) == 502)
    ang = a.sampled_angles()
    assert float(ang["rx_pitch"].abs().max()) > 0.3 and float(ang["rx_pitch"].abs().max()) <= 5.0


@pytest.mark.parametrize("stm", [("SkytemLM.stm",), ("SkytemHM.stm", "SkytemLM.stm")])
def test_tdem_chains_fit_synthetic_soundings_and_stay_coherent(stm):
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    B = 256
    s, h, data, scale, opts, groups = _survey(B, seed=5, stm=stm)
    dc = TdemDeviceChains(s, h, data, OFFSET, seed=1, **opts)
    m0 = dc.misfit.clone()
    dc.run(1500)
    k = dc.k.cpu().numpy()
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64))
    tb = TdemBatch(s, k, dc.sigma.cpu().numpy(), thk.cpu().numpy(), h, OFFSET, data=data, relative_error=dc.rel.cpu().numpy(),
                   additive_error=dc.add.cpu().numpy())
    chi2, logl = tb.forward_loglike()
    assert torch.allclose(tb.predicted, dc.pred, rtol=1e-8, atol=0) and torch.allclose(chi2, dc.misfit, rtol=1e-7)
    assert torch.allclose(logl, dc.like, rtol=1e-9)
    n_ch = data.shape[1]
    print("TDEM misfit: start median", float(m0.median()), "-> after 1500 iterations", float(dc.misfit.median()), "channels", n_ch,
          "mean k", k.mean())
    assert float(dc.misfit.median()) < 2.0 * n_ch < float(m0.median())
    assert int(dc.k_hist.sum()) == 1500 * B and (k >= 2).mean() > 0.8
