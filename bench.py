#!/usr/bin/env python
"""bench.py -- forward + likelihood evals/sec on synthetic FDEM soundings (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2]): 65 536 soundings x 10 zz frequencies x 8 layers, fp64,
sharded in contiguous blocks over the N ranks (strong scaling: the total is fixed).  One proposal ROUND
over the whole batch: every sounding gets a fresh conductivity vector (one of R pre-generated sets
resident in HBM, so nothing can be cached), the fused kernel evaluates forward solve + chi^2 +
log-likelihood, and the per-sounding summaries are gathered to rank 0 (RCCL, on a side stream so that it
overlaps the next round).  One STEP = ``rounds_per_step`` rounds, chosen from the warm-up's round time so
that the K timed steps last at least 1 s (R >= 8 rounds, wall >= 1 s; a 1 ms round x the
driver's 20 steps would be a 20 ms region); value = soundings x K x rounds_per_step / elapsed.  Inputs
are resident in HBM before the timed region.

The JSON line also carries
  roofline      frac = the flops the kernel EXECUTES -- SURVEY 8(d)'s min-flop count per (frequency, abscissa)
                point x the abscissae of each sounding's own window -- / kernel time (HIP events on the launch
                stream) / the MI355X fp64 vector peak (the path is fp64-VALU bound, not HBM/MFMA): a hardware
                fraction.  frac_all_abscissae_equivalent counts all 120 abscissae per frequency as SURVEY 8(d)
                does (useful-work equivalent); roofline.all_abscissae is the same batch with every abscissa
                evaluated; plus the algorithmic HBM bytes/s for reference;
  parity_vs_cpu within_bar against the CPU oracle with the test suite's bars; the run exits non-zero when false;
  cpu_baseline  the C oracle (a port of the reference's scalar algorithm) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_TOTAL = 65536
N_FREQ = 10
N_LAYERS = 8
N_SIGMA_SETS = 4
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X datasheet, per GPU
HBM_PEAK_GBPS = 8000.0
MIN_TIMED_SECONDS = 1.0          # R >= 8 rounds and wall >= 1 s
TIMED_MARGIN = 1.5               # the probe rounds are not the timed ones (clocks, contention between ranks): aim above the minimum


def flop_per_eval(L, F):
    """SURVEY 8(d) min-flop convention: (72 L + 33) flop per (frequency, abscissa) point, 120 points per zz frequency."""
    return (72 * L + 33) * F * 120


def flop_per_jacobian_point(L, exact):
    """Min-flop count (SURVEY 8d convention: real op 1, cadd 2, c x real 2, cmul 6, cdiv 11, cexp 5, csqrt 10) of ONE abscissa point
    of the prediction + Jacobian pass (gbp_fdem_point.h sens_point = calcFdemSensitivity1D + M1_1, fdem1d_numba.py:130-154, 222-303):
    basement layer 18; per layer above it 138 (csqrt 10, cexp 7, S / De / e De / Dd / Nn 14, 1 / Dd 11, its square 6, u e 6, the two
    bracket products 21 + 15, accumulate 14, dY 8, i b / 2u and W 14, Y 12; + 9 for the reference's extra term), suffix propagation
    6 per (layer, deeper layer) pair = 3 L (L - 1), surface factor + forward term 57, Hankel factor 13, row sums 2 L + 2."""
    if L == 1:
        return 80
    return 18 + (138 + (0 if exact else 9)) * (L - 1) + 3 * L * (L - 1) + 57 + 13 + 2 * L + 2


def bytes_per_eval(L, F, with_pred):
    """Algorithmic HBM traffic per eval: sigma[L] + thk[L] + height + nlayers + obs[2F] + rel + add read,
    chi2 + logL (+ pred[2F]) written."""
    rd = 8 * (2 * L + 1 + 2 * F + 2) + 4
    wr = 16 + (8 * 2 * F if with_pred else 0)
    return rd + wr


def measured_traffic(Btot, L, world):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in
    separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note) -- only for the profiled shape."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]", f"summary_bench_{Btot}x{N_FREQ}x{L}.json")))
    if world != 1 or not found:
        return None
    path = found[-1]                                   # the latest round's PMC passes
    d = json.load(open(path))["derived"]
    TRAFFIC_SOURCES["headline"] = os.path.relpath(path, ROOT)
    return d["hbm_fetch_bytes_x2_gfx950_correction"] + d["hbm_write_bytes_raw"]


def executed_points(handle, heights):
    """Mean number of filter abscissae per sounding the kernels EVALUATE: every sounding takes the abscissa window of its own
    altitude bin (1 m bins, gbp_fdem_system_bin_points); all abscissae for a handle without windows."""
    if not hasattr(handle, "bin_points"):
        return float(handle.npoints)
    fl, cnt = np.unique(np.floor(np.asarray(heights, dtype=np.float64)), return_counts=True)
    return float(sum(handle.bin_points(float(a)) * c for a, c in zip(fl, cnt)) / cnt.sum())


def profiled_utilisation(case):
    """VALU issue utilisation of the dominant kernel from the latest committed rocprofv3 PMC summary of `case` (profiles/r*/summary_<case>.json):
    read, not measured by this run -- the file is named in the line."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]", f"summary_{case}.json")))
    if not found:
        return None, None
    d = json.load(open(found[-1]))
    u = d.get("derived", {}).get("valu_issue_utilisation_fp64_4cyc")
    if u is None:                                        # sampler summaries: per kernel (+ the whole iteration's, summarise_case.py)
        ks = d.get("kernels", {})
        u = {k.split("<")[0].replace("rj::", ""): round(v["valu_issue_utilisation"], 4) for k, v in ks.items() if v.get("share_of_kernel_time", 0) > 0.05}
        agg = d.get("derived", {}).get("valu_issue_utilisation_aggregate")
        if agg is not None:
            u["aggregate"] = round(agg, 4)
    return u, os.path.relpath(found[-1], ROOT)


TRAFFIC_SOURCES = {}     # which committed profile each ``traffic`` figure of the line was read from (they are NOT measured in this run:
                         # PMC counters need rocprofv3 passes of their own -- profiles/run_profile_r4.sh regenerates them)


def usable_cores():
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (the GPU boxes hand a
    16-CPU quota to a 256-thread host, and oversubscribed OpenMP threads would misstate the baseline)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def rjmcmc_traffic(n_chains):
    """HBM bytes per lock-step iteration of the sampler from the committed PMC passes (profiles/r*/summary_rjmcmc_<chains>.json)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]", f"summary_rjmcmc_{n_chains}.json")))
    if not found:
        return None
    d = json.load(open(found[-1]))
    TRAFFIC_SOURCES[f"rjmcmc_{n_chains}"] = os.path.relpath(found[-1], ROOT)
    return d.get("derived", {}).get("hbm_bytes_per_iteration")


def tdem_traffic():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]", "summary_tdem_config4.json")))
    if not found:
        return None
    TRAFFIC_SOURCES["tdem"] = os.path.relpath(found[-1], ROOT)
    return json.load(open(found[-1])).get("derived", {}).get("hbm_bytes_per_launch")


def cpu_baseline(system, nl, sigma, thk, height, obs, sample, threads):
    from oracle import fdem_oracle as fo
    osys = fo.OracleSystem(system.frequencies, system.transmitter.orientation, system.transmitter.moment,
                           np.c_[system.transmitter.x, system.transmitter.y, system.transmitter.z],
                           system.receiver.orientation, system.receiver.moment,
                           np.c_[system.receiver.x, system.receiver.y, system.receiver.z])
    n = sample
    rel = np.full(n, 0.05)
    add = np.full(n, 5.0)
    fo.forward_loglike_batch(osys, nl[:64], sigma[:64], thk[:64], height[:64], obs[:64], rel[:64], add[:64],
                             nthreads=threads)  # warm-up (thread pool, page-in)
    t0 = time.perf_counter()
    pred, chi2, logl = fo.forward_loglike_batch(osys, nl[:n], sigma[:n], thk[:n], height[:n], obs[:n], rel, add,
                                                nthreads=threads)
    dt = time.perf_counter() - t0
    return n / dt, dt, (pred, chi2, logl)


def rjmcmc_extra(system, height, obs, device, Btot):
    """BASELINE config 5 numbers for the bench line: 8 192 soundings x 10 000 iterations on this GPU in both Jacobian modes, the
    1 024-chain block that one GPU gets when config 5 is spread over 8 (lock-step vs persistent driver), and a bounded CPU replay
    (checker: tests/config5_replay.py = host emulation + C oracle) of a few chains counting exact matches."""
    import torch
    from geobipy_amd import DeviceChains
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    opts = dict(maximum_number_of_layers=30, minimum_depth=1.0, maximum_depth=150.0, initial_relative_error=0.05,
                minimum_relative_error=0.001, maximum_relative_error=0.5, initial_additive_error=5.0, minimum_additive_error=3.0,
                maximum_additive_error=20.0, relative_error_proposal_variance=1e-6, additive_error_proposal_variance=1e-6,
                probability_of_birth=1.0 / 6.0, probability_of_death=1.0 / 6.0, probability_of_perturb=1.0 / 6.0,
                probability_of_no_change=0.5)
    obs_np = obs.cpu().numpy() if torch.is_tensor(obs) else np.asarray(obs)
    nrj, n_it = min(Btot, 8192), 10000
    out = {"unit": "chain-iterations/s", "soundings": nrj, "iterations": n_it}

    def timed(dc, n, warm=50):
        dc.run(warm); torch.cuda.synchronize(device)
        t0 = time.perf_counter(); dc.run(n); torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    F = system.nFrequencies

    def evaluation_census(dc, exact, n_count=64):
        """What one lock-step iteration evaluates, counted on the device over n_count further iterations from the moves the proposal
        kernel recorded (action 0 none, 1 insert, 2 delete, 3 perturb; k_r layers after the move): proposals that keep their
        dimension run the fused forward, chains whose structure changed the prediction + Jacobian pass at the remapped model,
        dimension changes the same pass at the proposal.  -> evaluations and algorithmic flops per chain-iteration."""
        acc = torch.zeros(3, dtype=torch.float64, device=device)
        fl = torch.zeros(2, dtype=torch.float64, device=device)
        Lmax = dc.K
        fwd_tab = torch.tensor([0.0] + [flop_per_eval(L_, F) for L_ in range(1, Lmax + 1)], dtype=torch.float64, device=device)
        jac_tab = torch.tensor([0.0] + [flop_per_jacobian_point(L_, exact) * F * 120 for L_ in range(1, Lmax + 1)], dtype=torch.float64, device=device)
        ksum = torch.zeros(1, dtype=torch.float64, device=device)
        for _ in range(n_count):
            dc.run(1)
            act, kr = dc.t["action"].long(), dc.t["k_r"].long()
            zero = torch.zeros_like(kr)
            a = torch.where(act != 0, kr, zero)                               # fm_dlogc at the remapped model
            c_ = torch.where((act == 1) | (act == 2), kr, zero)               # fm_dlogc at the proposal
            b_ = torch.where((act == 0) | (act == 3), kr, zero)               # fused forward + chi^2 at the proposal
            acc += torch.stack([(a > 0).sum(), (b_ > 0).sum(), (c_ > 0).sum()]).double()
            fl += torch.stack([fwd_tab[b_].sum(), jac_tab[a].sum() + jac_tab[c_].sum()])
            ksum += dc.t["k"].double().sum()
        n = float(dc.B * n_count)
        acc, fl = acc.cpu().numpy() / n, fl.cpu().numpy() / n
        return {"forward_evals_per_chain_iteration": float(acc[1]), "jacobian_passes_per_chain_iteration": float(acc[0] + acc[2]),
                "jacobian_at_remapped_model": float(acc[0]), "jacobian_at_proposal": float(acc[2]), "mean_layers": float(ksum.item() / n),
                "flop_forward": float(fl[0]), "flop_jacobian": float(fl[1]), "flop_per_chain_iteration": float(fl[0] + fl[1])}

    for key, exact in (("reference_jacobian", False), ("exact_jacobian", True)):
        dc = DeviceChains(system, height[:nrj], obs_np[:nrj], seed=1, exact_jacobian=exact, device=device, **opts)
        dt = timed(dc, n_it)
        sm = dc.summaries().cpu().numpy()
        out[key] = {"value": nrj * n_it / dt, "seconds": dt, "ms_per_lockstep_iteration": 1e3 * dt / n_it,
                    "acceptance": float(sm[:, 4].mean()), "mean_layers": float(sm[:, 3].mean()),
                    "median_misfit": float(np.median(dc.misfit.cpu().numpy()))}
        cen = evaluation_census(dc, exact)
        ach_all = out[key]["value"] * cen["flop_per_chain_iteration"] / 1e12
        win = executed_points(dc._h, height[:nrj]) / float(dc._h.npoints)
        ach = ach_all * win
        util, util_src = profiled_utilisation(f"rjmcmc_{nrj}")
        out[key]["roofline"] = {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": rjmcmc_traffic(nrj),
                                "achieved_all_abscissae_equivalent": ach_all,
                                "frac_all_abscissae_equivalent": ach_all / FP64_VECTOR_PEAK_TFLOPS,
                                "evaluated_share_of_abscissae": win,
                                "valu_issue_utilisation": util, "valu_issue_utilisation_source": util_src,
                                "valu_issue_utilisation_aggregate": util.get("aggregate") if isinstance(util, dict) else None,
                                "kernel": "k_rj_physics (fm_dlogc at the remapped models; fused forward / fm_dlogc at the proposals) "
                                          "+ propose / newton / accept stages", **cen,
                                "note": "frac = flops the kernels EXECUTE (min-flop count of SURVEY 8(d) per evaluated (frequency, abscissa, layer) "
                                        "point x the evaluations one iteration makes, counted on the device over 64 iterations) / wall time of the "
                                        "whole iteration (physics + per-chain stages) / fp64 vector peak; *_all_abscissae_equivalent counts all 120 "
                                        "abscissae per frequency as SURVEY 8(d) does (useful-work equivalent, not a hardware fraction)"}
        del dc
    out["roofline"] = out["reference_jacobian"]["roofline"]
    out["value"] = out["reference_jacobian"]["value"]          # the pinned parity mode is the headline of this object
    # config 5 over 8 GPUs = 1 024 chains per GPU: the small-block regime
    small = {}
    for key, mode in (("lockstep", 1), ("persistent", 2)):
        dc = DeviceChains(system, height[:1024], obs_np[:1024], seed=1, exact_jacobian=False, device=device, **opts)
        dc.run_mode = mode
        small[key] = 1024 * 2000 / timed(dc, 2000, warm=100)
        del dc
    out["block_of_1024"] = dict(small, unit="chain-iterations/s", note="one GPU's block when config 5 is spread over 8 GPUs; "
                                "gbp_rj_run_mode 1 / 2 walk bit-identical chains")
    # one GPU's block at 2 and 4 GPUs (gbp_rj_run's own choice of driver): the measured inputs of DESIGN.md 7's projection table
    blocks = {}
    for nb in (4096, 2048):
        if nb <= nrj:
            dc = DeviceChains(system, height[:nb], obs_np[:nb], seed=1, exact_jacobian=False, device=device, **opts)
            blocks[str(nb)] = nb * 3000 / timed(dc, 3000, warm=100)
            del dc
    out["blocks"] = dict(blocks, unit="chain-iterations/s", note="one GPU's block of config 5 at 2 / 4 GPUs, driver chosen by gbp_rj_run")
    # bounded replays on the host cores: 16 chains x 1 500 iterations in both Jacobian modes (the checker; never the measurement)
    try:
        import config5_replay
        n_rep, it_rep, every = 16, 1500, 100
        arms = {}
        for arm, exact in (("reference_jacobian", False), ("exact_jacobian", True)):
            dc = DeviceChains(system, height[:nrj], obs_np[:nrj], seed=1, exact_jacobian=exact, device=device, **opts)
            rows = np.linspace(0, nrj - 1, n_rep).astype(int)
            specs = config5_replay.specs_from_device(dc, rows, None, it_rep, every, obs_np, height, system=system, exact=exact)
            pool, pending = config5_replay.start(specs)
            try:
                rows_t = torch.as_tensor(rows, device=device)
                marks = []
                for _ in range(it_rep // every):
                    dc.run(every)
                    marks.append(torch.stack([dc.k[rows_t].double(), dc.n_accepted[rows_t].double(), dc.misfit[rows_t]], dim=1).cpu().numpy())
                res = pending.get(timeout=600)
            finally:
                pool.terminate()
            cmp = config5_replay.compare(res, np.array(marks), dc.k_hist[rows_t].cpu().numpy(), dc.edge_hist[rows_t].cpu().numpy(), rows)
            same = [c for c in cmp if c["first_divergent_checkpoint"] < 0 and c["histograms_equal"]]
            arms[arm] = {"exact_matches": len(same), "first_divergence_iteration": [int((c["first_divergent_checkpoint"] + 1) * every) for c in cmp
                                                                                  if c["first_divergent_checkpoint"] >= 0],
                         "median_rel_misfit_diff_of_matching_chains": float(np.median([c["max_rel_misfit_diff"] for c in same])) if same else None}
            del dc
        out["cpu_replay"] = {"chains": n_rep, "iterations": it_rep, "exact_matches": arms["reference_jacobian"]["exact_matches"],
                             "first_divergence_iteration": arms["reference_jacobian"]["first_divergence_iteration"], "arms": arms,
                             "full_size": "tests/test_config5_gpu.py re-runs 64 chains x 10 000 iterations against the CPU replay on every GPU test run "
                                          "of THIS build (its printed line: exact matches, divergent rows by iteration, drift of the matching chains)",
                             "note": "layer-count / interface-depth histograms and every checkpoint of (layers, accepted steps) equal to a CPU "
                                     "replay (rjmcmc.py stage emulation + C oracle, same random streams); the chains that diverge do so in "
                                     "ill-conditioned stretches of the chain map (cond of the Newton precision ~1e4 - 4e5): two CPU runs that differ "
                                     "by 1e-10-level perturbations of the oracle part diverge at the same rate (docs/notes_r1_r4.md)"}
    except Exception as e:                                       # the replay is a checker, never the measurement
        out["cpu_replay"] = {"error": repr(e)}
    out["note"] = ("BASELINE config 5 on ONE GPU: full birth/death/perturb rjMCMC (gbp_rj_run), 10-frequency synthetic survey; value = "
                   "reference-Jacobian-expression mode (the pinned parity mode); the reference does ~165 iterations/s per core")
    return out


def survey_extra(n_soundings=8192, n_lines=16, n_markov_chains=2000):
    """The whole user flow around the hot path, timed end to end in this process: a data file read from CSV
    (the reference's Resolve wedge file, rows repeated with 2 % noise), every sounding inverted under the reference's burn-in / stop
    schedule, the per-line results containers written.  (scripts/bench_survey.py is the stand-alone form.)"""
    import shutil, tempfile
    import torch
    from geobipy_amd import survey
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    src = os.path.join(golden, "resolve_glacial_clean.csv")
    hdr = open(src).readline().strip()
    raw = np.loadtxt(src, delimiter=",", skiprows=1)
    rng = np.random.default_rng(1)
    rows = raw[rng.integers(0, raw.shape[0], n_soundings)].copy()
    rows[:, 6:] *= 1.0 + 0.02 * rng.standard_normal((n_soundings, raw.shape[1] - 6))
    rows[:, 0] = np.repeat(np.arange(n_lines), -(-n_soundings // n_lines))[:n_soundings] + 100.0
    rows[:, 1] = np.arange(n_soundings)
    d = tempfile.mkdtemp()
    try:
        np.savetxt(os.path.join(d, "survey.csv"), rows, delimiter=",", header=hdr, comments="")
        from geobipy_amd import hdf
        kind = hdf.container_type("auto")          # the product's default here: "hdf5" wherever an HDF5 library can be loaded

        def once(timings, container):
            shutil.rmtree(os.path.join(d, "out"), ignore_errors=True)         # (the run before's 0.36 GB of files: not this run's time)
            t0 = time.perf_counter()
            ds = survey.FdemData.read_csv(os.path.join(d, "survey.csv"), os.path.join(golden, "resolve.stm"))
            t_csv = time.perf_counter() - t0
            res = survey.infer(os.path.join(golden, "resolve_options_small"), data=ds, n_markov_chains=n_markov_chains,
                               burn_in_min_iterations=n_markov_chains // 4, results_directory=os.path.join(d, "out"),
                               output=os.path.join(d, "summary.npz"), timings=timings, container=container)
            torch.cuda.synchronize()
            return res, time.perf_counter() - t0, t_csv
        _, dt_first, _ = once(None, kind)          # untimed warm-up, like the kernels' (library loads, thread pools, first-touch pages): reported beside
        res, dt, _ = once(None, kind)              # the number of record: no phase clocks, nothing synchronised for them
        size = sum(os.path.getsize(os.path.join(d, "out", f)) for f in os.listdir(os.path.join(d, "out")))
        phases = {}
        _, dt_p, t_csv = once(phases, kind)        # the same run again with device-synchronised phase clocks
        phases = dict({"csv_read": t_csv}, **phases)
        phases["other"] = dt_p - sum(phases.values())
        other = {}
        if kind == "hdf5":                         # the stand-in's time beside it
            _, dt_o, _ = once(None, "npz")
            other = {"npz_stand_in_seconds": dt_o,
                     "npz_stand_in_megabytes": sum(os.path.getsize(os.path.join(d, "out", f)) for f in os.listdir(os.path.join(d, "out"))) / 1e6}
        return dict({"value": n_soundings / dt, "unit": "soundings/s", "soundings": n_soundings, "lines": n_lines, "seconds": dt,
                     "n_markov_chains": n_markov_chains, "burned_in": int((res["status"] == 1).sum()),
                     "mean_iterations": float(np.mean(res["iterations"])), "container": kind, "container_writer": hdf.hdf5_writer() if kind == "hdf5" else "npz stand-in",
                     "container_megabytes": size / 1e6,
                     "first_run_seconds": dt_first,
                     "phases_seconds": {k_: round(v_, 4) for k_, v_ in phases.items()}, "phases_run_seconds": dt_p,
                     "note": "end to end in one process: CSV read, chains on the device under the reference's burn-in / stop schedule "
                             "(resolve_options_small, n_markov_chains as stated; the second run of the survey in this process -- first_run_seconds is the first), hit maps to the host in run-length form, per-line results "
                             "containers in the reference's layout -- real HDF5 files <line>.h5 through the HDF5 C library (hit maps one "
                             "deflated chunk per sounding, traces dense) where one can be loaded, else the .npz stand-in -- and the summary "
                             "file written to a temporary directory"}, **other)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def survey_at_size(n_soundings, n_lines, n_markov_chains):
    """ONE end-to-end run of the survey driver at a north-star size, with device-synchronised phase clocks and the
    process's peak host memory: the synthetic Resolve survey of survey_extra (wedge rows repeated with 2 % noise), the reference's
    burn-in / stop schedule, per-line HDF5 containers with the posterior payload (hit maps, full-length traces)."""
    import resource, shutil, tempfile
    import torch
    from geobipy_amd import hdf, survey
    golden = os.path.join(ROOT, "tests", "golden")
    src = os.path.join(golden, "resolve_glacial_clean.csv")
    hdr = open(src).readline().strip()
    raw = np.loadtxt(src, delimiter=",", skiprows=1)
    rng = np.random.default_rng(1)
    rows = raw[rng.integers(0, raw.shape[0], n_soundings)].copy()
    rows[:, 6:] *= 1.0 + 0.02 * rng.standard_normal((n_soundings, raw.shape[1] - 6))
    rows[:, 0] = np.repeat(np.arange(n_lines), -(-n_soundings // n_lines))[:n_soundings] + 100.0
    rows[:, 1] = np.arange(n_soundings)
    d = tempfile.mkdtemp()
    try:
        np.savetxt(os.path.join(d, "survey.csv"), rows, delimiter=",", header=hdr, comments="")
        kind = hdf.container_type("auto")
        phases = {}
        t0 = time.perf_counter()
        ds = survey.FdemData.read_csv(os.path.join(d, "survey.csv"), os.path.join(golden, "resolve.stm"))
        t_csv = time.perf_counter() - t0
        res = survey.infer(os.path.join(golden, "resolve_options_small"), data=ds, n_markov_chains=n_markov_chains,
                           burn_in_min_iterations=n_markov_chains // 4, results_directory=os.path.join(d, "out"),
                           output=os.path.join(d, "summary.npz"), timings=phases, container=kind)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        phases = dict({"csv_read": t_csv}, **phases)
        phases["other"] = dt - sum(phases.values())
        size = sum(os.path.getsize(os.path.join(d, "out", f)) for f in os.listdir(os.path.join(d, "out")))
        chains = phases.get("chains", 0.0)
        return {"soundings": n_soundings, "lines": n_lines, "n_markov_chains": n_markov_chains, "seconds": dt,
                "value": n_soundings / dt, "unit": "soundings/s", "burned_in": int((res["status"] == 1).sum()),
                "mean_iterations": float(np.mean(res["iterations"])), "container": kind, "container_megabytes": size / 1e6,
                "phases_seconds": {k_: round(v_, 3) for k_, v_ in phases.items()},
                "chains_share_of_wall": (chains / dt) if chains else None,
                "peak_host_memory_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
                "note": "one run, phase clocks on (the device is synchronised at the phase borders); peak host memory of the whole bench process so far"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


COMPACT_LIMIT = 3072     # bytes: the driver's record keeps the last 8 KB of stdout and parses the last line (BENCH_r05.parsed was null for a 22.8 KB line)
EXTRAS_FILE = os.path.join("gpurun_out", "bench_extras.json")


def _r(x, sig=6):
    """Floats of the printed line at `sig` significant digits (the full-precision numbers are in the extras file)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    x = float(x)
    return float(f"{x:.{sig}g}") if np.isfinite(x) else None


def compact_line(full, extras_file=EXTRAS_FILE):
    """The ONE line bench.py prints: the contract's keys, the headline roofline with one (value, frac) pair per other measured
    kernel, the CPU baseline and the parity verdict -- under COMPACT_LIMIT bytes.  Everything else the run measured (notes,
    definitions, survey phases, replay arms, the TDEM parity table) is `full`, written to `extras_file`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "rounds_per_step", "timed_seconds", "finite")
    line = {k: full[k] for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "soundings", "frequencies", "layers", "rounds_per_step", "hankel_eps_ppm",
                                          "abscissa_points_per_sounding_mean", "abscissa_points_all") if k in cfg}
    rf = full.get("roofline", {})
    roof = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "evals_per_launch",
                                   "flop_per_eval", "valu_issue_utilisation", "frac_all_abscissae_equivalent")}
    if "all_abscissae" in rf:
        roof["all_abscissae"] = {k: rf["all_abscissae"][k] for k in ("value", "frac")}
    others = {}
    rj = full.get("rjmcmc", {})
    if "value" in rj:
        rr = rj.get("roofline", {})
        others["rjmcmc_8192"] = {"value": rj["value"], "frac": rr.get("frac"), "unit": "chain-it/s"}
        if rr.get("valu_issue_utilisation_aggregate") is not None:
            others["rjmcmc_8192"]["valu_issue_utilisation"] = rr["valu_issue_utilisation_aggregate"]
        blk = rj.get("block_of_1024", {})
        if "lockstep" in blk:
            best = max(blk["lockstep"], blk["persistent"])
            others["rjmcmc_1024"] = {"value": best, "x8_projection": 8 * best,
                                     "note": "config 5 over 8 GPUs = 1 024 chains per GPU; x8 is a projection, latency-bound per chain"}
    for key in ("jacobian", "tdem", "config2", "shard_8192"):
        if key in full and "roofline" in full[key]:
            others[key] = {"value": full[key]["value"], "frac": full[key]["roofline"]["frac"]}
    if others:
        roof["other_kernels"] = others
    line["roofline"] = roof
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:160], "single_thread": {"value": cb["single_thread"]["value"]},
                                "parity_within_bar": cb.get("parity_within_bar")}
    if "parity_vs_cpu" in full:
        pv = full["parity_vs_cpu"]
        line["parity_vs_cpu"] = {k: pv[k] for k in ("within_bar", "max_abs_pred_ppm", "soundings_compared") if k in pv}
    sv = full.get("survey", {}).get("north_star_sizes", {})
    if "soundings_65536" in sv:
        line["survey"] = {k: {"seconds": sv[k]["seconds"], "chains_share_of_wall": sv[k]["chains_share_of_wall"]}
                          for k in ("config5_schedule_8192_x_10000", "soundings_65536") if k in sv and "seconds" in sv[k]}
    if "forced_collective" in full:                    # (--force-collective: the N > 1 round on a one-rank RCCL group)
        line["forced_collective"] = {k: full["forced_collective"][k] for k in ("backend", "world", "rounds", "gathered_equals_local")}
    line["extras_file"] = extras_file
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:                      # never print a line the driver cannot keep: drop the optional objects first
        for k in ("survey", "parity_vs_cpu"):
            line.pop(k, None)
        line["roofline"].pop("other_kernels", None)
    return line


def write_extras(full, path=EXTRAS_FILE):
    try:
        os.makedirs(os.path.dirname(os.path.join(ROOT, path)), exist_ok=True)
        with open(os.path.join(ROOT, path), "w") as f:
            json.dump(full, f, indent=1)
        return True
    except OSError:
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--rounds-per-step", type=int, default=0,
                    help="proposal rounds per step (0 = chosen so that the timed region is >= 1 s, SURVEY 8d)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--soundings", type=int, default=B_TOTAL, help="total soundings (default: BASELINE config)")
    ap.add_argument("--layers", type=int, default=N_LAYERS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-windowed", action="store_true", help="skip the extra opt-in abscissa-window measurement")
    ap.add_argument("--no-rjmcmc", action="store_true", help="skip the extra full-rjMCMC-step measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra Jacobian and time-domain measurements")
    ap.add_argument("--no-survey-sizes", action="store_true", help="skip the end-to-end survey runs at the north-star sizes (~30 s)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="soundings in the CPU baseline sample (0 = auto)")
    ap.add_argument("--extras-file", default=EXTRAS_FILE, help="where the run's full record goes (relative to the repository root; '' = nowhere)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: initialise RCCL with ONE rank and run the N > 1 exchange anyway (side stream + all_gather_into_tensor "
                         "per round) -- executes the multi-GPU code path on a single-GPU box; not the headline configuration")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # one process per GPU; the modulo only matters for the single-GPU functional test of the N > 1 code path
    # (GBP_BENCH_BACKEND=gloo with two ranks sharing one device), never for a real multi-GPU run
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("GBP_BENCH_BACKEND", "nccl")
    exchange = world > 1 or args.force_collective          # does a round end with the gather of the summaries?
    if exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from geobipy_amd import FdemBatch, synthetic
    from geobipy_amd.distributed import SummaryGather, shard

    Btot, L, F = args.soundings, args.layers, N_FREQ
    system = synthetic.syn10_system()
    start, Bloc = shard(Btot, rank, world)

    # ---- synthetic inputs: drawn for the WHOLE job with one seed, then sliced, so the workload does not
    # depend on N.  "Observed" data = forward(true model) with 5 % + 5 ppm noise; proposals = fresh sigma.
    nl, sigma_true, thk, height = synthetic.draw_models(Btot, L, seed=synthetic.SEED + 2)
    sl = slice(start, start + Bloc)
    truth = FdemBatch(system, nl[sl], sigma_true[sl], thk[sl], height[sl], device=device)
    clean = truth.forward().cpu().numpy()
    rngn = np.random.Generator(np.random.PCG64DXSM(synthetic.SEED + 3))   # one noise field for any N
    g1 = rngn.normal(size=(Btot, 2 * F))
    g2 = rngn.normal(size=(Btot, 2 * F))
    obs = clean * (1.0 + 0.05 * g1[sl]) + 5.0 * g2[sl]
    rel = np.full(Bloc, 0.05)
    add = np.full(Bloc, 5.0)
    sig_sets = [synthetic.redraw_sigma(Btot, L, seed=synthetic.SEED + 10 + i)[sl] for i in range(N_SIGMA_SETS)]
    batches = [FdemBatch(system, nl[sl], s, thk[sl], height[sl], data=obs, relative_error=rel, additive_error=add,
                         device=device) for s in sig_sets]
    gather = SummaryGather(Btot, 2, device, force_collective=args.force_collective)
    side = torch.cuda.Stream(device=device) if exchange else None

    def round_(i, pending):
        b = batches[i % N_SIGMA_SETS]
        chi2, logl = b.forward_loglike(want_pred=False)
        if not exchange:
            return None          # single rank: the summaries are already resident on "rank 0"
        # finish the previous round's gather, then start this one on the side stream behind the kernel
        if pending is not None:
            pending[0].wait()
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            work = gather.launch(chi2, logl)
        return (work,)

    def sync_all():
        torch.cuda.synchronize(device)
        if exchange:
            dist.barrier()
            torch.cuda.synchronize(device)

    # One STEP = rounds_per_step proposal rounds over the whole batch.  SURVEY 8(d) wants R >= 8 rounds and a timed region of at
    # least 1 s: the driver's --steps 20 of ONE 1 ms round would be a 20 ms region, so the rounds per step are chosen from the
    # warm-up's measured round time such that steps x rounds_per_step x round >= 1 s (every rank computes the same number from
    # the max over ranks).  evals = soundings x steps x rounds_per_step.
    pending = None
    for i in range(args.warmup):
        pending = round_(i, pending)
    if pending is not None:
        pending[0].wait()
        pending = None
    sync_all()
    tw0 = time.perf_counter()
    n_probe = 8
    for i in range(n_probe):
        pending = round_(i, pending)
    if pending is not None:
        pending[0].wait()
        pending = None
    sync_all()
    t_round = (time.perf_counter() - tw0) / n_probe
    if world > 1:
        tr = torch.tensor([t_round], dtype=torch.float64, device=device)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        t_round = float(tr[0])
    rounds_per_step = max(1, int(np.ceil(TIMED_MARGIN * MIN_TIMED_SECONDS / (args.steps * t_round)))) if args.rounds_per_step <= 0 else args.rounds_per_step
    if args.steps * rounds_per_step < 8:
        rounds_per_step = int(np.ceil(8 / args.steps))

    def step(i, pending):
        for r in range(rounds_per_step):
            pending = round_(i * rounds_per_step + r, pending)
        return pending

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        pending = step(i, pending)
    ev1.record()
    if pending is not None:
        pending[0].wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    n_rounds = args.steps * rounds_per_step
    kernel_ms = ev0.elapsed_time(ev1) / n_rounds        # launch stream only: the fused kernel (+ tiny copies), per round

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    if exchange:
        result = gather.finish()
    else:
        last = batches[(n_rounds - 1) % N_SIGMA_SETS]
        result = torch.stack([last.chi2, last.logL], dim=1)

    if rank == 0:
        evals = Btot * n_rounds
        value = evals / elapsed
        fpe_all = flop_per_eval(L, F)                      # all 120 abscissae per frequency
        pts_all = batches[0]._h.npoints
        pts_exec = executed_points(batches[0]._h, height[sl])  # what the default path evaluates: each sounding's own abscissa window
        fpe = (72 * L + 33) * pts_exec                     # flops the kernel EXECUTES per eval (same per-point count)
        per_launch_evals = int(np.ceil(Btot / world))
        achieved = per_launch_evals * fpe / (kernel_ms * 1e-3) / 1e12
        achieved_all = per_launch_evals * fpe_all / (kernel_ms * 1e-3) / 1e12
        bpe = bytes_per_eval(L, F, with_pred=False)
        util, util_src = profiled_utilisation(f"bench_{Btot}x{N_FREQ}x{L}") if world == 1 else (None, None)
        line = {
            "metric": "forward+likelihood evals/sec (whole node)",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "rounds_per_step": rounds_per_step,
            "ms_per_round": 1e3 * elapsed / n_rounds,
            "timed_seconds": elapsed,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{Btot} FDEM soundings x {F} zz frequencies x {L} layers, fused forward+chi2+logL, "
                                   f"contiguous shards over {world} GPU(s), gather of (chi2, logL) to rank 0 per round",
                       "soundings": Btot, "frequencies": F, "layers": L, "seed": synthetic.SEED,
                       "proposal_sets": N_SIGMA_SETS, "rounds_per_step": rounds_per_step,
                       "hankel_eps_ppm": batches[0].hankel_eps_ppm,
                       "abscissa_points_per_sounding_mean": pts_exec,
                       "abscissa_points_all": pts_all,
                       "step": "rounds_per_step proposal rounds over the whole batch (chosen so that the timed region is >= 1 s)"},
            "roofline": {
                "bound": "fp64_valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": measured_traffic(Btot, L, world),
                "flop_per_eval": fpe, "evals_per_launch": per_launch_evals, "kernel_ms": kernel_ms,
                "kernel": "k_fdem_forward<true, false>",
                "definition": "frac = flops the kernel EXECUTES (SURVEY 8(d) min-flop count (72 L + 33) per evaluated (frequency, abscissa) "
                              "point x the abscissae of every sounding's own window) / kernel time / fp64 vector peak -- a hardware fraction",
                "achieved_all_abscissae_equivalent": achieved_all,
                "frac_all_abscissae_equivalent": achieved_all / FP64_VECTOR_PEAK_TFLOPS,
                "flop_per_eval_all_abscissae": fpe_all,
                "all_abscissae_equivalent_note": "SURVEY 8(d)'s 730 800 flop/eval counts all 120 abscissae per frequency; the default path "
                                                 "evaluates each sounding's abscissa window only: a useful-work equivalent, NOT a hardware fraction",
                "valu_issue_utilisation": util, "valu_issue_utilisation_source": util_src,
                "hbm": {"algorithmic_bytes_per_eval": bpe,
                        "achieved_GBps": per_launch_evals * bpe / (kernel_ms * 1e-3) / 1e9,
                        "peak_GBps": HBM_PEAK_GBPS},
            },
            "finite": bool(torch.isfinite(result).all()),
        }
        if args.force_collective:
            last = batches[(n_rounds - 1) % N_SIGMA_SETS]
            line["forced_collective"] = {"backend": backend, "world": world, "rounds": n_rounds,
                                         "gathered_equals_local": bool(torch.equal(result, torch.stack([last.chi2, last.logL], dim=1)[:result.shape[0]])),
                                         "note": "one-rank process group: every round ends with all_gather_into_tensor on a side stream (the N > 1 path)"}
        if world == 1 and not args.no_windowed:
            # the same rounds with ALL 120 abscissae per frequency (FdemBatch(hankel_eps_ppm=0)): what the default's per-sounding
            # abscissa window leaves out is bounded by 1e-12 ppm per output (|rTE| <= 1); measured difference below
            xb = [FdemBatch(system, nl[sl], sg, thk[sl], height[sl], data=obs, relative_error=rel, additive_error=add,
                            device=device, hankel_eps_ppm=0.0) for sg in sig_sets]
            xsteps = max(10, n_rounds // 4)
            for i in range(3):
                xb[i % N_SIGMA_SETS].forward_loglike(want_pred=False)
            torch.cuda.synchronize(device)
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tw = time.perf_counter()
            x0.record()
            for i in range(xsteps):
                xb[i % N_SIGMA_SETS].forward_loglike(want_pred=False)
            x1.record()
            torch.cuda.synchronize(device)
            tw = time.perf_counter() - tw
            xms = x0.elapsed_time(x1) / xsteps
            c_e, l_e = xb[0].forward_loglike(want_pred=True)
            c_w, l_w = batches[0].forward_loglike(want_pred=True)
            torch.cuda.synchronize(device)
            xach = per_launch_evals * fpe_all / (xms * 1e-3) / 1e12          # (every abscissa evaluated: executed = SURVEY 8(d)'s count)
            pts = [batches[0]._h.bin_points(a) for a in (25.0, 35.0, 45.0)]
            line["abscissa_window"] = {
                "eps_ppm": batches[0].hankel_eps_ppm, "points_per_sounding_at_25_35_45_m": pts, "points_all_abscissae": xb[0]._h.npoints,
                "max_abs_diff_pred_ppm": float((xb[0].predicted - batches[0].predicted).abs().max()),
                "max_abs_diff_logL": float((l_w - l_e).abs().max()), "max_abs_diff_chi2": float((c_w - c_e).abs().max()),
                "all_abscissae": {"value": Btot * xsteps / tw, "unit": "evals/s", "kernel_ms": xms,
                                  "roofline_frac": xach / FP64_VECTOR_PEAK_TFLOPS, "achieved_TFLOPs": xach},
                "note": "default path: each sounding is evaluated with the filter abscissae whose terms can exceed eps_ppm in total at its "
                        "own altitude (1 m bins; bound |rTE| <= 1), independent of the batch; the full 120-point sums carry ~1e-8 ppm of "
                        "rounding error themselves and the parity bar is 1e-7 ppm.  all_abscissae = FdemBatch(hankel_eps_ppm=0) in the same run."}
            # (inside `roofline` so that a reader of the driver's parsed line sees both rates: with all abscissae evaluated the
            #  executed and the SURVEY 8(d) counts coincide)
            line["roofline"]["all_abscissae"] = {"value": Btot * xsteps / tw, "unit": "evals/s", "kernel_ms": xms, "achieved": xach,
                                                 "frac": xach / FP64_VECTOR_PEAK_TFLOPS,
                                                 "note": "FdemBatch(hankel_eps_ppm=0) in the same run: all 1 200 abscissa points per sounding"}
            del xb
        if world == 1 and not args.no_rjmcmc:
            # the caller of the hot path (BASELINE config 5): complete rjMCMC iterations, every chain resident on
            # the device (gbp_rj_run).  Reported next to the headline, not in it.
            line["rjmcmc"] = rjmcmc_extra(system, height, obs, device, Btot)
        if world == 1 and not args.no_extras:
            # the rows either side of the headline path, measured in the same run and reported next to it: the Jacobian
            # kernel on the same batch, and the time-domain path (rows 13-14, BASELINE config 4 shape)
            def per_call(fn, n):
                # (the GPU idles while the host builds a case: warm the clocks for ~50 ms before timing, or launches of ~0.1 ms
                #  read 8 % slow -- shard_8192 60 M evals/s cold against 66 M warm in the same process)
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 0.05:
                    fn()
                torch.cuda.synchronize(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record(); torch.cuda.synchronize(device)
                return e0.elapsed_time(e1) / n
            def roof(Bk, Lk, Fk, ms, kernel, pts_exec_k, pts_all_k):
                ach_all = Bk * flop_per_eval(Lk, Fk) / (ms * 1e-3) / 1e12
                ach = ach_all * pts_exec_k / pts_all_k
                return {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "flop_per_eval": (72 * Lk + 33) * pts_exec_k,
                        "achieved_all_abscissae_equivalent": ach_all, "frac_all_abscissae_equivalent": ach_all / FP64_VECTOR_PEAK_TFLOPS,
                        "flop_per_eval_all_abscissae": flop_per_eval(Lk, Fk),
                        "evals_per_launch": Bk, "kernel_ms": ms, "kernel": kernel}
            # BASELINE config 2 (4 096 soundings x 10 frequencies x 5 layers, one GPU) and one GPU's shard of config 3 (8 192 of
            # the 65 536 soundings at 8 GPUs): same kernel, same generator, smaller launches
            for key, Bk, Lk, note in (("config2", 4096, 5, "BASELINE config 2: 4 096 soundings x 10 frequencies x 5 layers, 1 GPU"),
                                      ("shard_8192", 8192, L, "one GPU's shard of the headline workload at 8 GPUs (8 192 soundings)"),
                                      ("shard_16384", 16384, L, "one GPU's shard of the headline workload at 4 GPUs"),
                                      ("shard_32768", 32768, L, "one GPU's shard of the headline workload at 2 GPUs")):
                nlk, sgk, thkk, hk = synthetic.draw_models(Bk, Lk, seed=synthetic.SEED + 2)
                cl = FdemBatch(system, nlk, sgk, thkk, hk, device=device).forward().cpu().numpy()
                ob = synthetic.noisy_observations(cl, seed=synthetic.SEED + 3)
                bk = [FdemBatch(system, nlk, synthetic.redraw_sigma(Bk, Lk, seed=synthetic.SEED + 10 + i), thkk, hk, data=ob,
                                relative_error=np.full(Bk, 0.05), additive_error=np.full(Bk, 5.0), device=device) for i in range(N_SIGMA_SETS)]
                it = [0]
                def one():
                    bk[it[0] % N_SIGMA_SETS].forward_loglike(want_pred=False); it[0] += 1
                ms = per_call(one, max(50, int(200e3 / Bk) * 10))
                line[key] = {"value": Bk / ms * 1e3, "unit": "evals/s", "soundings": Bk, "frequencies": F, "layers": Lk,
                             "ms_per_step": ms, "roofline": roof(Bk, Lk, F, ms, "k_fdem_forward<true, false>", executed_points(bk[0]._h, hk), bk[0]._h.npoints),
                             "note": note}
                del bk
            Jbuf = batches[0].sensitivity()
            ms = per_call(lambda: batches[0].sensitivity(out=Jbuf), 100)
            fpj_pt = flop_per_jacobian_point(L, exact=False)
            fpj_all = fpj_pt * F * 120
            ptsj = executed_points(batches[0]._h, height)
            fpj = fpj_pt * ptsj
            achj = Btot * fpj / (ms * 1e-3) / 1e12
            achj_all = Btot * fpj_all / (ms * 1e-3) / 1e12
            utilj, utilj_src = profiled_utilisation("jacobian_headline")
            line["jacobian"] = {"value": Btot / ms * 1e3, "unit": "Jacobians/s", "ms_per_launch": ms, "soundings": Btot,
                                "roofline": {"bound": "fp64_valu", "achieved": achj, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": achj / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "flop_per_eval": fpj,
                                             "flop_per_abscissa_point": fpj_pt,
                                             "achieved_all_abscissae_equivalent": achj_all,
                                             "frac_all_abscissae_equivalent": achj_all / FP64_VECTOR_PEAK_TFLOPS,
                                             "flop_per_eval_all_abscissae": fpj_all,
                                             "valu_issue_utilisation": utilj, "valu_issue_utilisation_source": utilj_src,
                                             "evals_per_launch": Btot, "kernel_ms": ms, "kernel": "k_fdem_sens<false, %d>" % (1 if L <= 8 else (2 if L <= 16 else 4)),
                                             "count": "builder's extension of SURVEY 8(d) to the prediction + Jacobian pass (flop_per_jacobian_point: "
                                                      "same per-operation weights) x the abscissae each sounding's window evaluates (frac) or all 120 per "
                                                      "frequency (*_all_abscissae_equivalent); 100 timed launches after a 50 ms warm-up"},
                                "note": "d pred / d ln sigma [2F x L] of the same batch (gbp_fdem_sensitivity, reference expression)"}
            del Jbuf
            golden = os.path.join(ROOT, "tests", "golden")
            from geobipy_amd.tdem import TdemBatch, TdemSystem
            Bt, Lt = 16384, 6
            nlt, sgt, tht, ht = synthetic.draw_models(Bt, Lt, seed=synthetic.SEED + 4)
            # BASELINE config 4 as specified: 16 384 soundings x 30 gates log-spaced 1e-5 ... 1e-2 s x 6 layers, z
            # component, dB/dt, SkyTEM-LM-like waveform (tests/golden/config4_30gates.stm); and the two-moment SkyTEM system of
            # the reference's fixtures (26 + 19 gates); default path (per-sounding abscissa windows, relative budget 1e-12) and
            # all 120 abscissae (hankel_eps=0)
            td = {}
            for key, files, kw in (("config4", ["config4_30gates.stm"], {}), ("config4_all", ["config4_30gates.stm"], dict(hankel_eps=0.0)),
                                   ("skytem", ["SkytemHM.stm", "SkytemLM.stm"], {}),
                                   ("skytem_all", ["SkytemHM.stm", "SkytemLM.stm"], dict(hankel_eps=0.0))):
                systems = [TdemSystem(os.path.join(golden, f)) for f in files]
                tb = TdemBatch(systems, nlt, sgt, tht, ht, (-13.0, 0.0, 2.0), device=device, **kw)
                ms = per_call(tb.forward, 100)
                nodes = sum(sy.node_frequencies().size * sy.n_components for sy in systems)
                td[key] = dict(ms=ms, nodes=nodes, gates=tb.nChannels, points=sum(h.npoints for h in tb._h),
                               points_at_35_m=sum(h.bin_points(35.0) for h in tb._h), points_exec=sum(executed_points(h, ht) for h in tb._h), pred=tb.predicted.clone() if key.startswith("config4") else None)
                del tb
            c4 = td["config4"]
            d4 = float(((c4["pred"] - td["config4_all"]["pred"]).abs() / td["config4_all"]["pred"].abs().max(dim=1, keepdim=True).values).max())
            ach_all = Bt * (72 * Lt + 33) * c4["points"] / (c4["ms"] * 1e-3) / 1e12
            ach = Bt * (72 * Lt + 33) * c4["points_exec"] / (c4["ms"] * 1e-3) / 1e12
            utilt, utilt_src = profiled_utilisation("tdem_config4")
            line["tdem"] = {"value": Bt / c4["ms"] * 1e3, "unit": "evals/s", "soundings": Bt, "layers": Lt, "gates": c4["gates"],
                            "spline_nodes": c4["nodes"], "ms_per_step": c4["ms"],
                            "roofline": {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": tdem_traffic(), "flop_per_eval": (72 * Lt + 33) * c4["points_exec"],
                                         "evals_per_launch": Bt, "kernel_ms": c4["ms"],
                                         "achieved_all_abscissae_equivalent": ach_all,
                                         "frac_all_abscissae_equivalent": ach_all / FP64_VECTOR_PEAK_TFLOPS,
                                         "flop_per_eval_all_abscissae": (72 * Lt + 33) * c4["points"],
                                         "valu_issue_utilisation": utilt, "valu_issue_utilisation_source": utilt_src,
                                         "kernel": "k_fdem_forward<false, false> on the spline nodes + k_td_apply (window operator)",
                                         "count": "builder's extension of SURVEY 8(d) (which defines the FDEM count only): (72 L + 33) flop per "
                                                  "(spline node, abscissa) point x the abscissae of each sounding's window (frac) or all 120 "
                                                  "(*_all_abscissae_equivalent) x the system's spline nodes; 100 timed launches after a 50 ms warm-up"},
                            "abscissa_window": {"eps_relative": 1e-12, "points_all_abscissae": c4["points"], "points_at_35_m": c4["points_at_35_m"],
                                                "all_abscissae_value": Bt / td["config4_all"]["ms"] * 1e3,
                                                "max_rel_diff_to_all_abscissae": d4},
                            "skytem_two_moments": {"value": Bt / td["skytem"]["ms"] * 1e3, "all_abscissae_value": Bt / td["skytem_all"]["ms"] * 1e3,
                                                   "gates": td["skytem"]["gates"], "spline_nodes": td["skytem"]["nodes"]},
                            "note": "BASELINE config 4: one eval = frequency-domain solve at the .stm file's spline nodes (FrequenciesPerDecade) x "
                                    "the sounding's abscissa window (of 120) through the forward kernel + the window operator; roofline.achieved "
                                    "counts the algorithmic flops of all 120 abscissae, as the headline's does; parity against the reference's CSV known "
                                    "answers: every gate within 1e-3 |ref| + 7e-5 peak (Tempest) / 1e-2 |ref| + 4e-5 peak, median 6e-4 (SkyTEM), "
                                    "tests/test_tdem.py, scripts/tdem_study/README.md"}
            # parity of the time-domain half against REAL gatdaem1d numbers (the reference's rendered gallery:
            # tests/golden/make_tdem_doc_pins.py, tests/test_tdem_doc_pins.py), per quantity, from this GPU in this run
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import test_tdem_doc_pins as tp
                pins = np.load(os.path.join(golden, "tdem_doc_pins.npz"))
                gr = tp.gpu_report(pins)
                rel = lambda a, b: float(abs(a / float(b) - 1.0))
                line["tdem"]["parity"] = {
                    "source": "gatdaem1d outputs printed in /root/reference/docs/_sources/examples/Datapoints/plot_{tempest,skytem}_datapoint.rst.txt "
                              "(tests/golden/tdem_doc_pins.npz); bars in tests/test_tdem_doc_pins.py",
                    "tempest_jacobian_30x30": dict(tp.jacobian_report(gr["J_batch"], pins["tempest_J"]),
                                                   bars={"max_abs_over_max": tp.J_ATOL_ALL, "max_abs_over_row_max": tp.J_ATOL_ROW, "median_rel": tp.J_MEDIAN_REL}),
                    "tempest_chi2_rel": rel(gr["tempest_chi2"], pins["tempest_chi2"]), "tempest_logl_rel": rel(gr["tempest_logl"], pins["tempest_logl"]),
                    "tempest_like_bar": tp.TEMPEST_LIKE_RTOL,
                    "tempest_best_halfspace_same_cell": bool(rel(gr["tempest_best_halfspace"], pins["tempest_best_halfspace"]) < 1e-6),
                    "skytem_chi2_rel": rel(gr["skytem_chi2"], pins["skytem_chi2"]), "skytem_logl_rel": rel(gr["skytem_logl"], pins["skytem_logl"]),
                    "skytem_like_bar": tp.SKYTEM_LIKE_RTOL,
                    "skytem_best_halfspace_same_cell": bool(rel(gr["skytem_best_halfspace"], pins["skytem_best_halfspace"]) < 1e-6),
                    "skytem_chi2_last_trial_rel": rel(gr["skytem_chi2_after_search"], pins["skytem_chi2_best_halfspace"]),
                    "skytem_last_trial_bar": tp.SKYTEM_LAST_TRIAL_RTOL,
                    "forward_csv_bars": "tests/test_tdem.py: every gate of 474 soundings within 1e-3 |ref| + 7e-5 peak (Tempest) / 1e-2 |ref| + 4e-5 peak (SkyTEM)",
                    "reference_allclose_criterion": "met for FDEM; vacuous for SkyTEM (atol 1e-8 >> 1e-11 data); NOT met for Tempest (rtol 1e-5)"}
            except Exception as e:                               # an extra, never the measurement
                line["tdem"]["parity"] = {"error": repr(e)}
            # the device sampler on time-domain data (gbp_rj_run_td): SkyTEM low moment (22 spline nodes, 19 gates), 3-layer synthetic
            # soundings of tests/test_tdem_sampler.py, blocks of 8 192 and 1 024 chains
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from test_tdem_sampler import OFFSET, _survey
                from geobipy_amd.tdem import TdemDeviceChains
                smp = {}
                for Bs in (8192, 1024):
                    s_td, h_td, d_td, _, o_td, _ = _survey(Bs, seed=2)
                    dct = TdemDeviceChains(s_td, h_td, d_td, OFFSET, seed=1, device=device, **o_td)
                    dct.run(100); torch.cuda.synchronize(device)
                    t0 = time.perf_counter(); dct.run(300); torch.cuda.synchronize(device)
                    smp[str(Bs)] = Bs * 300 / (time.perf_counter() - t0)
                    del dct
                line["tdem"]["sampler"] = dict(smp, unit="chain-iterations/s", note="TdemDeviceChains (gbp_rj_run_td), SkyTEM low moment, "
                                               "22 spline nodes x 19 gates, mean ~4 layers; blocks of 8 192 and 1 024 chains")
            except Exception as e:                               # an extra, never the measurement
                line["tdem"]["sampler"] = {"error": repr(e)}
        if world == 1 and not args.no_extras:
            try:
                line["survey"] = survey_extra()
            except Exception as e:                               # an extra, never the measurement
                line["survey"] = {"error": repr(e)}
            if not args.no_survey_sizes and "error" not in line["survey"]:
                try:                                             # the north-star sizes end to end, one run each
                    line["survey"]["north_star_sizes"] = {
                        "config5_schedule_8192_x_10000": survey_at_size(8192, 16, 10000),
                        "soundings_65536": survey_at_size(65536, 64, 2000)}
                except Exception as e:
                    line["survey"]["north_star_sizes"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            # rank 0 times the C oracle on the host cores at EVERY N (the other ranks wait at the barrier in front of
            # destroy_process_group); rank 0's shard starts at sounding 0, so the first `sample` rows of the whole-job arrays are its own
            threads = usable_cores()
            sample = min(Bloc, args.cpu_sample or max(256, min(Btot, 16 * threads)))
            sigma0 = sig_sets[0]
            rate, dt, (p_ref, c_ref, l_ref) = cpu_baseline(system, nl, sigma0, thk, height, obs, sample, threads)
            if dt < 5.0 and not args.cpu_sample:   # scale the sample to ~15 s of CPU work
                sample = int(min(Bloc, max(sample, rate * 15.0)))
                rate, dt, (p_ref, c_ref, l_ref) = cpu_baseline(system, nl, sigma0, thk, height, obs, sample, threads)
            rounds = 1
            while dt < 10.0 and rounds < N_SIGMA_SETS and not args.cpu_sample:   # more proposal rounds of the same sample
                r2, d2, _ = cpu_baseline(system, nl, sig_sets[rounds], thk, height, obs, sample, threads)
                rate = sample * (rounds + 1) / (sample * rounds / rate + d2)
                dt += d2
                rounds += 1
            # the same sample doubles as a parity spot-check of the benchmarked kernel
            chi2, logl = batches[0].forward_loglike(want_pred=True)
            torch.cuda.synchronize(device)
            p = batches[0].predicted[:sample].cpu().numpy()
            r1, d1, _ = cpu_baseline(system, nl, sigma0, thk, height, obs, min(Bloc, 2048), 1)   # single thread beside all cores
            line["cpu_baseline"] = {"value": rate, "unit": "evals/s", "cores": threads, "kind": "port",
                                    "single_thread": {"value": r1, "unit": "evals/s", "sample": f"first {min(Bloc, 2048)} soundings, {d1:.1f} s"},
                                    "sample": f"first {sample} soundings of the same batch x {rounds} proposal round(s), C oracle "
                                              f"(oracle/fdem1d_oracle.c, gcc -O2, OpenMP {threads} threads), {dt:.1f} s"}
            c_gpu, l_gpu = chi2[:sample].cpu().numpy(), logl[:sample].cpu().numpy()
            ok_p = bool(np.all(np.abs(p - p_ref) <= 1e-7 + 1e-9 * np.abs(p_ref)))
            ok_c = bool(np.all(np.abs(c_gpu - c_ref) <= 1e-6 + 1e-9 * np.abs(c_ref)))
            ok_l = bool(np.all(np.abs(l_gpu - l_ref) <= 1e-6 + 1e-9 * np.abs(l_ref)))
            line["parity_vs_cpu"] = {"within_bar": ok_p and ok_c and ok_l,
                                     "bar": "every value: |gpu - cpu| <= 1e-7 ppm + 1e-9 |cpu| (predictions), 1e-6 + 1e-9 |cpu| (chi^2, logL) "
                                            "-- the bars of tests/test_gpu_parity.py; evaluated per element, so a large chi^2 passes through its relative term",
                                     "pred_within_bar": ok_p, "chi2_within_bar": ok_c, "logL_within_bar": ok_l,
                                     "max_abs_pred_ppm": float(np.max(np.abs(p - p_ref))),
                                     "max_abs_chi2": float(np.max(np.abs(c_gpu - c_ref))),
                                     "max_rel_chi2": float(np.max(np.abs(c_gpu - c_ref) / np.maximum(np.abs(c_ref), 1e-300))),
                                     "max_abs_logL": float(np.max(np.abs(l_gpu - l_ref))),
                                     "worst_chi2_excess_over_bar": float(np.max(np.abs(c_gpu - c_ref) - (1e-6 + 1e-9 * np.abs(c_ref)))),
                                     "soundings_compared": int(sample)}
            line["cpu_baseline"]["parity_within_bar"] = line["parity_vs_cpu"]["within_bar"]      # (visible in the driver's parsed line)
        if TRAFFIC_SOURCES:
            line["traffic_sources"] = dict(TRAFFIC_SOURCES, note="every 'traffic' figure is read from the committed rocprofv3 PMC summary named "
                                           "here (separate --pmc passes, gfx950 x2 fetch correction), not measured by this run")
        # the other measured kernels in one compact object inside `roofline` (the driver's record keeps `roofline` whole)
        others = {}
        if "rjmcmc" in line and "value" in line["rjmcmc"]:
            rj = line["rjmcmc"]
            others["rjmcmc_8192_chains"] = {"value": rj["value"], "unit": "chain-iterations/s", "frac": rj["roofline"]["frac"],
                                            "hbm_bytes_per_iteration": rj["roofline"]["traffic"]}
            others["rjmcmc_block_of_1024"] = {"value": max(rj["block_of_1024"]["lockstep"], rj["block_of_1024"]["persistent"]), "unit": "chain-iterations/s"}
        for key in ("jacobian", "tdem", "config2", "shard_8192"):
            if key in line and "roofline" in line[key]:
                others[key] = {"value": line[key]["value"], "unit": line[key]["unit"], "frac": line[key]["roofline"]["frac"]}
        if others:
            line["roofline"]["other_kernels"] = others
        # the full record goes to a side file; the printed line is the compact one the driver's record can hold (<= 3 KB)
        wrote = bool(args.extras_file) and write_extras(line, args.extras_file)
        print(json.dumps(compact_line(line, args.extras_file if wrote else None), separators=(",", ":")), flush=True)
        parity_failed = "parity_vs_cpu" in line and not line["parity_vs_cpu"]["within_bar"]
    else:
        parity_failed = False
    if exchange:
        dist.barrier()              # rank 0 may still be timing the CPU baseline: nobody tears the group down before it is done
        dist.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: the benchmarked kernel is outside the parity bar against the CPU oracle (parity_vs_cpu in the line above)")


if __name__ == "__main__":
    main()
