"""profiles/r1/summary_rjmcmc_65536.json from the rocprofv3 CSVs of profiles/run_profile_rjmcmc.sh (gpurun_out/prof/rj)."""
import csv, json, os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof", "rj")
B, N_IT = 65536, 120          # chains; iterations in the run (20 warm-up + 100 timed)


def short(name):
    for key in ("k_fdem_sens", "k_fdem_forward<true>", "k_fdem_forward<false>", "k_rj_accept8", "k_rj_accept", "k_rj_newton8",
                "k_rj_newton", "k_rj_propose_thread", "k_rj_propose_wave"):
        if key in name:
            return key
    return None


def modal_grid(rows):
    """Sampler launches only: per kernel the most common grid size (drops the chain-initialisation launches)."""
    cnt = collections.defaultdict(collections.Counter)
    for r in rows:
        k = short(r["Kernel_Name"])
        if k:
            cnt[k][r.get("Grid_Size", r.get("Grid_Size_X"))] += 1
    return {k: c.most_common(1)[0][0] for k, c in cnt.items()}


def counters(path):
    rows = list(csv.DictReader(open(path)))
    grid = modal_grid(rows)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and r["Grid_Size"] == grid[k]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


sq, fe, wr = (counters(os.path.join(src, d, "p_counter_collection.csv")) for d in ("pmc_sq", "pmc_fetch", "pmc_write"))
dur = collections.defaultdict(list)
_rows = list(csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_trace.csv"))))
_grid = modal_grid(_rows)
for r in _rows:
    k = short(r["Kernel_Name"])
    if k and r["Grid_Size_X"] == _grid[k]:
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"command": "python scripts/bench_rjmcmc_device.py 65536 100 under rocprofv3 (profiles/run_profile_rjmcmc.sh)",
       "workload": "65536 Resolve soundings (6 frequencies), exact Jacobian, k ~ 2.5 layers, 120 lockstep iterations", "kernels": {}}
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    n = len(dur[k])
    d = {"launches": n, "us_per_iteration": sum(dur[k]) / N_IT / 1e3, "avg_us": sum(dur[k]) / n / 1e3}
    s = sq.get(k, {})
    if s:
        tot = lambda c: sum(s[c]) if c in s else None
        waves, valu, act, busy, gui, lds = (tot(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES",
                                                              "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS"))
        d.update(waves_per_iteration=waves / N_IT, valu_instructions_per_wave=valu / waves, lds_instructions_per_wave=lds / waves,
                 # a wave64 VALU instruction occupies its SIMD for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
                 # (same accounting as summary_bench_65536x10x8.json).  Launches in which most workgroups exit at once
                 # (layer-count buckets, chains that do not need the kernel) dilute the per-wave averages.
                 valu_issue_utilisation=valu * 4.0 / (1024.0 * gui / 8.0) if gui else None)
    if k in fe:
        d["hbm_fetch_MB_per_iteration_x2_corrected"] = 2.0 * sum(fe[k]["FETCH_SIZE"]) * 1024 / N_IT / 1e6
    if k in wr:
        d["hbm_write_MB_per_iteration"] = sum(wr[k]["WRITE_SIZE"]) * 1024 / N_IT / 1e6
    out["kernels"][k] = {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d.items()}
out["us_per_iteration_all_kernels"] = round(sum(v["us_per_iteration"] for v in out["kernels"].values()), 1)
import re
try:
    wall = re.search(r"([0-9.]+ ms) per lockstep iteration", open(os.path.join(src, "kt.log")).read()).group(1)
except Exception:
    wall = "see kt.log"
out["note_overlap"] = ("at this block size the Jacobian pass of the dimension-changing proposals runs on a side stream next to the fused "
                       "forward kernel: their durations overlap (each is longer than it would be alone), so the sum over kernels exceeds the "
                       "wall time per iteration printed by the benchmark (%s in the traced run)" % wall)
json.dump(out, open(os.path.join(R, "profiles", (sys.argv[1] if len(sys.argv) > 1 else "r2"), "summary_rjmcmc_65536.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
