"""profiles/<round>/summary_bench_65536x10x8.json (python profiles/summarise_bench.py <tag> [round]) from the rocprofv3 CSVs of profiles/run_profile.sh <tag> (gpurun_out/prof/<tag>)."""
import csv, json, os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
rnd = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(R, "gpurun_out", "prof", tag)
KERNEL = "k_fdem_forward<true"          # (round 4: <true, false> -- the row-scale template argument)
B, L, F = 65536, 8, 10


def rows(path):
    return [r for r in csv.DictReader(open(path)) if KERNEL in r["Kernel_Name"]]


trace = rows(os.path.join(src, "kt", "kt_kernel_trace.csv"))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace]
pmc = collections.defaultdict(list)
disp = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for r in rows(os.path.join(src, d, "p_counter_collection.csv")):
        pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        disp = {k: int(r[k]) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size", "Scratch_Size")}
m = {k: sum(v) / len(v) for k, v in pmc.items()}
bench = json.loads(next(l for l in open(os.path.join(src, "bench_kt.log")).read().splitlines()[::-1] if l.startswith("{")))
fetch, write = m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
cycles = m["GRBM_GUI_ACTIVE"] / 8.0
timed = dur[-bench["steps"] * bench.get("rounds_per_step", 1):]
cfg = bench.get("config", {})
pts, pts_all = cfg.get("abscissa_points_per_sounding_mean", cfg.get("abscissa_points_per_sounding_at_35_m")), cfg.get("abscissa_points_all", 1200)
label = ("label unknown (bench line without config.abscissa_points_*)" if pts is None else
         "all %d abscissa points per sounding (hankel_eps_ppm = 0)" % pts_all if pts == pts_all else
         "default path: per-sounding abscissa window, eps = %g ppm, %.1f of %d abscissa points per sounding (batch mean)" % (cfg.get("hankel_eps_ppm"), pts, pts_all))
out = {
    "command": "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-windowed --no-rjmcmc --no-extras (under rocprofv3, profiles/run_profile.sh; reduced by profiles/summarise_bench.py)",
    "kernel": KERNEL,
    "workload": "65536 soundings x 10 zz freq x 8 layers, 1 GPU (%s)" % label,
    "kernel_trace": {"calls": len(dur), "avg_ns": sum(dur) / len(dur), "min_ns": min(dur), "max_ns": max(dur),
                     "timed_region_avg_ns": sum(timed) / len(timed),
                     "bench_kernel_ms_same_run_hip_events": bench["roofline"]["kernel_ms"],
                     "note": "warm-up launches run slower (clock ramp); the timed launches of the trace and bench.py's HIP-event "
                             "kernel_ms of the same run agree; un-profiled runs clock higher (MI355X_MICROARCH.md, DVFS note)"},
    "dispatch": disp,
    "pmc_mean_per_launch": m,
    "derived": {
        "hbm_fetch_bytes_raw": fetch, "hbm_write_bytes_raw": write, "hbm_fetch_bytes_x2_gfx950_correction": 2.0 * fetch,
        "note_hbm": "FETCH_SIZE/WRITE_SIZE are in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide "
                    "coalesced streams (other widths uncalibrated) -> both raw and x2 given",
        "algorithmic_bytes_per_launch": B * (8 * (2 * L + 1 + 2 * F + 2) + 4 + 16),
        "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
        "valu_issue_utilisation_fp64_4cyc": m["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles),
        "note_valu": "1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; an fp64 VALU op occupies its SIMD for 4 cycles per "
                     "wave64 (32-bit ops 2 cycles, so this slightly overstates)",
        "effective_clock_GHz_profiled": cycles / (sum(dur) / len(dur)),
    },
}
json.dump(out, open(os.path.join(R, "profiles", rnd, "summary_bench_65536x10x8.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_trace", "dispatch", "derived")}, indent=1))
