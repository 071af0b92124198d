"""profiles/<round>/summary_<case>.json from the rocprofv3 CSVs of profiles/run_profile_r3.sh (gpurun_out/prof/<case>):
per kernel -- launches, average duration, VALU instructions per wave, VALU issue utilisation, HBM bytes per launch (FETCH_SIZE with
and without the gfx950 x2 correction of MI355X_MICROARCH.md, WRITE_SIZE) -- and per-case derived figures bench.py reads back
(hbm_bytes_per_launch / hbm_bytes_per_iteration).   python profiles/summarise_case.py <case> [round]"""
import collections, csv, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
case, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r4")
src = os.path.join(R, "gpurun_out", "prof", case)
log = open(os.path.join(src, "kt.log")).read()
line = next(l for l in log.splitlines() if l.startswith("CASE"))


def short(name):
    m = re.match(r"(void )?((rj::)?k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(2) if m else None


def find(d, suffix):
    for root, _, files in os.walk(os.path.join(src, d)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError(d + " " + suffix)


trace = sorted(csv.DictReader(open(find("kt", "kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
for r in trace:
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_vgpr"] = [float(r.get("VGPR_Count", 0) or 0)]
            acc[k]["_scratch"] = [float(r.get("Scratch_Size", 0) or 0)]
            acc[k]["_lds"] = [float(r.get("LDS_Block_Size", 0) or 0)]
    return acc


# the timed launches of a case are the LAST n of the run ("launches 50 ms warm-up + n" in the CASE line): the launches before them
# bring the clocks up (and build the case) and read 15 - 17 % slow -- they are kept out of ``avg_us_timed``, which is what the
# bench line's ms_per_step is comparable with
m_timed = re.search(r"warm-up \+ (\d+)", line)
n_timed = int(m_timed.group(1)) if m_timed else None
sq, fe, wr = counters("pmc_sq"), counters("pmc_fetch"), counters("pmc_write")
total_ns = sum(sum(v) for v in dur.values())
out = {"case": case, "command": "python scripts/prof_case.py %s under rocprofv3 (profiles/run_profile_%s.sh)" % (case, rnd), "run": line, "kernels": {}}
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    n = len(dur[k])
    d = {"launches": n, "avg_us": sum(dur[k]) / n / 1e3, "share_of_kernel_time": sum(dur[k]) / total_ns}
    if n_timed and n >= n_timed:
        last = dur[k][-n_timed:]
        d["avg_us_timed"] = sum(last) / len(last) / 1e3
        d["timed_launches"] = len(last)
    s = sq.get(k)
    if s and "SQ_WAVES" in s and sum(s["SQ_WAVES"]) > 0:
        tot = lambda c: sum(s[c]) if c in s else 0.0
        d.update(waves_per_launch=tot("SQ_WAVES") / len(s["SQ_WAVES"]), valu_insts_per_wave=tot("SQ_INSTS_VALU") / tot("SQ_WAVES"),
                 lds_insts_per_wave=tot("SQ_INSTS_LDS") / tot("SQ_WAVES"),
                 # a wave64 fp64 VALU instruction occupies its SIMD for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
                 valu_issue_utilisation=tot("SQ_INSTS_VALU") * 4.0 / (1024.0 * tot("GRBM_GUI_ACTIVE") / 8.0) if tot("GRBM_GUI_ACTIVE") else None,
                 vgpr_granules=s["_vgpr"][0], scratch_bytes=s["_scratch"][0], lds_block_bytes=s["_lds"][0])
    if k in fe and "FETCH_SIZE" in fe[k]:
        v = fe[k]["FETCH_SIZE"]
        d["hbm_fetch_bytes_per_launch_raw"] = sum(v) * 1024 / len(v)
        d["hbm_fetch_bytes_per_launch_x2_gfx950"] = 2.0 * sum(v) * 1024 / len(v)
    if k in wr and "WRITE_SIZE" in wr[k]:
        v = wr[k]["WRITE_SIZE"]
        d["hbm_write_bytes_per_launch"] = sum(v) * 1024 / len(v)
    out["kernels"][k] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items()}
per_launch = lambda k: out["kernels"].get(k, {}).get("hbm_fetch_bytes_per_launch_x2_gfx950", 0.0) + out["kernels"].get(k, {}).get("hbm_write_bytes_per_launch", 0.0)
der = {}
if case == "tdem_config4":
    der["hbm_bytes_per_launch"] = sum(per_launch(k) for k in out["kernels"] if "k_fdem_forward" in k or "k_td_apply" in k)
    der["kernel_us_per_forward"] = sum(v.get("avg_us_timed", v["avg_us"]) for k, v in out["kernels"].items() if "k_fdem_forward" in k or "k_td_apply" in k)
if case.startswith("rjmcmc"):
    m = re.search(r"iterations (\d+)\+(\d+)", line)
    n_it = int(m.group(1)) + int(m.group(2))
    # bytes / time per lock-step iteration: launches of the sampler kernels only, summed and divided by the iterations of the run
    der["iterations_in_run"] = n_it
    der["hbm_bytes_per_iteration"] = sum(per_launch(k) * v["launches"] for k, v in out["kernels"].items() if "rj" in k or "fdem" in k) / n_it
    der["kernel_us_per_iteration"] = sum(v["avg_us"] * v["launches"] for k, v in out["kernels"].items() if "rj" in k or "fdem" in k) / n_it
    # aggregate VALU issue utilisation of an iteration: wave-level VALU instructions of all sampler launches of one
    # iteration (counted under the profiler) x 4 cycles / the issue slots of 1 024 SIMDs over the UN-PROFILED iteration time of the same
    # case on the same box (plain.log, run_profile_r6.sh) at the 2.4 GHz peak engine clock
    valu = sum(v.get("valu_insts_per_wave", 0.0) * v.get("waves_per_launch", 0.0) * v["launches"] for k, v in out["kernels"].items() if "rj" in k or "fdem" in k) / n_it
    der["valu_instructions_per_iteration"] = valu
    plain = os.path.join(src, "plain.log")
    if os.path.exists(plain):
        pl = next((l for l in open(plain).read().splitlines() if l.startswith("CASE")), None)
        if pl:
            ms = float(re.search(r"([0-9.]+) ms per iteration", pl).group(1))
            der["unprofiled_run"] = pl
            der["unprofiled_us_per_iteration"] = 1e3 * ms
            der["issue_slots_per_iteration_2p4GHz"] = 1024.0 * ms * 1e-3 * 2.4e9 / 4.0
            der["valu_issue_utilisation_aggregate"] = valu / der["issue_slots_per_iteration_2p4GHz"]
out["derived"] = der
os.makedirs(os.path.join(R, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(R, "profiles", rnd, "summary_%s.json" % case), "w"), indent=1)
print(json.dumps(out, indent=1))
