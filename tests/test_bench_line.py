"""bench.py prints ONE line the driver's record can hold (a 22.8 KB line left BENCH_r05.parsed null): the compact line
is built from the run's full record by bench.compact_line; here from the committed full record of an earlier run."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_record():
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]", "bench_extras.json"))
                   + glob.glob(os.path.join(ROOT, "profiles", "r5", "bench_line_builder_run.json")))
    assert found, "no committed full bench record"
    return json.load(open(found[-1]))


def test_compact_line_is_small_and_complete():
    full = _full_record()
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.COMPACT_LIMIT < 4096, len(text)
    assert len(json.dumps(line)) < 4096                       # with the default separators too
    assert json.loads(text) == line                            # strict JSON: no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_cpu", "extras_file"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    for k in ("soundings", "frequencies", "layers", "rounds_per_step", "abscissa_points_per_sounding_mean"):
        assert isinstance(line["config"][k], (int, float)), k
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "valu_issue_utilisation",
              "frac_all_abscissae_equivalent", "all_abscissae", "other_kernels"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert set(rf["all_abscissae"]) == {"value", "frac"}
    for k in ("rjmcmc_8192", "rjmcmc_1024", "jacobian", "tdem", "config2"):
        assert "value" in rf["other_kernels"][k], k
    assert rf["other_kernels"]["rjmcmc_1024"]["x8_projection"] == 8 * rf["other_kernels"]["rjmcmc_1024"]["value"] or \
        abs(rf["other_kernels"]["rjmcmc_1024"]["x8_projection"] / (8 * rf["other_kernels"]["rjmcmc_1024"]["value"]) - 1) < 1e-4
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread", "parity_within_bar"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and "value" in cb["single_thread"]
    assert line["parity_vs_cpu"]["within_bar"] is True
    assert abs(line["value"] / full["value"] - 1) < 1e-5      # rounded to 6 digits, not changed


def test_compact_line_at_n_gt_1_keeps_the_contract():
    """A multi-GPU run measures no extras: the line still carries roofline and cpu_baseline."""
    full = _full_record()
    for k in ("rjmcmc", "jacobian", "tdem", "config2", "shard_8192", "shard_16384", "shard_32768", "survey", "abscissa_window"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["roofline"].pop("all_abscissae", None)
    full["roofline"].pop("other_kernels", None)
    line = bench.compact_line(full)
    assert "cpu_baseline" in line and "roofline" in line and line["roofline"]["frac"] > 0
    assert len(json.dumps(line)) < 2048


def test_overlong_input_never_prints_an_overlong_line():
    full = _full_record()
    full["config"]["workload"] = "x" * 2600
    line = bench.compact_line(full)
    assert len(json.dumps(line, separators=(",", ":"))) <= 4096
    assert "roofline" in line and "cpu_baseline" in line


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_prints_one_parsable_line_on_the_gpu(tmp_path):
    """python bench.py (bounded flags) on cuda:0: the LAST stdout line is the compact JSON line -- under 3 KB, the contract's keys, roofline and
    cpu_baseline objects, parity within the bar -- and the full record lands in the extras file."""
    import subprocess
    extras = os.path.join(str(tmp_path), "extras.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--soundings", "8192",
                        "--no-extras", "--no-rjmcmc", "--cpu-sample", "512", "--extras-file", extras],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) <= bench.COMPACT_LIMIT
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity_vs_cpu", "extras_file"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 2 and line["dtype"] == "f64" and line["value"] > 1e6
    assert line["roofline"]["frac"] > 0.05 and line["roofline"]["kernel"].startswith("k_fdem_forward")
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    assert line["parity_vs_cpu"]["within_bar"] is True
    full = json.load(open(extras))
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and "definition" in full["roofline"]
