"""The committed bench line (profiles/r03_bench_cfg4_1gpu.json = `python bench.py` on an MI355X) carries every field the
driver's contract names, and the numbers in it are mutually consistent and agree with the committed rocprofv3 summary."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_cfg4_1gpu.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["scaling"] == "weak" and b["vs_baseline"] is None and b["dtype"] == "f64"
    assert b["higher_is_better"] is True and b["data"] == "synthetic" and "workload" in b["config"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the kernel gathers lookup-table entries from LDS: that pipe is its roofline (256 B/clk/CU x 256 CUs x 2.4 GHz)
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == pytest.approx(256 * 256 * 2.4)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["evals_per_launch"] * r["lds_bytes_per_eval"] / (r["kernel_ms"] * 1e-3) / 1e9)
    # HBM side fields: the survey's 16-B convention, and the bytes that really reach HBM (PMC, measured in the run)
    assert r["traffic"] is not None and r["traffic"] > 0 and "rocprofv3 --pmc" in r["traffic_note"]
    h = r["hbm"]
    assert h["peak_GBs"] == 8000.0 and h["measured_GBs"] == pytest.approx(r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9)
    assert h["measured_frac_of_peak"] < 0.05 < 1.0 < h["algorithmic_GBs_16B_convention"] / h["peak_GBs"]
    # value = iterations x lineages x chains / time
    cfg = b["config"]
    assert b["value"] == pytest.approx(b["steps"] * cfg["lineages"] * cfg["chains_total"] / (b["ms_per_step"] * 1e-3 * b["steps"]))
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["unit"] == b["unit"]
    # every BASELINE configuration carries a number, cfg5 its own CPU leg
    for name in ("cfg2", "cfg3", "cfg5", "cfg4_general", "cfg4_shard128"):
        s = b["configs"][name]
        assert s["evals_per_s"] > 0 and s["us_per_iter"] > 0 and 0 < s["lds_frac"] < 1 and s["kernel"].startswith("lr_")
    assert b["configs"]["cfg5"]["cpu_baseline"]["kind"] == "port"
    # the unit is stated for what it is: gathers and fp64 operations per counted eval, the un-amortised 16-B figure, the
    # aggregation in words; every host core in the CPU leg; the co-headline on continuous times with its own roofline
    assert r["gathers_per_eval"] == pytest.approx(8 / 28) and r["fp64_ops_per_eval"] == pytest.approx(17 / 28)
    assert "pre-summed" in r["aggregation"] and "frozen" in cfg["eval_note"] and "no per-bin event counts" not in cfg["eval_note"]
    assert h["effective_GBs_16B_per_eval"] == pytest.approx(16 * r["kernel_evals_per_s"] / 1e9)
    assert h["frac_of_peak_16B_convention"] == pytest.approx(h["algorithmic_GBs_16B_convention"] / 8000.0)
    assert c["all_cores"]["cores"] >= 16 and "sched_getaffinity" in c["all_cores"]["sample"]
    co = b["co_headline"]
    assert "continuous times" in co["workload"] and co["value"] == pytest.approx(b["configs"]["cfg4_general"]["evals_per_s"])
    assert co["roofline"]["gathers_per_eval"] == pytest.approx(16 / 28) and 0 < co["roofline"]["frac"] < 1
    assert cfg["wall_over_device"] == pytest.approx(b["ms_per_step"] * b["steps"] / r["kernel_ms"], rel=1e-6)


def test_driver_args_bench_line_measures_the_kernel():
    """profiles/r03_bench_cfg4_1gpu_driver_args.json = `python bench.py --steps 20 --warmup 5` (the driver's command): the
    wall-clock figure stays within 35 % of the event-bracketed device time of the same region (round 2: 66 %; what is left
    is one launch, two event markers and the completion wake-up of a ~150-us region: 1.22-1.29 by box of the pool)."""
    b = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_cfg4_1gpu_driver_args.json")))
    assert b["steps"] == 20 and b["warmup"] == 5 and b["n_gpus"] == 1
    r = b["roofline"]
    assert b["ms_per_step"] * 1e3 <= 1.35 * r["engine"]["us_per_iter_device"]
    assert r["frac_engine"] >= 0.74 * r["frac"]
    assert b["config"]["trace_rows_gathered_in_region"] == 0


def test_profile_summary_agrees_with_the_bench_line():
    """profiles/r03_launch_durations.json: the timed launch in rocprofv3's kernel trace against bench.py's HIP events of
    the same (profiled) run; profiles/r03_pmc_1000it.json names the same kernel."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_launch_durations.json")))
    assert d["timed_2000_iteration_launch_ms_kernel_trace"] == pytest.approx(
        d["bench_hip_event_ms_for_the_timed_2000_iteration_launch"], rel=0.01)
    p = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_1000it.json")))
    b = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_cfg4_1gpu.json")))
    assert p["kernel"].split("<")[0] == b["roofline"]["kernel"].split("<")[0] == d["kernel"].split("<")[0]
    assert 0 < p["lds_busy_fraction"] < 1 and 0 < p["valu_busy_fraction"] < 1
