"""The committed bench output (profiles/r05_bench_cfg4_1gpu.json = the DETAIL of `python bench.py` on an MI355X,
profiles/r05_bench_compact_lines.json = the final stdout lines the driver parses) carries every field the driver's contract
names, and the numbers are mutually consistent and agree with the committed rocprofv3 summaries (engine kernel:
r05_launch_durations.json / r05_pmc_*.json; the HBM-streaming entry points: r05_abi_kernel_stats.json)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_cfg4_1gpu.json")))
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
    # a team per CHAIN scores one chain per gather (8-byte reads of the one-chain scan): its LDS bytes per eval follow that
    for name in ("cfg2", "cfg3", "cfg5", "cfg4_shard128"):
        s = b["configs"][name]
        assert s["chains_per_gather"] == 1 and s["gathers_per_eval"] == pytest.approx(8 / 14) and "lr_spec_kernel" in s["kernel"]
        assert s["lds_bytes_per_eval"] == pytest.approx(8 * 16 / 14.0)
    assert b["configs"]["cfg4_general"]["chains_per_gather"] == 2
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
    # the issue-rate view follows the per-class costs of profiles/r04_ubench.txt
    i = r["issue"]
    assert i["simd_cycles_per_trip"] == pytest.approx(i["half_rate_instr_per_trip"] * 4.2 + i["full_rate_instr_per_trip"] * 2.35)
    assert i["frac"] == pytest.approx(r["kernel_evals_per_s"] / i["peak_evals_per_s"]) and 0 < i["frac"] < 1


@pytest.mark.parametrize("detail", ["r05_bench_cfg4_1gpu.json", "r05_bench_cfg4_1gpu_driver_args.json", "r04_bench_cfg4_1gpu.json",
                                    "r04_bench_cfg4_1gpu_driver_args.json", "r03_bench_cfg4_1gpu.json"])
def test_final_stdout_line_is_compact_and_survives_the_drivers_tail(detail, tmp_path, monkeypatch):
    """Round 4's line had grown to 28 KB and the driver, which keeps the last 8 KB of stdout, could not parse it.  bench.py
    now writes the detail to bench_detail.json (and as `#detail` lines to STDERR) and prints ONE compact line built by
    compact_line() - the only stdout line: under 4 KB, scalars only inside `config` / `roofline` / `cpu_baseline` (the
    driver's parser drops nested objects there).  Even with the detail lines on the same stream (as here) it is recoverable
    from the last 8081 characters - the size of the driver's tail."""
    import io
    import sys
    sys.path.insert(0, ROOT)
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", detail)))
    monkeypatch.setenv("LR_BENCH_DETAIL", str(tmp_path / "bench_detail.json"))
    out, err = io.StringIO(), io.StringIO()
    assert bench.emit(dict(d), out, err) == out.getvalue().rstrip("\n") and out.getvalue().count("\n") == 1       # stdout: the line alone
    assert err.getvalue().startswith("#detail ") and all(l.startswith("#detail ") for l in err.getvalue().splitlines())
    buf = io.StringIO()
    line = bench.emit(d, buf)
    assert len(line) < bench.COMPACT_LIMIT < 8192
    assert json.load(open(tmp_path / "bench_detail.json"))["value"] == d["value"]        # nothing is lost: the detail file
    tail = buf.getvalue()[-8081:]
    last = tail.rstrip("\n").splitlines()[-1]
    assert last == line and sum(l.startswith("{") for l in buf.getvalue().splitlines()) == 1
    b = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["value"] == pytest.approx(d["value"]) and b["ms_per_step"] == pytest.approx(d["ms_per_step"])
    assert b["steps"] == d["steps"] and b["warmup"] == d["warmup"] and b["n_gpus"] == d["n_gpus"]
    for sec in ("config", "roofline", "cpu_baseline"):
        assert all(not isinstance(v, (dict, list)) for v in b[sec].values()), sec
        assert all(len(v) <= 120 for v in b[sec].values() if isinstance(v, str)), sec     # the driver cuts strings at 120
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "iterations_per_launch"):
        assert k in r, k
    assert r["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-4) and r["kernel"] == d["roofline"]["kernel"]
    assert r["traffic"] == pytest.approx(d["roofline"]["traffic"], rel=1e-4)
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] == pytest.approx(d["cpu_baseline"]["value"], rel=1e-4)
    assert set(b["configs"]) >= {"cfg2", "cfg3", "cfg5", "cfg4_general"} and b["co_headline"]["value"] > 0
    if "abi" in d:
        a = b["abi"]
        assert a["lr_bin_unit_events"]["hbm_frac"]["3e+07"] > 0.4 and a["lr_bd_loglik_batch_c1"]["hbm_frac"]["3e+07"] > 0.4
        assert len(a["engine_streaming"]) in (2, 3) and a["seam"]["us_per_call_1_state"] > a["seam"]["numpy_us_per_call"]
    if detail.startswith("r05"):
        # ... and it IS what the run printed last (profiles/r05_bench_compact_lines.json: the lines as captured)
        printed = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_compact_lines.json")))["driver_args" if "driver" in detail else "default"]
        assert {k: v for k, v in printed.items() if k != "detail"} == {k: v for k, v in b.items() if k != "detail"}
        assert len(json.dumps(printed, separators=(",", ":"))) < bench.COMPACT_LIMIT
        a = b["abi"]
        assert a["lr_bin_unit_events"]["hbm_frac"]["1e+08"] >= 0.55 and a["lr_bd_loglik_batch_c1"]["hbm_frac"]["1e+08"] >= 0.55
        assert a["lr_bin_unit_events"]["traffic_over_algorithmic"] == pytest.approx(1.0, abs=0.01)


def test_fetch_summary_counts_the_librarys_own_kernels_only():
    """Round 4 reported traffic = 7/6 of the algorithmic bytes for BOTH streaming kernels: the child's FETCH_SIZE rows were
    summed over every kernel whose name CONTAINED "lr_" - and __amd_rocclr_copyBuffer (torch's copy while the input is
    made, half a pass at FETCH_SIZE's gfx950 scale) does.  abi_fetch_summary takes names that START with lr_, lists every
    dispatch, and calibrates the counter on the read-only yardstick (16 n known bytes).  Rows as the 1e8-lineage child of
    round 5 printed them."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    n = 100_000_000
    rows = [{"Kernel_Name": "__amd_rocclr_copyBuffer", "Counter_Value": "390660.5", "Dispatch_Id": "3"},
            {"Kernel_Name": "__amd_rocclr_fillBufferAligned", "Counter_Value": "14.5", "Dispatch_Id": "4"}]
    rows += [{"Kernel_Name": "lr_build_tables_kernel(double const*)", "Counter_Value": "5.3", "Dispatch_Id": str(10 + 3 * i)} for i in range(3)]
    rows += [{"Kernel_Name": "void lr_scan_fast_kernel<1, 136>(double const*, double const*)", "Counter_Value": v, "Dispatch_Id": str(11 + 3 * i)}
             for i, v in enumerate(("781328.0", "781295.7", "781295.25"))]
    rows += [{"Kernel_Name": "lr_debug_stream2_kernel(HIP_vector_type<double, 2u> const*)", "Counter_Value": v, "Dispatch_Id": str(30 + i)}
             for i, v in enumerate(("781265.2", "781263.1", "781262.75"))]
    traffic, note, d = bench.abi_fetch_summary(rows, n)
    assert traffic / (16.0 * n) == pytest.approx(1.0, abs=2e-3)                 # x 2 (gfx950): one pass, no over-fetch
    assert d["yardstick_raw_bytes_over_algorithmic"] == pytest.approx(0.5, abs=1e-4)      # the guide's factor, measured
    assert d["traffic_over_algorithmic_calibrated"] == pytest.approx(1.0, abs=2e-3)
    assert d["dispatches"]["void lr_scan_fast_kernel<1, 136>"]["count"] == 3 and "__amd_rocclr_copyBuffer" in d["dispatches"]
    # what round 4's filter would have said
    old = sum(float(x["Counter_Value"]) for x in rows if "lr_" in x["Kernel_Name"] and "stream2" not in x["Kernel_Name"]) * 2 * 1024 / 3
    assert old / (16.0 * n) == pytest.approx(7 / 6, abs=2e-3)


def test_abi_section_prices_the_hbm_streaming_entry_points():
    """`abi`: lr_bin_unit_events and lr_bd_loglik_batch at 1e7 / 3e7 / 1e8 lineages against the 8 TB/s HBM peak - the kernels
    for which HBM IS the bound (16 B per lineage and pass) - timed on rotating copies of the input (no call finds
    its input in the Infinity Cache), beside the read-only yardstick, with the FETCH_SIZE traffic of the same call, and the
    cost of the calc_likelihood seam."""
    b = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_cfg4_1gpu.json")))
    a = b["abi"]
    assert a["peak_GBs"] == 8000.0 and a["bytes_per_lineage_pass"] == 16 and a["rotate_bytes"] >= 1e9
    sizes = {r["lineages"] for r in a["rows"]}
    assert sizes == {10_000_000, 30_000_000, 100_000_000}
    for r in a["rows"]:
        assert r["achieved_GBs"] == pytest.approx(16.0 * r["lineages"] * r["passes"] / (r["ms"] * 1e-3) / 1e9)
        assert r["hbm_frac"] == pytest.approx(r["achieved_GBs"] / 8000.0)
        assert r["rotated_copies"] * 16.0 * r["lineages"] >= 1.2e9          # >= 1.2 GB streamed between two uses of a copy
        assert r["frac_of_stream2"] == pytest.approx(r["achieved_GBs"] / a["stream2_GBs"][str(r["lineages"])])
        if r["kernel"] == "lr_bd_loglik_batch":
            assert r["passes"] == -(-r["chains"] // r["Cb"]) and r["Cb"] in (1, 2, 4, 8, 16)
            assert r["Cb"] == (16 if r["chains"] > 8 else r["chains"])      # sixteen chains per pass from nine chains on
    # the read-only yardstick (what a pass over ts / te can reach on the box) sits between the kernels and the nominal peak
    for n, y in a["stream2_GBs"].items():
        assert 5000 < y < 7000, (n, y)
    for k in ("lr_bin_unit_events", "lr_bd_loglik_batch"):
        h = a[k]
        assert h["lineages"] == 100_000_000                        # the summary figures are taken past the Infinity Cache
        assert h["hbm_frac"] >= 0.55, (k, h)                       # (the north star asks for 0.40)
        assert 0.85 <= h["frac_of_stream2"] <= 1.0
        # no wasted re-reads: one pass = the algorithmic bytes, with the guide's x 2 AND calibrated on the yardstick
        assert h["traffic_over_algorithmic"] == pytest.approx(1.0, abs=0.01)
        assert h["traffic_over_algorithmic_calibrated"] == pytest.approx(1.0, abs=0.01)
        assert h["fetch_detail"]["yardstick_raw_bytes_over_algorithmic"] == pytest.approx(0.5, abs=0.005)
    # every sorted-input row of ONE pass with at most eight chains is above the north star's bar at every size; the
    # sixteen-chain pass is bound by the LDS bank conflicts of its death-side gathers and stays below it
    one_pass = [r for r in a["rows"] if r["kernel"] == "lr_bd_loglik_batch" and r["passes"] == 1 and r["order"] == "sorted"]
    assert min(r["hbm_frac"] for r in one_pass if r["chains"] <= 8) >= 0.40
    wide = [r for r in one_pass if r["chains"] == 16 and r["lineages"] == 100_000_000 and not r["general_times"]][0]
    eight = [r for r in a["rows"] if r["kernel"] == "lr_bd_loglik_batch" and r["chains"] == 8 and r["lineages"] == 100_000_000 and not r["general_times"]][0]
    assert 0.30 <= wide["hbm_frac"] < 0.50 and wide["ms"] < 2 * eight["ms"]          # ... but beats two passes of eight
    # the kernel trace of the same child command agrees with the HIP-event time of the call (three cold calls against
    # back-to-back warm ones: within 15 %)
    st = json.load(open(os.path.join(ROOT, "profiles", "r05_abi_kernel_stats.json")))["rows"]
    for run, kern, rows_key in (("abi_lr_bd_loglik_batch_c1_n100000000", "lr_scan_fast_kernel<1, 136>", dict(kernel="lr_bd_loglik_batch", chains=1)),
                                ("abi_lr_bd_loglik_batch_c16_n100000000", "lr_scan_wide_kernel<136>", dict(kernel="lr_bd_loglik_batch", chains=16)),
                                ("abi_lr_bin_unit_events_c0_n100000000", "lr_bin_unit_kernel", dict(kernel="lr_bin_unit_events", chains=0))):
        k_ns = [x["avg_ns"] for x in st if x["run"] == run and kern in x["kernel"]][0]
        call_ms = [r["ms"] for r in a["rows"] if r["lineages"] == 100_000_000 and not r["general_times"] and r["order"] == "sorted"
                   and all(r[k] == v for k, v in rows_key.items())][0]
        assert k_ns * 1e-6 == pytest.approx(call_ms, rel=0.15)
        y_ns = [x["avg_ns"] for x in st if x["run"] == run and "lr_debug_stream2_kernel" in x["kernel"]][0]
        assert y_ns <= k_ns                                        # the yardstick of the same child: never slower than the kernel
    # the RJMCMC loop itself on few chains x 1e7 / 3e7 / 1e8 lineages: what the planner runs (the launch-based engine scanning
    # the PACKED lineages: LDS-bound) and, beside it, the form that re-reads ts / te in every iteration (HBM-bound): the WHOLE
    # iteration against the LDS / HBM peak, and the scan kernel alone
    es = a["engine_streaming"]
    assert [r["lineages"] for r in es] == [10_000_000, 30_000_000, 100_000_000] and all(r["persistent"] == 0 and r["passes"] == 1 for r in es)
    for r in es:
        assert r["packed_scan"] == 1 and "lr_packscan_kernel" in r["kernel"] and r["packed_bytes_per_lineage"] == pytest.approx(16 / 14)
        assert r["evals_per_s"] == pytest.approx(r["lineages"] * r["chains"] / (r["us_per_iter"] * 1e-6)) and r["evals_per_s"] > 5e12
        assert r["lds_frac"] == pytest.approx(r["evals_per_s"] * r["lds_bytes_per_eval"] / 1e9 / (256 * 256 * 2.4), rel=1e-6)
        # (the iteration may beat the ONE launch over all chains that `scan_kernel_us` times: long scans run in two partitions)
        assert 0.2 <= r["lds_frac"] <= 1.0 and 0.2 <= r["scan_lds_frac"] <= 1.0
        t = r["ts_te"]
        assert t["packed_scan"] == 0 and "lr_scan_unit_kernel" in t["kernel"] and t["passes"] == 1
        assert t["hbm_GBs"] == pytest.approx(16.0 * r["lineages"] * t["passes"] / (t["us_per_iter"] * 1e-6) / 1e9)
        assert t["hbm_frac"] >= 0.45 and t["scan_hbm_frac"] >= 0.60
        assert r["us_per_iter"] < 0.75 * t["us_per_iter"]                       # the packed scan is what makes the difference
    assert es[1]["ts_te"]["hbm_frac"] >= 0.55 and es[2]["ts_te"]["hbm_frac"] >= 0.65 and es[2]["lds_frac"] >= 0.55
    seam = a["seam"]["BDI_partial_lik"]
    assert seam["us_per_call_1_state"] > seam["numpy_binned_us_per_call"] and seam["states_per_call_to_break_even"] < 64


def test_driver_args_bench_line_measures_the_kernel():
    """profiles/r05_bench_cfg4_1gpu_driver_args.json = `python bench.py --steps 20 --warmup 5` (the driver's command): the
    wall-clock figure stays within 40 % of the event-bracketed device time of the same region (round 2: 66 %; what is left
    is one launch, two event markers and the completion wake-up of a ~140-us region: 1.16-1.36 by box of the pool)."""
    b = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_cfg4_1gpu_driver_args.json")))
    assert b["steps"] == 20 and b["warmup"] == 5 and b["n_gpus"] == 1
    r = b["roofline"]
    assert b["ms_per_step"] * 1e3 <= 1.40 * r["engine"]["us_per_iter_device"]
    assert r["frac_engine"] >= 0.70 * r["frac"]
    assert b["config"]["trace_rows_gathered_in_region"] == 0


def test_profile_summary_agrees_with_the_bench_line():
    """profiles/r05_launch_durations.json: the timed launch in rocprofv3's kernel trace against bench.py's HIP events of
    the same (profiled) run; profiles/r05_pmc_1000it.json names the same kernel; the strong-scaling leg of a two-rank run
    is in round 4's file."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_launch_durations.json")))
    assert d["timed_2000_iteration_launch_ms_kernel_trace"] == pytest.approx(
        d["bench_hip_event_ms_for_the_timed_2000_iteration_launch"], rel=0.01)
    p = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_1000it.json")))
    b = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_cfg4_1gpu.json")))
    assert p["kernel"].split("<")[0] == b["roofline"]["kernel"].split("<")[0] == d["kernel"].split("<")[0]
    assert 0 < p["lds_busy_fraction"] < 1 and 0 < p["valu_busy_fraction"] < 1
