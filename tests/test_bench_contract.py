"""The committed bench line (profiles/r01_bench_cfg4_1gpu.json = `python bench.py` on an MI355X) carries every field the
driver's contract names, and the numbers in it are mutually consistent."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_cfg4_1gpu.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["scaling"] == "weak" and b["vs_baseline"] is None and b["dtype"] == "f64"
    assert b["higher_is_better"] is True and b["data"] == "synthetic" and "workload" in b["config"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9)
    # value = iterations x lineages x chains / time
    cfg = b["config"]
    assert b["value"] == pytest.approx(b["steps"] * cfg["lineages"] * cfg["chains_total"] / (b["ms_per_step"] * 1e-3 * b["steps"]))
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["unit"] == b["unit"]
    # the traffic file the bench reads names the kernel the bench line reports
    t = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
    assert t["kernel"] == r["kernel"] and t["workload"] == "cfg4"
