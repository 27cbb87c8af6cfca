import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%.4e evals/s  %.2f us/step  kernel %.2f us frac %.3f  scan-only %.2f us" % (
    d["value"], d["ms_per_step"] * 1e3, r["kernel_ms"] * 1e3, r["frac"], r["scan_only_kernel_ms_all_chains"] * 1e3))
