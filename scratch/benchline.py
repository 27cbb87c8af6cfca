import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%.4e evals/s  %.2f us/step  scan kernel %.2f us frac %.3f lds_frac k/e %.2f/%.2f  engine %.2f us/step" % (
    d["value"], d["ms_per_step"] * 1e3, r["kernel_ms"] * 1e3, r["frac"], r["lds_frac_kernel"], r["lds_frac_engine"],
    r["engine"]["device_ms_per_step_hip_events"] * 1e3))
