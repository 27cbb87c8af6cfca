import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%.4e evals/s  %.2f us/step  %s %.3f ms/launch (%d it) frac %.3f lds k/e %.2f/%.2f  events %.2f us/step" % (
    d["value"], d["ms_per_step"] * 1e3, r["kernel"], r["kernel_ms"], r["iterations_per_launch"], r["frac"], r["lds_frac_kernel"],
    r["lds_frac_engine"], r["engine"]["device_ms_per_step_hip_events"] * 1e3))
