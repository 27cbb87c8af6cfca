"""The planner's choice against every engine (ChainEngine.plan_check) on the reference's own datasets and on cfg-like sizes,
for the chain counts a user would pass."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "binning_lik.npz"))
cases = [("example_TBP", G["example_TBP/ts"], G["example_TBP/te"]), ("metal_bands", G["metal_bands/ts"], G["metal_bands/te"])]
for n in (1000, 10000):
    ts, te, _ = synth.make_lineages(n, 128, 20, 0)
    cases.append(("synthetic %d" % n, ts, te))
for name, ts, te in cases:
    for C in (64, 256, 512, 1024, 4096):
        eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=1 << 30, n_trace_slots=2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = eng.plan_check(n_iters=600)
        eng.close()
        t = {k: (None if v is None else round(v, 2)) for k, v in r["us_per_iter"].items()}
        print("%-16s C=%5d: %-42s %s best=%s %s" % (name, C, r["auto"][:42], t, r["best"], "" if r["ok"] else "  <-- planner off by %.0f %%" % (100 * (t["auto"] / t[r["best"]] - 1))), flush=True)
