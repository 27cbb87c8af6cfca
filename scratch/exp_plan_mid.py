"""ChainEngine.plan_check between the persistent engines' home ground and the packed launches': 3e5 / 1e6 lineages x 16 ... 1024
chains, unit resolution and continuous times."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for n, chains in ((300_000, (16, 64, 256, 1024)), (1_000_000, (16, 64, 256, 1024))):
    ts0, te0, _ = synth.make_lineages(n, 128, 20, 0)
    for general in (False, True):
        ts, te = ts0, te0
        if general:
            rng = np.random.default_rng(7)
            ts = ts0 + rng.uniform(0.0, 1.0, len(ts0)) * 0.999
            te = np.maximum(np.ceil(te0) - 1.0 + rng.uniform(1e-3, 0.999, len(te0)), ts + 1e-3)
        for C in chains:
            eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=1 << 30, n_trace_slots=2)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r = eng.plan_check(n_iters=200)
            eng.close()
            t = {k: (None if v is None else round(v, 2)) for k, v in r["us_per_iter"].items()}
            print("n=%8d %s C=%5d: %-40s %s best=%s %s" % (n, "general" if general else "unit   ", C, r["auto"][:40], t, r["best"],
                  "" if r["ok"] else "  <-- planner off by %.0f %%" % (100 * (t["auto"] / t[r["best"]] - 1))), flush=True)
