"""ChainEngine.plan_check where the launch-based engine competes: 3e6 / 1e7 synthetic lineages, unit resolution and continuous
times, 16 ... 256 chains (after the launch-based scans got the scalar birth side, round 5)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts1, te1, _ = synth.make_lineages(100_000, 128, 20, 0)
for n, chains in ((3_000_000, (16, 64, 256)), (10_000_000, (16, 64, 256))):
    reps = n // len(ts1)
    ts0, te0 = np.sort(np.tile(ts1, reps), kind="stable"), None
    order = np.argsort(np.tile(ts1, reps), kind="stable")
    te0 = np.tile(te1, reps)[order]
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
                r = eng.plan_check(n_iters=100)
            eng.close()
            t = {k: (None if v is None else round(v, 2)) for k, v in r["us_per_iter"].items()}
            print("n=%8d %s C=%5d: %-44s %s best=%s %s" % (n, "general" if general else "unit   ", C, r["auto"][:44], t, r["best"],
                  "" if r["ok"] else "  <-- planner off by %.0f %%" % (100 * (t["auto"] / t[r["best"]] - 1))), flush=True)
