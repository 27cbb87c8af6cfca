"""Launch-based engine on 16 chains x 1e7 / 3e6 lineages: us per iteration (quick A/B of the chain-step kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts0, te0, _ = synth.make_lineages(100_000, 128, 20, 0)
for N in (10_000_000, 3_000_000):
    ts, te = np.tile(ts0, N // len(ts0)), np.tile(te0, N // len(ts0))
    eng = ChainEngine(ts, te, 16, model=0, seed=1, s_freq=100, n_trace_slots=8, engine="launch")
    eng.init(); eng.steps(200); torch.cuda.synchronize()
    v = sorted(eng.timed_steps(100) / 100 * 1e3 for _ in range(5))
    print("N=%.0e: %.2f us/iter (min %.2f)  scan alone %.2f" % (N, v[2], v[0], eng.time_scan(20) * 1e3), flush=True)
    eng.close()
