"""Device time of one launch of n iterations (cfg4, four-chain kernel) for small n: the per-launch intercept."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=100, n_trace_slots=400)
eng.init(); eng.steps(3000); torch.cuda.synchronize()
for n in (1, 2, 4, 8, 16, 20, 32, 64, 128, 512):
    v = []
    for rep in range(7):
        eng.steps(50); v.append(eng.timed_steps(n) * 1e3)
    v = np.sort(v)
    print("n=%4d: device us median %.2f (min %.2f)  per iteration %.3f" % (n, v[3], v[0], v[3] / n))
