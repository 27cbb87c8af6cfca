"""Iteration time of the speculative kernel over (lineages, team size): the data behind lr_spec_model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
general = len(sys.argv) > 1 and sys.argv[1] == "general"
C = 32
for N in (1000, 3000, 10000, 20000, 30000, 50000, 100000, 200000):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    if general:
        rng = np.random.default_rng(5)
        ts = ts + rng.uniform(0, 1, len(ts)) * 0.999
        te = np.maximum(te + rng.uniform(-0.49, 0.49, len(te)), ts + 1e-3)
    row = []
    for k in (1, 2, 4, 8):
        eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=80, engine="spec", team=k)
        eng.init(); eng.steps(300); torch.cuda.synchronize()
        ms = eng.timed_steps(3000)
        row.append("k=%d %.2f" % (eng.layout.team_blocks, ms / 3000 * 1e3))
        eng.close()
    print("N=%6d groups~%5d %s: %s" % (N, (N + 13) // 14, "general" if general else "unit", "  ".join(row)), flush=True)
