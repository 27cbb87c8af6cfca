"""Few chains x very long input: speculative team kernel vs the tiled launch-based engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for N in (1000000, 3000000, 10000000):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    for C in (4, 16, 64):
        row = []
        for name in ("spec", "launch", "auto"):
            try:
                eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=10, engine=name)
                eng.init(); eng.steps(50); torch.cuda.synchronize()
                n = 300
                ms = eng.timed_steps(n)
                row.append("%s(%d,k=%d) %.1f" % (name, eng.layout.persistent, eng.layout.team_blocks, ms / n * 1e3))
                eng.close()
            except Exception as e:
                row.append("%s: %s" % (name, str(e)[:50]))
        print("N=%8d C=%3d: %s" % (N, C, "  ".join(row)), flush=True)
