"""Which persistent kernel for 256 chain pairs (one block per CU, no room for teams)?  us per iteration by engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for C in (512, 384):
    for N in (10000, 30000, 100000, 300000, 1000000):
        ts, te, _ = synth.make_lineages(N, 128, 20, 0)
        row = []
        for name in ("spec", "persistent2", "persistent4", "auto"):
            try:
                eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=40, engine=name)
                eng.init(); eng.steps(200); torch.cuda.synchronize()
                n = 2000 if N <= 100000 else 400
                ms = eng.timed_steps(n)
                row.append("%s(%d) %.2f" % (name, eng.layout.persistent, ms / n * 1e3))
                eng.close()
            except Exception as e:
                row.append("%s: %s" % (name, str(e)[:40]))
        print("C=%d N=%7d: %s" % (C, N, "  ".join(row)), flush=True)
