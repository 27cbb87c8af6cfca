"""Two-chain vs four-chain persistent kernel beyond 512 chains: the data behind the last rule of lr_persist_variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for N in (3000, 10000, 30000, 100000, 1000000, 3000000):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    for C in (768, 1024, 1536):
        row = []
        for name in ("persistent2", "persistent4", "auto"):
            eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=40, engine=name)
            eng.init(); eng.steps(100); torch.cuda.synchronize()
            n = 1000 if N <= 100000 else 200
            ms = eng.timed_steps(n)
            row.append("%s(%d) %.2f" % (name, eng.layout.persistent, ms / n * 1e3))
            eng.close()
        print("N=%7d C=%4d: %s" % (N, C, "  ".join(row)), flush=True)
