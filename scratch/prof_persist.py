import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=100, n_trace_slots=40)
assert eng.layout.persistent >= 1
n_it = int(os.environ.get("LR_PROF_ITERS", "1000"))       # (at most 4096: one launch)
eng.init(); eng.steps(n_it); torch.cuda.synchronize()      # exactly ONE persistent-kernel launch of n_it iterations
eng.close()
