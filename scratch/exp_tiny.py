import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for N, C in ((200, 256), (10000, 256), (10000, 8), (200, 8)):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=60, engine="spec", team=1)
    eng.init(); eng.steps(500); torch.cuda.synchronize()
    ms = eng.timed_steps(4000)
    print('N=%d C=%d: %.2f us/iter' % (N, C, ms / 4000 * 1e3), flush=True)
    eng.close()
