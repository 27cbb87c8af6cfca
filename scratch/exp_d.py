import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
eng.init(); eng.steps(300); torch.cuda.synchronize()
n = 1500
t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
print('D %s/%s: %.2f us/iter' % (os.environ.get('LR_P4_D1'), os.environ.get('LR_P4_D2'), el / n * 1e6), flush=True)
