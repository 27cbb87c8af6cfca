import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for N in (100000, 300000):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
    eng.init(); eng.steps(300); torch.cuda.synchronize()
    n = 1000
    t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
    print('SPLIT=%s N=%6d: %.2f us/iter -> %.3e evals/s' % (os.environ.get('LR_P4_SPLIT'), N, el / n * 1e6, n * N * 1024 / el), flush=True)
    eng.close()
