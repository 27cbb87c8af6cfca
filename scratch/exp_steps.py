import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=200)
eng.init(); eng.steps(512); torch.cuda.synchronize()
for K in (50, 500, 2000, 50, 33, 32, 31, 500, 100, 1):
    t = time.perf_counter(); eng.steps(K); ti = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
    print('K=%4d issue %.2f ms total %.2f ms -> %.1f us/step' % (K, ti * 1e3, tt * 1e3, tt / K * 1e6), flush=True)
