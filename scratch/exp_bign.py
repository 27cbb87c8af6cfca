import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for N in (300000, 1000000):
    ts, te, _ = synth.make_lineages(N, 128, 20, 0)
    for eng_name in ("persistent", "persistent4"):
        eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=10, engine=eng_name)
        eng.init(); eng.steps(100); torch.cuda.synchronize()
        n = 300
        t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
        print('BIG N=%7d %-11s variant %d: %.2f us/iter -> %.3e evals/s' % (N, eng_name, eng.layout.persistent, el / n * 1e6, n * N * 1024 / el), flush=True)
        eng.close()
