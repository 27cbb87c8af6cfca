import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
for C in (256, 512, 768, 1024, 1536, 2048, 4096):
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=40)
    eng.init(); eng.steps(300); torch.cuda.synchronize()
    n = 1000
    t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
    print('PERSIST=%s C=%4d persistent=%d: %.2f us/iter -> %.3e evals/s' % (os.environ.get('LR_PERSIST'), C, eng.layout.persistent, el / n * 1e6, n * 1e5 * C / el), flush=True)
    eng.close()
