import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=100)
eng.init(); eng.steps(256); torch.cuda.synchronize()
for n in (256, 1024):
    t=time.perf_counter(); eng.steps(n); t_issue=time.perf_counter()-t; torch.cuda.synchronize(); el=time.perf_counter()-t
    print(os.environ.get('LR_GRAPH_ITERS'), n, 'us/iter %.1f' % (el/n*1e6), 'issue us/iter %.1f' % (t_issue/n*1e6))
print('scan ms', eng.time_scan(50))
