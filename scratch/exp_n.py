import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for n, C in ((100000, 1024), (1000000, 1024), (100000, 4096), (10000, 1024), (1000000, 128)):
    ts, te, _ = synth.make_lineages(n, 128, 20, 0)
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=4)
    eng.init(); eng.steps(8); torch.cuda.synchronize()
    ms = eng.time_scan(20)
    print('N=%d C=%d tiles=%d scan %.1f us -> %.3e pairs/s' % (n, C, eng.layout.tiles, ms * 1e3, n * C / ms * 1e3), flush=True)
    eng.close()
