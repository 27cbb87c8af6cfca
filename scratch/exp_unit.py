import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for n, C, unit in ((100000, 1024, True), (100000, 1024, False), (1000000, 1024, True), (1000000, 1024, False), (100000, 4096, True)):
    ts, te, _ = synth.make_lineages(n, 128, 20, 0)
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=4, unit_resolution=unit)
    eng.init(); eng.steps(8); torch.cuda.synchronize()
    ms = eng.time_scan(20)
    print('slots=%s N=%d C=%d unit=%s cb=%d tiles=%d scan %.1f us -> %.3e pairs/s' % (os.environ.get('LR_SLOTS'), n, C, unit, eng.layout.chains_per_block, eng.layout.tiles, ms * 1e3, n * C / ms * 1e3), flush=True)
    eng.close()
