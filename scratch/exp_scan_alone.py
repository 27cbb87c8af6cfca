"""cfg4: the packed scan alone (lr_pairscan_kernel: every pair's pending proposal scored once, tiles x pairs blocks of 256
threads, plain loads) against an iteration of the four-chain kernel - how much of an iteration is the scan itself?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
for C in (1024, 2048):
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
    eng.init(); eng.steps(3000); torch.cuda.synchronize()
    it = eng.timed_steps(2000) / 2000 * 1e3
    sc = eng.time_scan(50) * 1e3
    print("C=%d: iteration %.2f us; one scan of all %d pairs by lr_pairscan_kernel %.2f us (%s)" % (C, it, C // 2, sc, eng.kernel_name()))
    eng.close()
