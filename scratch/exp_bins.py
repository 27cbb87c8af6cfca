"""us per iteration by window size (table class): 1024 chains x 100k lineages, auto engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
for nb in (30, 64, 128, 200, 254, 300, 512, 600):
    ts, te, _ = synth.make_lineages(100000, nb, 10, 0)
    ts = np.concatenate([[0.0], ts]); te = np.concatenate([[float(nb) + 0.5], te])
    for C in (1024, 128):
        eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=20, engine="auto")
        eng.init(); eng.steps(200); torch.cuda.synchronize()
        ms = eng.timed_steps(1000)
        print("n_bins=%4d C=%4d: persistent=%d H=%s threads=%d  %.2f us/iter  %.3e evals/s" % (
            eng.n_bins, C, eng.layout.persistent, eng.kernel_name(), eng.layout.reserved1, ms, 1000.0 * len(ts) * C / (ms * 1e-3)), flush=True)
        eng.close()
