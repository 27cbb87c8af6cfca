import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from literate_amd import synth
from literate_amd.ddrate import DDRateEngine
for N in (10000, 50000, 200000):
    ts5, te5, _ = synth.make_lineages(N, 64, 6, 0)
    for k in (1, 2):
        try:
            eng = DDRateEngine(ts5, te5, float(ts5.min()), float(te5.max()), 256, m_birth=2, m_death=2, seed=1, s_freq=100, n_trace_slots=80, team=k)
        except TypeError as e:
            print("no team kw", e); break
        eng.init(); eng.steps(300); torch.cuda.synchronize()
        ms = eng.timed_steps(3000)
        print("DD N=%d team=%d (got %d): %.2f us" % (N, k, eng.layout.team_blocks, ms / 3))
        eng.close()
