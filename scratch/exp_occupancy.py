"""Does the speculative kernel's iteration time depend on how many CUs are busy?  Same per-block work, 16..256 blocks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
team = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
for C in (32, 64, 128, 256, 512):
    if (C // 2) * team > 256:
        continue
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=80, engine="spec", team=team)
    eng.init(); eng.steps(300); torch.cuda.synchronize()
    ms = eng.timed_steps(4000)
    print("N=%d C=%d team=%d blocks=%d: %.2f us/iter" % (N, C, eng.layout.team_blocks, (C // 2) * eng.layout.team_blocks, ms / 4000 * 1e3), flush=True)
    eng.close()
