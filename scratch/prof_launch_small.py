import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ts0, te0, _ = synth.make_lineages(100000, 128, 20, 0)
reps = max(1, N // 100000)
ts, te = np.tile(ts0, reps), np.tile(te0, reps)
eng = ChainEngine(ts, te, 16, model=0, seed=2026, s_freq=100, n_trace_slots=8, engine="launch")
print(eng.layout.tiles, eng.kernel_name())
eng.init(); eng.steps(100); torch.cuda.synchronize()
eng.close()
