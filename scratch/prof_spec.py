import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
N, C = int(sys.argv[1]), int(sys.argv[2])
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=40, engine=sys.argv[3] if len(sys.argv) > 3 else "spec")
eng.init(); eng.steps(2000); torch.cuda.synchronize()
print(eng.kernel_name(), eng.layout.team_blocks)
eng.close()
