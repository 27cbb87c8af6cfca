import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
n_chains = int(os.environ.get("LR_CHAINS", "1024"))
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, n_chains, model=int(os.environ.get("LR_MODEL", "0")), seed=1, s_freq=100, n_trace_slots=10)
eng.init(); eng.steps(300); torch.cuda.synchronize()
eng.close()
