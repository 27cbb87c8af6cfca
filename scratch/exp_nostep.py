"""cfg4 with the chain steps compiled out of the four-chain kernel (scratch/ab/nostep.so: -DLR_P4_NOSTEP; results void): what
the phases cost when they hold only the scans, barriers, sums and draws - against the kernel that ships."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=1 << 30, n_trace_slots=2, engine="persistent4")
eng.init(); eng.steps(300); torch.cuda.synchronize()
print("help trips %s: %.3f us per iteration (%s)" % (os.environ.get("LR_P4_HELP_TRIPS", "default"), eng.timed_steps(2000) / 2000 * 1e3, eng.kernel_name()))
