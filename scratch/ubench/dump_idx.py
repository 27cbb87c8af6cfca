"""Dump the packed lineage groups of the cfg4 workload (for scratch/ubench/lds_gather: the scan's real LDS access pattern)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4, engine="persistent4")
eng.init(); torch.cuda.synchronize()
off = eng.layout.lineage_idx
n8 = (N + 13) // 14 + 200
raw = eng.workspace[off:off + n8 * 16].cpu().numpy().reshape(-1, 16)
n_real = int((raw[:, 1] > 0).sum())
raw[:n_real].tofile(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/idx8.bin")
print("groups", n_real, "H", eng.layout.H if hasattr(eng.layout, "H") else "?")
