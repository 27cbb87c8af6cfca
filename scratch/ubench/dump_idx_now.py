"""Dump the packed lineage groups of the cfg4 workload in the CURRENT format (scratch/ubench/scan_now)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
os.environ["LR_P4_HELP_TRIPS"] = "0"
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4, engine="persistent4")
eng.init(); torch.cuda.synchronize()
off = int(eng.layout.lineage_idx)
raw = eng.workspace[off:off + 9000 * 16].cpu().numpy().reshape(-1, 16)
cnt = raw.view(np.uint16)[:, 0] & 0xf
n_real = int(np.nonzero(cnt)[0].max()) + 1
raw[:n_real].tofile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/idx8_now.bin")
print("groups", n_real, "lineages in them", int(raw[:n_real].view(np.uint16)[:, 0].astype(np.int64).__and__(0xf).sum()))
