"""How full are the packed groups of the bench's abi lineages (16 chains x N)?  Reads lineage_idx back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd.engine import ChainEngine
n = int(float(os.environ.get("LR_EXP_SIZES", "1e7")))
ts, te = bench.abi_lineages(n, False, "sorted")
eng = ChainEngine(ts, te, 16, model=0, seed=2026, s_freq=100, n_trace_slots=8, sort_lineages=os.environ.get('LR_EXP_SORT', '1') == '1')
eng.init(); torch.cuda.synchronize()
lay = eng.layout
idx = eng.workspace[int(lay.lineage_idx):int(lay.lineage_idx) + (n // 4 + 4096) * 16].view(torch.int32).view(-1, 4).cpu().numpy().view(np.uint32)
hdr = idx[:, 0] & 0xffff
cnt = hdr & 0xf
used = np.nonzero(cnt)[0]
n8 = used.max() + 1
print(eng.kernel_name(), "groups", n8, "lineages/group %.2f" % (n / n8), "count histogram", np.bincount(cnt[:n8], minlength=16))
# slots: how many of the 7 slots are non-padding (offset != E[0] entry = H*16)?
H = int(lay.table_stride)
offs = np.stack([(idx[:n8, 0] >> 16), idx[:n8, 1] & 0xffff, idx[:n8, 1] >> 16, idx[:n8, 2] & 0xffff, idx[:n8, 2] >> 16, idx[:n8, 3] & 0xffff, idx[:n8, 3] >> 16], 1)
print("slots used per group: mean %.2f" % (offs != H * 16).sum(1).mean(), " plane histogram of slots", np.bincount((offs // (H * 16)).ravel(), minlength=6))
