"""Segments of the chain step inside lr_chain_step_kernel (launch-based engine, 16 chains x 1e7 lineages; needs a library built
with LR_EXTRA_FLAGS=-DLR_DIAG): clock64 stamps of the even waves' last step."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts0, te0, _ = synth.make_lineages(100000, 128, 20, 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ts, te = np.tile(ts0, N // len(ts0)), np.tile(te0, N // len(ts0))
eng = ChainEngine(ts, te, 16, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="launch")
eng.init(); eng.steps(300); torch.cuda.synchronize()
lib = _hip.load()
seg = (ctypes.c_ulonglong * (64 * 16))()
lib.lr_diag_dump_seg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
acc = {}
order = [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 6, 7]
names = {1: 'state + partials load, sum', 2: 'decide + Philox call', 3: 'move', 4: 'stage segments (log)', 5: 'prior', 9: 'table: entry',
         10: 'table: bin ranks', 11: 'table: rates of the bins', 12: 'table: prefix sum', 13: 'table: S, E writes',
         6: 'pair planes / end of propose', 7: 'bookkeeping', 8: 'state store'}
for rep in range(20):
    eng.steps(37 + rep); torch.cuda.synchronize()
    lib.lr_diag_dump_seg(seg, 64 * 16, 0)
    sg = np.frombuffer(seg, dtype=np.uint64).reshape(64, 16).astype(np.float64)[:16:2]
    for a_, b in zip(order[:-1], order[1:]):
        d = (sg[:, b] - sg[:, a_]) / 2400.0
        d = d[(d > 0) & (d < 30)]
        acc.setdefault(names[b], []).extend(d.tolist())
print({k: round(float(np.mean(v)), 3) for k, v in acc.items()}, 'sum %.2f' % sum(float(np.mean(v)) for v in acc.values()))
