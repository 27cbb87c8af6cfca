"""Segments of the chain step inside lr_persist4_kernel (needs LR_EXTRA_FLAGS=-DLR_DIAG python -m literate_amd.build):
clock64 stamps of the even stepper wave of chains < 64 during their last step."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
eng.init(); eng.steps(300); torch.cuda.synchronize()
lib = _hip.load()
seg = (ctypes.c_ulonglong * (64 * 16))()
lib.lr_diag_dump_seg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
acc = {}
for rep in range(20):
    eng.steps(37 + rep); torch.cuda.synchronize()
    lib.lr_diag_dump_seg(seg, 64 * 16, 0)
    sg = np.frombuffer(seg, dtype=np.uint64).reshape(64, 16).astype(np.float64)
    sg = sg[::2]                      # even chains: wave 0 of each block
    order = [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 6, 7, 8]
    names = {1: 'state load + decide', 2: 'Philox call', 3: 'move', 4: 'stage segments (log)', 5: 'prior', 9: 'table: entry',
             10: 'table: bin ranks', 11: 'table: rates of the bins', 12: 'table: prefix sum', 13: 'table: S, E writes',
             6: 'pair planes', 7: 'bookkeeping', 8: 'state store'}
    for a_, b in zip(order[:-1], order[1:]):
        d = (sg[:, b] - sg[:, a_]) / 2400.0
        d = d[(d > 0) & (d < 20)]
        acc.setdefault(names[b], []).extend(d.tolist())
print({k: round(float(np.mean(v)), 3) for k, v in acc.items()}, 'sum %.2f' % sum(float(np.mean(v)) for v in acc.values()))
