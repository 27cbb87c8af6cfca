import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4)
eng.init(); eng.steps(int(os.environ.get('NIT', '203'))); torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', lib.lr_diag_dump_step(buf, 4096 * 12))
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 12)[:1024].astype(np.int64)
names = ['load+partials', 'accept+trace', 'propose', 'stage(log)', 'prior', 'tables', 'store']
kinds = a[:, 8]
for mk, label in ((0, 'L-mult'), (1, 'L-times'), (2, 'M-mult'), (3, 'M-times'), (4, 'RJ'), (5, 'Gibbs')):
    sel = kinds == mk
    if sel.sum() == 0: continue
    d = (a[sel, 1:8] - a[sel, 0:7]) * 10 / 1000.0
    tot = (a[sel, 7] - a[sel, 0]) * 10 / 1000.0
    print('%-8s n=%4d total med %.2f max %.2f | ' % (label, sel.sum(), np.median(tot), tot.max()) + ' '.join('%s %.2f' % (n, np.median(d[:, i])) for i, n in enumerate(names)))
