import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4)
assert eng.layout.persistent >= 1
eng.init(); eng.steps(200); torch.cuda.synchronize()
NIT = 400
import time
t = time.perf_counter(); eng.steps(NIT); torch.cuda.synchronize(); print('us/iter %.2f' % ((time.perf_counter() - t) / NIT * 1e6))
lib = _hip.load()
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step(buf, 4096 * 12)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 12)[:340].astype(np.float64)
for k, name in ((9, 'scan'), (10, 'reduce+barrier'), (11, 'step+barrier')):
    v = a[:, k] * 10 / 1000.0 / NIT
    print('%-16s per iteration: min %.2f med %.2f max %.2f us' % (name, v.min(), np.median(v), v.max()))

b = np.frombuffer(buf, dtype=np.uint64)[2048 * 12: 2048 * 12 + 1024].reshape(512, 2).astype(np.int64)
t0 = b[:, 0].min()
st = (b - t0) * 10 / 1000.0
print('block start min/med/max %.1f %.1f %.1f us ; end min/med/max %.1f %.1f %.1f us' % (st[:, 0].min(), np.median(st[:, 0]), st[:, 0].max(), st[:, 1].min(), np.median(st[:, 1]), st[:, 1].max()))
late = (st[:, 0] > 100).sum()
print('blocks starting >100us after the first:', late, ' life med %.1f us' % np.median(st[:, 1] - st[:, 0]))
life = (b[:, 1] - b[:, 0]) * 10 / 1000.0 / NIT
order = np.argsort(life[:340])
for idx in list(order[:4]) + list(order[-4:]):
    print('block %3d life/iter %.2f  scan %.2f red %.2f step %.2f' % (idx, life[idx], a[idx, 9] * 10 / 1000 / NIT, a[idx, 10] * 10 / 1000 / NIT, a[idx, 11] * 10 / 1000 / NIT))
print('corr(life, blockIdx) = %.2f' % np.corrcoef(life, np.arange(512))[0, 1])
print('life by block range:', [round(float(life[i:i + 64].mean()), 2) for i in range(0, 512, 64)])
