import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4)
eng.init(); eng.steps(8); torch.cuda.synchronize()
print('scan ms', eng.time_scan(5), 'tiles', eng.layout.tiles, 'cb', eng.layout.chains_per_block)
lib = _hip.load()
n_blocks = eng.layout.tiles * (1024 // eng.layout.chains_per_block)
buf = (ctypes.c_ulonglong * (8192 * 8))()
lib.lr_diag_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', lib.lr_diag_dump(buf, 8192 * 8))
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:n_blocks].astype(np.int64)
t0 = a[:, 0].min()
st = (a[:, :5] - t0) * 10.0 / 1000.0   # us
print('blocks', n_blocks)
print('start   min/med/max  %.2f %.2f %.2f' % (st[:, 0].min(), np.median(st[:, 0]), st[:, 0].max()))
print('end     min/med/max  %.2f %.2f %.2f' % (st[:, 4].min(), np.median(st[:, 4]), st[:, 4].max()))
for k, name in ((1, 'staging'), (2, 'loop'), (3, 'barrier'), (4, 'reduce+store')):
    d = st[:, k] - st[:, k - 1]
    print('%-14s min/med/max  %.2f %.2f %.2f' % (name, d.min(), np.median(d), d.max()))
d = st[:, 4] - st[:, 0]
print('block life     min/med/max  %.2f %.2f %.2f' % (d.min(), np.median(d), d.max()))
xcc = a[:, 6] & 0xf
print('blocks per xcc', np.bincount(xcc, minlength=8))
# do tiles of one group share an XCC?
grp = (a[:, 7] & 0xffffffff) // eng.layout.chains_per_block
same = [len(set(xcc[grp == g])) for g in range(int(grp.max()) + 1)]
print('distinct XCCs per chain group: min %d max %d' % (min(same), max(same)))
# concurrency over time
ev = sorted([(s, 1) for s in st[:, 0]] + [(e, -1) for e in st[:, 4]])
cur = 0; last = 0; hist = {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += dlt
tot = sum(hist.values())
print('resident blocks (time-weighted): avg %.0f ; >=900: %.0f%% ; <=256: %.0f%%' % (sum(k * v for k, v in hist.items()) / tot, 100 * sum(v for k, v in hist.items() if k >= 900) / tot, 100 * sum(v for k, v in hist.items() if k <= 256) / tot))
