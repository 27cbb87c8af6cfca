"""In-kernel wall-clock stamps of the resident streaming kernel (a library built with -DLR_STREAM_STAMPS): the last but one
iteration of a launch, stepper of chain 0, first and last scanner block.  Microseconds relative to the stepper's start."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd import _hip
from literate_amd.engine import ChainEngine
n = int(float(os.environ.get("LR_EXP_SIZES", "1e7")))
ts, te = bench.abi_lineages(n, False, "sorted")
eng = ChainEngine(ts, te, 16, model=0, seed=2026, s_freq=100, n_trace_slots=8, sort_lineages=False)
eng.init(); eng.steps(40); torch.cuda.synchronize()
print(eng.kernel_name(), "%.2f us/iter" % (eng.timed_steps(200) / 200 * 1e3))
for rep in range(3):
    eng.steps(50); torch.cuda.synchronize()
    buf = (C.c_uint64 * (48 + 4096))()
    assert eng.lib.lr_stream_dump_stamps(buf) == 0
    v = np.array(buf[:], dtype=np.float64)
    t0 = v[0]
    us = lambda x: (x - t0) / 100.0
    print("stepper 0 : start 0.00 | early step done %.2f | released %.2f | all arrived seen %.2f | partials in %.2f | ready added %.2f" % tuple(us(v[i]) for i in (1, 2, 3, 4, 5)))
    for name, b in (("scanner 0 ", 16), ("scanner -1", 32)):
        print("%s: top %.2f | ready seen %.2f | scan+reduce done %.2f | partials drained %.2f | early/changed seen %.2f | next tables staged %.2f" % ((name,) + tuple(us(v[b + i]) for i in range(6))))
    blk = v[48:].reshape(1024, 4)
    blk = blk[blk[:, 3] > 0]
    seen, done = us(blk[:, 1]), us(blk[:, 3])
    dur = done - seen
    q = lambda x: " ".join("%.1f" % t for t in np.percentile(x, [0, 10, 50, 90, 99, 100]))
    print("  %d blocks: ready seen [min p10 p50 p90 p99 max] %s | done %s | scan time %s" % (len(blk), q(seen), q(done), q(dur)))
    order = np.argsort(done)
    print("  last to finish: blocks", order[-8:], "mod 8:", order[-8:] % 8, " first:", order[:8])
    for x in range(8):
        m = np.arange(len(blk)) % 8 == x
        print("    blocks = %d mod 8: done p50 %.1f max %.1f" % (x, np.median(done[m]), done[m].max()))
