"""Per-wave scan time inside the two-chain kernel (needs -DLR_DIAG)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
N = int(os.environ.get("N", "100000")); C = int(os.environ.get("C", "1024"))
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=4, engine="persistent")
assert eng.layout.persistent == 1
NIT = 400
eng.init(); eng.steps(NIT); torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step(buf, 4096 * 12)
nb = min(512, (C + 1) // 2)
a = np.frombuffer(buf, dtype=np.uint64)[20000:20000 + nb * 8].reshape(nb, 8).astype(np.float64) * 10 / 1000.0 / NIT
for lo, hi, name in ((0, min(256, nb), "first 256 blocks (older)"), (256, nb, "blocks 256.. (younger)")):
    if hi > lo:
        print(name, "per-wave scan us:", np.round(a[lo:hi].mean(0), 2), " slowest-mean %.2f" % float((a[lo:hi].max(1) - a[lo:hi].mean(1)).mean()))
