"""Per-wave scan time inside the four-chain kernel (needs -DLR_DIAG): which scanner waves are the stragglers."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=4, engine="persistent4")
NIT = 400
eng.init(); eng.steps(NIT); torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step(buf, 4096 * 12)
a = np.frombuffer(buf, dtype=np.uint64)[20000:20000 + 64 * 16].reshape(64, 16).astype(np.float64) * 10 / 1000.0 / (2 * NIT)
print("per-wave scan us per phase, mean over 64 blocks:")
for w in range(2, 16):
    print("  wave %2d (SIMD %d): %.2f  (min %.2f max %.2f)" % (w, w & 3, a[:, w].mean(), a[:, w].min(), a[:, w].max()))
print("by SIMD:", [round(float(a[:, [w for w in range(2, 16) if (w & 3) == s]].mean()), 2) for s in range(4)])
print("slowest wave per block - mean wave:", round(float((a[:, 2:].max(1) - a[:, 2:].mean(1)).mean()), 2))
