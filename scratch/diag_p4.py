"""Per-wave work / barrier-wait times inside lr_persist4_kernel (needs LR_EXTRA_FLAGS=-DLR_DIAG python -m literate_amd.build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
eng.init(); eng.steps(300); torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step(buf, 4096 * 12)
a0 = np.frombuffer(buf, dtype=np.uint64)[16384:16384 + 64 * 16 * 4].reshape(64, 16, 4).astype(np.float64).copy()
NIT = 2000
ms = eng.timed_steps(NIT)
lib.lr_diag_dump_step(buf, 4096 * 12)
a = (np.frombuffer(buf, dtype=np.uint64)[16384:16384 + 64 * 16 * 4].reshape(64, 16, 4).astype(np.float64) - a0) * 10 / 1000.0 / (2 * NIT)
print('%.2f us/iter; per PHASE and wave (mean over 64 blocks): work | barrier wait' % (ms / NIT * 1e3))
for w in range(16):
    print('wave %2d %s: %5.2f %5.2f' % (w, 'step' if w < 2 else 'scan', a[:, w, 0].mean(), a[:, w, 1].mean()))
