"""Work and barrier-wait time per wave and phase in lr_persist4_kernel (needs the LR_DIAG build: LR_EXTRA_FLAGS=-DLR_DIAG
python -m literate_amd.build): wall_clock64 sums (100 MHz) of blocks < 64 over a launch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
eng.init(); eng.steps(3000); torch.cuda.synchronize()
lib = _hip.load()
N = 16384 + 64 * 16 * 4
buf = (ctypes.c_ulonglong * N)()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step(buf, N)
b0 = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[16384:].reshape(64, 16, 4).copy()
n = 2000
ms = eng.timed_steps(n); torch.cuda.synchronize()
lib.lr_diag_dump_step(buf, N)
b1 = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[16384:].reshape(64, 16, 4)
d = (b1 - b0) / (2 * n) / 100.0          # us per phase
print("launch: %.2f us per iteration (%.2f per phase)" % (ms * 1e3 / n, ms * 1e3 / n / 2))
print("wave: work / wait (us per phase, mean over 64 blocks)")
for w in range(16):
    help_ = eng.kernel_name().endswith("true>")
    role = "step" if w < 2 else ("help" if help_ and w < 4 else "scan")
    print("  w%-2d %s  %.2f / %.2f%s" % (w, role, d[:, w, 0].mean(), d[:, w, 1].mean(),
                                       "   (of the work: %.2f waiting for the hand-over)" % d[:, w, 2].mean() if role == "help" else ""))
