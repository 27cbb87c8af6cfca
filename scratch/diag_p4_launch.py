"""Where the ~12 us per launch of the four-chain kernel go (needs LR_EXTRA_FLAGS=-DLR_DIAG python -m literate_amd.build):
wall-clock stamps (100 MHz) of blocks < 64 at the stages of ONE launch of n iterations - entry, state + tables in LDS, pair
planes, prologue done, iterations done, stores issued - against the HIP-event time of the same launch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=100, n_trace_slots=400)
eng.init(); eng.steps(3000); torch.cuda.synchronize()
lib = _hip.load()
N = 28672 + 64 * 8
buf = (ctypes.c_ulonglong * N)()
lib.lr_diag_dump_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry -> state, tables, constants in LDS", "pair planes", "prologue (carried sums, first draws)", "the n iterations",
         "state / table / carry stores issued"]
for n in (1, 20, 200):
    rows, ev = [], []
    for rep in range(9):
        eng.steps(50); torch.cuda.synchronize()
        ev.append(eng.timed_steps(n) * 1e3)
        lib.lr_diag_dump_step(buf, N)
        st = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[28672:].reshape(64, 8)[:, :6] / 100.0    # us
        rows.append(st)
    st = np.median(np.array(rows), axis=0)
    d = np.diff(st, axis=1)
    first, last = st[:, 0].min(), st[:, 5].max()
    print("n = %d iterations: HIP events %.2f us; first block's entry -> last block's last stamp %.2f us; block entries spread over %.2f us"
          % (n, np.median(ev), last - first, st[:, 0].max() - first))
    for j, nm in enumerate(names):
        print("    %-44s %7.2f us (mean over 64 blocks; max %.2f)" % (nm, d[:, j].mean(), d[:, j].max()))
    print("    per iteration inside the loop: %.3f us; events - (in-kernel span): %.2f us = markers, dispatch, drain" % (
        d[:, 3].mean() / n, np.median(ev) - (last - first)))
