"""Per-wave phase times inside lr_spec_kernel (needs LR_EXTRA_FLAGS=-DLR_DIAG python -m literate_amd.build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth, _hip
from literate_amd.engine import ChainEngine
N, C = int(sys.argv[1]), int(sys.argv[2])
team = int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = 0
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
if os.environ.get("LR_DIAG_DATA") == "metal_bands":
    G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "binning_lik.npz"))
    ts, te, model = G["metal_bands/ts"], G["metal_bands/te"], 2
if os.environ.get("LR_DIAG_DATA") == "ddrate":
    from literate_amd.ddrate import DDRateEngine
    ts, te, _ = synth.make_lineages(50000, 64, 6, 0)
    eng = DDRateEngine(ts, te, float(ts.min()), float(te.max()), C, m_birth=2, m_death=2, seed=1, s_freq=100, n_trace_slots=40)
else:
    eng = ChainEngine(ts, te, C, model=model, seed=1, s_freq=100, n_trace_slots=40, engine="spec", team=team)
eng.init(); eng.steps(300); torch.cuda.synchronize()
NIT = 2000
lib = _hip.load()
seg = (ctypes.c_ulonglong * (64 * 16))()
lib.lr_diag_dump_seg_spec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lib.lr_diag_dump_seg_spec(seg, 64 * 16, 1)
ms = eng.timed_steps(NIT)
lib.lr_diag_dump_seg_spec(seg, 64 * 16, 1)
sg = np.frombuffer(seg, dtype=np.uint64).reshape(64, 16).astype(np.float64)[:min(C, 64)]
order = [8, 0, 2, 3, 4, 5, 6, 7]
names = {0: 'set load', 2: 'draws load', 3: 'move', 4: 'stage (log)', 5: 'prior', 6: 'tables', 7: 'set store'}
d = {names[b]: round(float(np.mean(sg[:, b] - sg[:, a_])) / 2400.0, 3) for a_, b in zip(order[:-1], order[1:])}
sub = {n: round(float(np.mean(sg[:, b] - sg[:, a_])) / 2400.0, 3) for n, a_, b in (('to builder', 5, 9), ('kb+marks/reuse', 9, 10), ('rate reads', 10, 11), ('f64 scan', 11, 12), ('writes', 12, 13), ('after', 13, 6))}
print('table builder:', sub)
print('candidate segments (last iteration, mean over chains, us at 2.4 GHz):', d, 'total %.2f' % (float(np.mean(sg[:, 7] - sg[:, 8])) / 2400.0))
print('N=%d C=%d team=%d: %.2f us/iter' % (N, C, eng.layout.team_blocks, ms / NIT * 1e3))
buf = (ctypes.c_ulonglong * (4096 * 12))()
lib.lr_diag_dump_step_spec.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lr_diag_dump_step_spec(buf, 4096 * 12)
a = np.frombuffer(buf, dtype=np.uint64)[16384:16384 + 64 * 16 * 4].reshape(64, 16, 4).astype(np.float64) * 10 / 1000.0 / NIT
nb = min(64, (C + 1) // 2 * eng.layout.team_blocks)
a = a[:nb]
print('per wave (mean over blocks), us per iteration: work | wait B1 | phase 2 | wait B2')
b2 = np.frombuffer(buf, dtype=np.uint64)[24576:24576 + 64 * 16 * 4].reshape(64, 16, 4).astype(np.float64)[:nb] * 10 / 1000.0 / NIT
print('phase 2 of the candidate waves: sums %.2f | decisions %.2f | select + roles %.2f (then clerk + stamp -> phase 2 total)' % tuple(b2[:, :4, k].mean() for k in range(3)))
for w in range(eng.layout.reserved1 // 64):
    print('wave %2d %s: %5.2f %5.2f %5.2f %5.2f' % (w, 'cand' if w < 4 else 'scan', *a[:, w].mean(0)))
print('scanner waves: scan | (wave 4) wait for the block | exchange  [us per iteration]')
for w in range(4, eng.layout.reserved1 // 64):
    print('wave %2d: %5.2f %5.2f %5.2f   (min/max over blocks of scan: %.2f %.2f; of exchange: %.2f %.2f)' % (w, *b2[:, w, :3].mean(0), b2[:, w, 0].min(), b2[:, w, 0].max(), b2[:, w, 2].min(), b2[:, w, 2].max()))
