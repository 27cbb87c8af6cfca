import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
steps, warmup, se = int(sys.argv[1]), int(sys.argv[2]), 100
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
n_slots = (steps + warmup) // se + 2
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=se, n_trace_slots=n_slots)
eng.init()
t_spin = time.perf_counter()
while time.perf_counter() - t_spin < 0.4:
    eng.steps(256); torch.cuda.synchronize()
eng.init()
eng.steps(warmup)
torch.cuda.synchronize()
t = time.perf_counter(); eng.steps(steps); t1 = time.perf_counter() - t
heads = eng.trace[:, :, :13].contiguous(); t2 = time.perf_counter() - t
torch.cuda.synchronize(); t3 = time.perf_counter() - t
print('steps=%d issue %.2f ms, +heads %.2f ms, +sync %.2f ms -> %.1f us/step' % (steps, t1*1e3, t2*1e3, t3*1e3, t3/steps*1e6))
snap = eng.snapshot()
print('K_l mean %.2f max %d  K_m mean %.2f max %d accepted mean %.1f' % (snap['K_l'].mean(), snap['K_l'].max(), snap['K_m'].mean(), snap['K_m'].max(), snap['accepted'].mean()))
t = time.perf_counter(); eng.steps(steps); torch.cuda.synchronize(); print('again: %.1f us/step' % ((time.perf_counter()-t)/steps*1e6))
