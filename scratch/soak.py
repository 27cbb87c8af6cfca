import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, truth = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=99, s_freq=1000, n_trace_slots=400)
eng.init()
t = time.perf_counter()
for k in range(8):
    eng.steps(50000); torch.cuda.synchronize()
    s = eng.snapshot()
    print('it %d  %.1f s  likA mean %.1f (finite %s)  K_l mean %.2f max %d  K_m mean %.2f max %d  acc rate %.3f  poi %.2f g0 %.2f' % (
        s['it'][0], time.perf_counter() - t, s['likA'].mean(), np.isfinite(s['likA']).all(), s['K_l'].mean(), s['K_l'].max(),
        s['K_m'].mean(), s['K_m'].max(), s['accepted'].mean() / s['it'][0], s['poi'].mean(), s['gamma_rate'][:, 0].mean()), flush=True)
tr = eng.trace_rows()
print('trace rows', tr.shape, 'finite heads', np.isfinite(tr[:, :, :13]).all())
