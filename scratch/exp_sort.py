import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
rng = np.random.default_rng(0)
def run(tag, ts, te):
    eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=100)
    eng.init(); eng.steps(64); torch.cuda.synchronize()
    print(tag, 'scan ms %.4f' % eng.time_scan(50), eng.snapshot()['likA'][:2])
    eng.close()
run('ts-sorted(te random)', ts, te)
o = np.lexsort((te, ts)); run('sorted (ts,te)', ts[o], te[o])
o = np.lexsort((ts, te)); run('sorted (te,ts)', ts[o], te[o])
p = rng.permutation(len(ts)); run('random order', ts[p], te[p])
