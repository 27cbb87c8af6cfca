import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
rng = np.random.default_rng(5)
ts, te, _ = synth.make_lineages(20000, 128, 20, 0)
ts = ts + rng.uniform(0, 1, len(ts)) * 0.999
te = np.maximum(np.ceil(te) - 1.0 + rng.uniform(1e-3, 0.999, len(te)), ts + 1e-3)
ref = None
for C, eng_name, team in ((8, "launch", 0), (8, "spec", 1), (8, "spec", 2), (8, "persistent4", 0)):
    eng = ChainEngine(ts, te, C, model=0, seed=3, s_freq=1, n_trace_slots=100, engine=eng_name, team=team)
    eng.init(); eng.steps(100); torch.cuda.synchronize()
    tr = eng.trace_rows()[:, :, 2]
    if ref is None:
        ref = tr
    rel = np.abs(tr - ref) / np.abs(ref)
    first_bad = [int(np.argmax(rel[:, c] > 1e-9)) if (rel[:, c] > 1e-9).any() else -1 for c in range(C)]
    print(eng_name, team, 'persistent', eng.layout.persistent, 'mode', eng.layout.table_mode, eng.kernel_name(), 'max rel', rel.max(0), 'first bad it', first_bad, flush=True)
    eng.close()
