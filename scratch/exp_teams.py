"""Iteration time of the speculative kernel over (lineages, chains per team, team size): the data behind lr_spec_model.
    python scratch/exp_teams.py <cpb 1|2> [general|dd]"""
import sys, os
cpb = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["LR_SPEC_CPB"] = cpb
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
from literate_amd.ddrate import DDRateEngine
mode = sys.argv[2] if len(sys.argv) > 2 else "unit"
C = 32
for N in (1000, 3000, 10000, 20000, 30000, 50000, 100000, 200000):
    ts, te, _ = synth.make_lineages(N, 64 if mode == "dd" else 128, 6 if mode == "dd" else 20, 0)
    if mode == "general":
        rng = np.random.default_rng(5)
        ts = ts + rng.uniform(0, 1, len(ts)) * 0.999
        te = np.maximum(te + rng.uniform(-0.49, 0.49, len(te)), ts + 1e-3)
    row = []
    for k in (1, 2, 4, 8):
        if mode == "dd":
            eng = DDRateEngine(ts, te, float(ts.min()), float(te.max()), C, m_birth=2, m_death=2, seed=1, s_freq=100, n_trace_slots=80,
                               engine="spec", team=k)
        else:
            eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=80, engine="spec", team=k)
        eng.init(); eng.steps(3000); torch.cuda.synchronize()
        ms = eng.timed_steps(3000)
        row.append("k=%d %.2f" % (eng.layout.team_blocks, ms / 3000 * 1e3))
        eng.close()
    print("cpb=%s N=%6d groups~%5d %s: %s" % (cpb, N, (N + 13) // 14, mode, "  ".join(row)), flush=True)
