"""The engine on the reference's own small datasets (example_TBP: 75 lineages; metal_bands: 30k), 1 .. 1024 chains: what the
planner picks and us per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd.engine import ChainEngine
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "binning_lik.npz"))
for name in ("example_TBP", "metal_bands"):
    ts, te = G[name + "/ts"], G[name + "/te"]
    for C in (1, 16, 128, 256, 1024):
        for team in ((0, 1) if C <= 128 else (0,)):
            eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=1000, n_trace_slots=40, team=team)
            eng.init(); eng.steps(3000); torch.cuda.synchronize()
            v = sorted(eng.timed_steps(2000) / 2000 * 1e3 for _ in range(3))
            print("%-12s %6d lineages C=%4d team_request=%d: %-44s persistent=%d team=%d  %.2f us/iter" % (
                name, len(ts), C, team, eng.kernel_name()[:44], eng.layout.persistent, eng.layout.team_blocks, v[1]), flush=True)
            eng.close()
