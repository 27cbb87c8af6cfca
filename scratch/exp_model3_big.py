"""Model 3 (Keiding on the extinct lineages only) on 16 chains x 3e6 / 1e7 lineages: the planner's engine against the launches
on ts / te (the generic two-class scan)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd.engine import ChainEngine
for n in (3_000_000, 10_000_000):
    ts, te = bench.abi_lineages(n, False, "sorted")
    for engine in ("auto", "packed", "launch"):
        eng = ChainEngine(ts, te, 16, model=3, seed=2026, s_freq=100, n_trace_slots=8, engine=engine)
        eng.init(); eng.steps(40); torch.cuda.synchronize()
        us = min(eng.timed_steps(60) for _ in range(2)) / 60 * 1e3
        print("N=%.0e model 3 engine=%-6s %-36s: %8.2f us/iter" % (n, engine, eng.kernel_name()[:36], us), flush=True)
        eng.close()
    del ts, te
