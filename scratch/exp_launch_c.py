"""Launch-based engine, 100k lineages, by chain count: us per iteration, tiles, scan kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100_000, 128, 20, 0)
for C in (16, 32, 48, 64, 96, 128, 256, 1024):
    eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=1 << 30, n_trace_slots=2, engine="launch")
    eng.init(); eng.steps(300); torch.cuda.synchronize()
    v = sorted(eng.timed_steps(300) / 300 * 1e3 for _ in range(3))
    print("C=%5d: %-38s tiles=%d cb=%d  %.2f us/iter  scan alone %.2f us" % (C, eng.kernel_name()[:38], eng.layout.tiles, eng.layout.chains_per_block, v[1], eng.time_scan(20) * 1e3), flush=True)
    eng.close()
