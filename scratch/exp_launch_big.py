"""The launch-based engine on few chains x very many lineages (the regime where the ENGINE streams ts / te from HBM in every
iteration): us per iteration, evals/s and the HBM rate 16 B x N x ceil(C / Cb) per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts0, te0, _ = synth.make_lineages(100_000, 128, 20, 0)
for N, C in ((10_000_000, 16), (10_000_000, 64), (30_000_000, 16), (3_000_000, 16)):
    reps = N // len(ts0)
    ts, te = np.tile(ts0, reps), np.tile(te0, reps)
    for engine in ("auto", "launch"):
        eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=100, n_trace_slots=8, engine=engine)
        eng.init(); eng.steps(40); torch.cuda.synchronize()
        n = 100
        ms = eng.timed_steps(n)
        cb = eng.layout.chains_per_block
        passes = -(-C // cb) if not eng.layout.persistent else None
        us = ms / n * 1e3
        print("N=%.0e C=%3d engine=%-6s kernel %-40s persistent=%d team=%d: %8.2f us/iter  %.3e evals/s  %s" % (
            N, C, engine, eng.kernel_name()[:40], eng.layout.persistent, eng.layout.team_blocks, us, N * C / (us * 1e-6),
            ("HBM %.0f GB/s (Cb=%d, %d passes)" % (16.0 * N * passes / (us * 1e-6) / 1e9, cb, passes)) if passes else ""), flush=True)
        scan_ms = eng.time_scan(20) if not eng.layout.persistent else None
        if scan_ms:
            print("      scan kernel alone %.1f us = %.0f GB/s" % (scan_ms * 1e3, 16.0 * N * passes / (scan_ms * 1e-3) / 1e9))
        eng.close()
