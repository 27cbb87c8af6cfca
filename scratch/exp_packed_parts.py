"""The packed launches in one partition or two (LR_PACKED_PARTS): us per iteration over (lineages, chains)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd.engine import ChainEngine
for n, chains in ((10_000_000, (16, 32, 64, 128)), (30_000_000, (16, 32)), (3_000_000, (128, 256))):
    ts, te = bench.abi_lineages(n, False, "sorted")
    for C in chains:
        out = []
        for parts in ("1", "2"):
            os.environ["LR_PACKED_PARTS"] = parts
            eng = ChainEngine(ts, te, C, model=0, seed=2026, s_freq=100, n_trace_slots=8, engine="packed")
            eng.init(); eng.steps(40); torch.cuda.synchronize()
            us = min(eng.timed_steps(100) for _ in range(3)) / 100 * 1e3
            out.append("%d partition(s) %7.2f us" % (eng.layout.n_parts, us))
            eng.close()
        print("N=%.0e C=%3d (n C = %.1e): %s" % (n, C, n * C, "   ".join(out)), flush=True)
    del ts, te
