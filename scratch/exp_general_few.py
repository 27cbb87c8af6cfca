"""The launch-based engine on general lineage times, few chains x very many lineages: 8 / 12 / 16 chains x 1e7 / 3e7 (16 B per
lineage and pass; 9-16 chains take ONE pass of the sixteen-chain scan, LR_ENGINE_WIDE=0: two pipelined halves of eight)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd.engine import ChainEngine
for n in [int(float(x)) for x in os.environ.get("LR_EXP_SIZES", "1e7,3e7").split(",")]:
    ts, te = bench.abi_lineages(n, True, "sorted")
    for C in (8, 12, 16):
        eng = ChainEngine(ts, te, C, model=0, seed=2026, s_freq=100, n_trace_slots=8, engine=os.environ.get("LR_EXP_ENGINE", "launch"))
        eng.init(); eng.steps(40); torch.cuda.synchronize()
        us = min(eng.timed_steps(100) for _ in range(3)) / 100 * 1e3
        print("N=%.0e C=%2d %-30s Cb=%2d pipelined=%d: %8.2f us/iter  %.3e evals/s" % (n, C, eng.kernel_name()[:30], eng.layout.chains_per_block, eng.layout.pipelined, us, n * C / (us * 1e-6)), flush=True)
        eng.close()
    del ts, te
