"""Device simulator throughput: lineages created / living-lineage steps per second."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import ops, synth
for n_start, T, la, mu in ((100_000, 128, .06, .04), (1_000_000, 128, .05, .04)):
    lam, m = synth.rates_constant(T, la, mu)
    ops.simulate_bd(1000, 8, 1, lam, m)          # load kernels
    torch.cuda.synchronize()
    t = time.perf_counter()
    ts, te, trace = ops.simulate_bd(n_start, T, 3, lam, m, capacity=40_000_000)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    steps = int(trace.sum())
    print("SIM n_start=%d -> %d lineages, %d living-lineage steps in %.1f ms: %.2e lineage-steps/s, %.2e lineages/s, %.1f GB/s" % (
        n_start, ts.numel(), steps, el * 1e3, steps / el, ts.numel() / el, steps * 8 / el / 1e9), flush=True)
