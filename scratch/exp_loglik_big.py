"""lr_bd_loglik_batch at 1e7 / 3e7 lineages, C = 1, 8, 16: device time of the call (bench.py abi rows, quick form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
for n in (10_000_000, 30_000_000):
    ts, te = bench.abi_lineages(n, False, "sorted")
    for c in (1, 8, 16):
        call, outs, info, keep = bench.abi_calls("lr_bd_loglik_batch", ts, te, c)
        ms = bench.abi_time(call, 20)
        gbs = 16.0 * n * info["passes"] / (ms * 1e-3) / 1e9
        print("N=%.0e C=%2d: %.1f us  %.0f GB/s  frac %.3f" % (n, c, ms * 1e3, gbs, gbs / 8000), flush=True)
        del call, keep
    del ts, te
