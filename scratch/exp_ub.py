"""lr_bin_unit_events at 1e7 / 3e7 lineages under LR_UB_RSHIFT / LR_UB_BLOCKS_PER_CU (one setting per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
for n in (10_000_000, 30_000_000):
    for order in ("sorted", "shuffled"):
        ts, te = bench.abi_lineages(n, False, order)
        call, outs, info, keep = bench.abi_calls("lr_bin_unit_events", ts, te, 0)
        ms = bench.abi_time(call, 20)
        print("RSHIFT=%s BPC=%s N=%.0e %-8s: %.1f us  %.0f GB/s  frac %.3f" % (os.environ.get("LR_UB_RSHIFT"), os.environ.get("LR_UB_BLOCKS_PER_CU"), n, order, ms * 1e3, 16.0 * n / (ms * 1e-3) / 1e9, 16.0 * n / (ms * 1e-3) / 1e9 / 8000), flush=True)
        del ts, te, call, keep
