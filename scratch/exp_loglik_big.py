"""lr_bd_loglik_batch at 3e7 / 1e8 lineages, C = 1, 8, 16, 32, 256: device time of the call (bench.py abi rows, quick
form; LR_SCAN_WIDE=0 plans the eight-chain kernel for every C)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
for n in (30_000_000, 100_000_000):
    ts, te = bench.abi_lineages(n, False, "sorted")
    for c in [int(x) for x in os.environ.get('LR_EXP_C', '1,8,16,32,256').split(',')]:
        call, outs, info, keep = bench.abi_calls("lr_bd_loglik_batch", ts, te, c)
        ms = bench.abi_time(call, 10 if c <= 32 else 3)
        gbs = 16.0 * n * info["passes"] / (ms * 1e-3) / 1e9
        print("N=%.0e C=%3d Cb=%2d: %8.1f us  %.0f GB/s  frac %.3f   %.3e evals/s" % (n, c, info["Cb"], ms * 1e3, gbs, gbs / 8000, n * c / (ms * 1e-3)), flush=True)
        del call, keep
    del ts, te
