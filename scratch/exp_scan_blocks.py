"""Launch-based engine, 16 chains x 1e7 / 3e7 / 1e8 lineages: us per iteration and the scan kernel alone against the number
of lineage tiles (LR_SCAN_BLOCKS, read once per process: run per value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from literate_amd.engine import ChainEngine
for n in (10_000_000, 30_000_000, 100_000_000):
    ts, te = bench.abi_lineages(n, False, "sorted")
    eng = ChainEngine(ts, te, 16, model=0, seed=2026, s_freq=100, n_trace_slots=8, sort_lineages=False)
    eng.init(); eng.steps(40); torch.cuda.synchronize()
    us = eng.timed_steps(100) / 100 * 1e3
    scan_us = eng.time_scan(20) * 1e3
    print("LR_SCAN_BLOCKS=%s n=%.0e: %7.2f us/iter (%.3f of 8 TB/s)   scan alone %7.2f us (%.3f)   tiles %d" % (
        os.environ.get("LR_SCAN_BLOCKS", "2048"), n, us, 16.0 * n / (us * 1e-6) / 8e12, scan_us, 16.0 * n / (scan_us * 1e-6) / 8e12,
        int(eng.layout.tiles) if hasattr(eng.layout, "tiles") else -1), flush=True)
    eng.close(); del eng, ts, te
    torch.cuda.empty_cache()
