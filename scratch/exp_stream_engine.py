"""The ENGINE on 16 chains x 1e7 / 3e7 / 1e8 lineages at unit resolution (bench.py's abi.engine_streaming rows, quick form):
us per iteration, the scan kernel alone, 16 B x N per pass against 8 TB/s.  LR_EXP_SIZES=1e7,3e7 picks the sizes,
LR_EXP_ENGINES=auto,launch the engine modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from literate_amd.engine import ChainEngine
sizes = [int(float(x)) for x in os.environ.get("LR_EXP_SIZES", "1e7,3e7,1e8").split(",")]
for n in sizes:
    ts, te = bench.abi_lineages(n, False, "sorted")
    for engine in os.environ.get("LR_EXP_ENGINES", "auto,stream").split(","):
        eng = ChainEngine(ts, te, 16, model=0, seed=2026, s_freq=100, n_trace_slots=8, sort_lineages=os.environ.get('LR_EXP_SORT', '1') == '1', engine=engine)
        eng.init(); eng.steps(40); torch.cuda.synchronize()
        it = 200 if n <= 30_000_000 else 60
        us = min(eng.timed_steps(it) for _ in range(3)) / it * 1e3
        cb = eng.layout.chains_per_block
        passes = -(-16 // cb)
        line = "N=%.0e engine=%-6s %-34s tiles=%d: %8.2f us/iter = %.3f of 8 TB/s" % (n, engine, eng.kernel_name()[:34], eng.layout.tiles, us, 16.0 * n * passes / (us * 1e-6) / 8e12)
        if not eng.layout.persistent:
            sc = eng.time_scan(20) * 1e3
            line += "   scan alone %.2f us = %.3f" % (sc, 16.0 * n * passes / (sc * 1e-6) / 8e12)
        print(line, flush=True)
        eng.close()
    del ts, te
