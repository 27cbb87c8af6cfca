"""cfg5: DDRate sampler, 50k synthetic lineages, 256 chains (and more chains) - us per iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.ddrate import DDRateEngine
ts, te, _ = synth.make_lineages(50_000, n_bins=64, n_shifts=6, seed=5)
for C, engine in ((256, "auto"), (256, "launch"), (1024, "auto"), (1024, "persistent4")):
    eng = DDRateEngine(ts, te, float(ts.min()), float(te.max()), C, seed=1, s_freq=100, n_trace_slots=60, engine=engine)
    eng.init(); eng.steps(500); torch.cuda.synchronize()
    n = 3000
    t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
    acc = eng.snapshot()["accepted"].mean() / eng.iterations
    print('DD C=%4d %-11s persistent=%d: %6.2f us/iter  %7.0f it/s/chain  %.3e evals/s  acc %.2f' % (
        C, engine, eng.layout.persistent, el / n * 1e6, n / el, n * len(ts) * C / el, acc), flush=True)
    eng.close()
