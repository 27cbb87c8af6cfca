"""us per iteration of the latency-bound configurations: cfg2 (metal_bands, 128 chains), cfg3 (10k lineages, 256 chains)
and a handful of chains, under both engines."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "binning_lik.npz"))
cases = [("cfg2 metal_bands", G["metal_bands/ts"], G["metal_bands/te"], 128, 2)]
ts3, te3, _ = synth.make_lineages(10000, 128, 20, 3)
cases.append(("cfg3 synthetic 10k", ts3, te3, 256, 0))
for C in (4, 16, 64):
    cases.append(("metal_bands C=%d" % C, G["metal_bands/ts"], G["metal_bands/te"], C, 2))
for name, ts, te, C, model in cases:
    for engine in ("persistent", "launch"):
        eng = ChainEngine(ts, te, C, model=model, seed=1, s_freq=100, n_trace_slots=80, engine=engine)
        eng.init(); eng.steps(500); torch.cuda.synchronize()
        n = 4000
        t = time.perf_counter(); eng.steps(n); torch.cuda.synchronize(); el = time.perf_counter() - t
        print('%-22s %-10s persistent=%d: %6.2f us/iter  %8.0f it/s/chain  %.3e evals/s' % (
            name, engine, eng.layout.persistent, el / n * 1e6, n / el, n * len(ts) * C / el), flush=True)
        eng.close()
