"""General times: the log-likelihood the engines carry against lr_bd_loglik_batch (fp64 fractions) on the same accepted states."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import ops, synth
from literate_amd.engine import ChainEngine
from oracle import literate_oracle as lo
for n_lin in (100_000, 400_000, 1_300_000):
    ts, te, _ = synth.make_lineages(n_lin, n_bins=128, n_shifts=20, seed=4)
    rng = np.random.default_rng(9)
    grid = lambda x: np.round(x * 2.0 ** 32) / 2.0 ** 32
    ts = ts + grid(rng.uniform(0, 0.999, n_lin))
    te = np.maximum(np.ceil(te) - 1.0 + grid(rng.uniform(1e-3, 0.999, n_lin)), ts + 0.0078125)
    t0, sp, ex, br = lo.bin_events_cli(ts, te)
    for engine, C in (("persistent4", 24), ("spec", 24), ("spec", 6), ("launch", 24)):
        eng = ChainEngine(ts, te, C, model=0, seed=77, s_freq=1, n_trace_slots=40, engine=engine)
        eng.init(); eng.steps(40)
        snap = eng.snapshot(); n_bins = eng.n_bins
        lam = np.stack([snap["L"][c][lo.get_rate_index(np.floor(snap["tL"][c]), n_bins)] for c in range(C)])
        mu = np.stack([snap["M"][c][lo.get_rate_index(np.floor(snap["tM"][c]), n_bins)] for c in range(C)])
        lik = ops.bd_loglik_batch(eng.ts, eng.te, eng.t0, lam, mu, 0, br_length=br).cpu().numpy()
        d = lik - snap["likA"]
        print("N=%8d %-12s C=%2d persistent=%d: max |abs diff| %.3e  max rel %.3e   (diffs %s)" % (
            n_lin, engine, C, eng.layout.persistent, np.abs(d).max(), np.abs(d / lik).max(), np.array2string(d[:4], precision=4)), flush=True)
        eng.close()
