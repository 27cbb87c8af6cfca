"""Long run of the four-chain kernel (cfg4 workload, model 0 and 3, unit resolution and general times): every chain's carried
log-likelihood must stay finite and equal an independent evaluation of its accepted state at the end."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import ops, synth
from literate_amd.engine import ChainEngine
from oracle import literate_oracle as lo
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
rng = np.random.default_rng(1)
tsg = ts + np.round(rng.uniform(0, 0.999, len(ts)) * 2.0 ** 32) / 2.0 ** 32
teg = np.maximum(np.ceil(te) - 1.0 + np.round(rng.uniform(1e-3, 0.999, len(ts)) * 2.0 ** 32) / 2.0 ** 32, tsg + 0.0078125)
for label, a, b, model in (("unit model 0", ts, te, 0), ("general model 0", tsg, teg, 0), ("unit model 2", ts, te, 2)):
    C = 1024
    eng = ChainEngine(a, b, C, model=model, seed=11, s_freq=1000, n_trace_slots=8, engine="persistent4")
    eng.init()
    t0 = time.time(); done = 0
    while done < n_it:
        k = min(50000, n_it - done)
        eng.steps(k); done += k
    torch.cuda.synchronize(); dt = time.time() - t0
    snap = eng.snapshot(); n_bins = eng.n_bins
    assert np.all(snap["it"] == n_it) and np.all(np.isfinite(snap["likA"])), label
    _, sp, ex, br = lo.bin_events_cli(a, b)
    lam = np.stack([snap["L"][c][lo.get_rate_index(np.floor(snap["tL"][c]), n_bins)] for c in range(C)])
    mu = np.stack([snap["M"][c][lo.get_rate_index(np.floor(snap["tM"][c]), n_bins)] for c in range(C)])
    lik = ops.bd_loglik_batch(eng.ts, eng.te, eng.t0, lam, mu, model, br_length=br).cpu().numpy()
    rel = np.abs((lik - snap["likA"]) / lik).max()
    print("%-16s %d iterations x %d chains in %.1f s (%.2f us/iter): K_l %d..%d, accepted %.3f, max rel diff of carried log-lik %.2e" % (
        label, n_it, C, dt, dt / n_it * 1e6, int(np.min(snap["K_l"])), int(np.max(snap["K_l"])), float(np.mean(snap["accepted"])) / n_it if "accepted" in snap else float("nan"), rel), flush=True)
    assert rel < 1e-9, label
    eng.close()
print("ok")
