"""How often does a chain's ACCEPTED STATE change per iteration on cfg4-sized data?  (CPU, oracle loop on binned
statistics = the same trajectory as the device's.)  What the reject-speculation of the four-chain kernel rides on."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literate_amd import synth
from oracle import literate_oracle as lo, mcmc_oracle as mo

n_lin = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ts, te, _ = synth.make_lineages(n_lin, n_bins=128, n_shifts=20, seed=0)
t0, sp, ex, br = lo.bin_events_cli(ts, te)
stats = dict(sp=sp, ex=ex, br=br)
for chain in range(3):
    n_it = 30000
    with np.errstate(all="ignore"):
        out = mo.run_mcmc(stats, ts.min(), te.max(), mo.Settings(model_BDI=0), mo.PhiloxDraws(2026, chain), n_it, 1, k_max=32)
    rows = np.array(out["mcmc"])
    for a, b in ((0, 3000), (3000, 10000), (10000, 30000)):
        r = rows[a:b]
        changed = (np.diff(r[:, 2]) != 0) | (np.diff(r[:, 6]) != 0) | (np.diff(r[:, 7]) != 0)
        gibbs = (np.diff(r[:, 10]) != 0) | (np.diff(r[:, 12]) != 0)
        print("chain %d iterations %5d..%5d: state changed %.3f, gibbs %.4f, K_l %.1f K_m %.1f" % (
            chain, a, b, changed.mean(), gibbs.mean(), r[:, 6].mean(), r[:, 7].mean()))
