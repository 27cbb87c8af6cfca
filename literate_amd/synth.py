"""Synthetic lineage data for the benchmark configurations.

Time axis [0, T] in unit bins; piecewise-constant true rates with `n_shifts` shifts per process
(min spacing 2), lambda ~ LogU(0.05, 0.6), mu ~ LogU(0.02, 0.3).  Exactly N lineages: birth times
are drawn from the expected-diversity-weighted birth intensity lambda(t) * D(t), lifetimes by
inversion of the piecewise-exponential death hazard; times are then discretised like the
reference's input files (integer ts/te, extant lineages end at T) and the CLI's death_jitter
0.5 is added to te (LRF:471).
"""
import numpy as np


def _piecewise(rng, T, n_shifts, lo, hi):
    while True:
        s = np.sort(rng.choice(np.arange(2, T - 1), size=n_shifts, replace=False)) if n_shifts else np.array([], int)
        edges = np.concatenate([[0], s, [T]])
        if n_shifts == 0 or np.min(np.diff(edges)) >= 2:
            break
    rates = np.exp(rng.uniform(np.log(lo), np.log(hi), n_shifts + 1))
    per_bin = np.repeat(rates, np.diff(edges))
    return edges.astype(float), rates, per_bin


def make_lineages(n, n_bins=128, n_shifts=20, seed=0, jitter=0.5):
    """Returns (ts[n], te[n], truth) as float64 numpy arrays; te already carries the jitter."""
    rng = np.random.default_rng(seed)
    T = int(n_bins)
    eL, rL, lam = _piecewise(rng, T, n_shifts, 0.05, 0.6)
    eM, rM, mu = _piecewise(rng, T, n_shifts, 0.02, 0.3)
    # expected diversity trajectory, growth capped so that births are not all in the last bins
    net = np.clip(np.cumsum(lam - mu), None, None)
    net = net * min(1.0, np.log(max(n, 2)) / max(net.max() - net.min(), 1e-9))
    w = lam * np.exp(net - net.max())
    cdf = np.concatenate([[0.0], np.cumsum(w)])
    cdf /= cdf[-1]
    # continuous birth times: inverse cdf of the piecewise-constant intensity
    ts_c = np.interp(rng.random(n), cdf, np.arange(T + 1, dtype=float))
    # death by inversion of the cumulative hazard H(t) = int_0^t mu
    H = np.concatenate([[0.0], np.cumsum(mu)])
    grid = np.arange(T + 1, dtype=float)
    target = np.interp(ts_c, grid, H) + rng.exponential(1.0, n)
    te_c = np.where(target >= H[-1], float(T) + 1.0, np.interp(target, H, grid))
    ts = np.floor(ts_c)
    te = np.where(te_c >= T, float(T), np.maximum(np.floor(te_c), ts))
    order = np.argsort(ts, kind="stable")          # files are written in order of appearance
    truth = dict(edges_l=eL, rates_l=rL, edges_m=eM, rates_m=rM, lam_bins=lam, mu_bins=mu)
    return ts[order].astype(np.float64), (te[order] + jitter).astype(np.float64), truth
