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


# ---- discrete-time simulator: rate generators of notebook 4 and the device run ----------
def rates_constant(time_len, lam, mu):
    """Constant_Rate_Generator."""
    return np.full(time_len, float(lam)), np.full(time_len, float(mu))


def rates_key_innovation(time_len, lam, mu, timing_innov, mag_innov):
    """Key_Innovation_Rate_Generator: the birth rate jumps by mag_innov from int(time_len * timing_innov) on."""
    la = np.full(time_len, float(lam))
    la[int(time_len * timing_innov):] += mag_innov
    return la, np.full(time_len, float(mu))


def rates_mass_extinction(time_len, base_la, base_mu, timing_ext, mag_ext, mag_rebound):
    """Mass_Extinction_Rate_Generator: one bin of extra death rate, then a birth-rate rebound."""
    k = int(time_len * timing_ext)
    la, mu = np.full(time_len, float(base_la)), np.full(time_len, float(base_mu))
    la[k + 1:] += mag_rebound
    mu[k] += mag_ext
    return la, mu


def rates_trend(trend, lambda_0, mu_0, alpha, beta):
    """Environmental_Trend_Rate_Generator: rate_t = base + slope * trend[t]."""
    trend = np.asarray(trend, dtype=float)
    return lambda_0 + alpha * trend, mu_0 + beta * trend


def simulate(n_start, time_len, scale=1, seed=0, rates=None, dd=None, capacity=None, jitter=0.5):
    """Run the reference's discrete-time birth-death scheme on the device (lr_simulate_bd) and return LiteRate input:
    (ts, te) in time units (floor(step / scale), as simulateRateABC.v2.py:218 floors them; extant lineages end at
    time_len) with death_jitter added to te, sorted by (ts, te), plus the living count per step.

    rates: (lam[time_len], mu[time_len]) per time unit from a generator above (evaluated at floor(t / scale) and
    divided by scale, like Simulator.run_simulation does);  dd: dict(mode=1|2, l0, m0, K) for diversity dependence."""
    from . import ops
    n_steps = int(time_len) * int(scale)
    if dd is None:
        lam, mu = rates
        t = np.arange(n_steps) // int(scale)
        ts_s, te_s, trace = ops.simulate_bd(n_start, n_steps, seed, np.asarray(lam, float)[t] / scale,
                                            np.asarray(mu, float)[t] / scale, capacity=capacity)
    else:
        ts_s, te_s, trace = ops.simulate_bd(n_start, n_steps, seed, mode=int(dd["mode"]), l0=dd["l0"], m0=dd["m0"],
                                            K=dd["K"], scale=scale, capacity=capacity)
    ts = np.floor(ts_s.cpu().numpy() / scale)
    te = np.floor(te_s.cpu().numpy() / scale)
    order = np.lexsort((te, ts))
    return ts[order], te[order] + jitter, trace.cpu().numpy()
