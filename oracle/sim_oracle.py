"""Discrete-time birth-death lineage simulator: restatement of the scheme both of the reference's simulators use
(simulateRateABC.v2.py:103-234 `Simulator.simulate`; notebook 4 `Simulator.run_simulation`, `Population`).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for this file: both reference simulators live in
a plotting script / a Colab notebook that run a simulation and draw figures at import (matplotlib windows, tqdm
notebook widgets, altair) and draw from numpy's global stream, so no reference output was generated for them; the
scheme below is restated from their source text and the device is checked against this restatement only.
Per step t every living lineage draws one uniform r:
r < lambda_t -> it spawns a lineage born at t; lambda_t <= r < lambda_t + mu_t -> it dies at t (nb4: birther_indices /
dying_indices; ABC: r_sp_indx / r_ex_indx).  Lineages born at t first act at t + 1.  Rates per step come from a
rate generator evaluated at the scaled-down time, or from the living count (diversity dependence).
Randomness is the device's addressed Philox stream keyed by (seed, lineage slot): the newborns of one step are
exchangeable, so the multiset of (birth, death) pairs does not depend on the order slots are handed out.
"""
import numpy as np

from . import philox as px


def step_rates(mode, lam_steps, mu_steps, t, alive, l0, m0, K, scale):
    """mode 0: per-step probabilities as given; 1: nb4 Diversity_Dependence_Rate_Generator (la = max(0, l0 - l0 D/K),
    mu = max(0, m0 + m0 D/K)); 2: simulateRateABC.v2.py:153-154 (la = max(0, l0 - (l0 - m0) D/K), mu = max(0, m0 +
    (l0 - m0) D/K)); modes 1, 2 are divided by `scale` like there."""
    if mode == 0:
        return lam_steps[t], mu_steps[t]
    D = float(alive)
    if mode == 1:
        la, mu = max(0.0, l0 - l0 * D / K), max(0.0, m0 + m0 * D / K)
    else:
        la, mu = max(0.0, l0 - (l0 - m0) * D / K), max(0.0, m0 + (l0 - m0) * D / K)
    return la / scale, mu / scale


def simulate_bd(n_start, n_steps, seed, lam_steps=None, mu_steps=None, mode=0, l0=0.0, m0=0.0, K=1.0, scale=1.0,
                capacity=None):
    """Returns (ts, te, alive_trace): birth step and death step of every lineage (extant: te = n_steps), sorted by
    (ts, te), and the living count at the start of every step."""
    ts = np.zeros(n_start)
    te = np.full(n_start, float(n_steps))
    trace = np.zeros(n_steps, dtype=np.int64)
    for t in range(n_steps):
        alive = np.nonzero(te == n_steps)[0]
        trace[t] = len(alive)
        lt, mt = step_rates(mode, lam_steps, mu_steps, t, len(alive), l0, m0, K, scale)
        r = px.uniform_a_np(t, px.P_SIM, 0, seed & px.MASK, alive.astype(np.uint64))
        births = int(np.sum(r < lt))
        dying = alive[(r >= lt) & (r < lt + mt)]
        te[dying] = t
        if capacity is not None and len(ts) + births > capacity:
            raise OverflowError("capacity")
        ts = np.concatenate([ts, np.full(births, float(t))])
        te = np.concatenate([te, np.full(births, float(n_steps))])
    order = np.lexsort((te, ts))
    return ts[order], te[order], trace
