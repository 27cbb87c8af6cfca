"""Host-side mirror of the reference's literate_library.py function surface, backed by the HIP
kernels (no CPU compute path: every numeric entry point needs libliterate_hip.so and a GPU).

Same names, argument meaning and return types as the reference (lib = literate_library.py,
LRF = LiteRateForward.py in /root/reference).  Where the reference's functions read module
globals (`sp_events_bin`, `br_length_bin`, `model_BDI`, ... LRF:150-156) this module keeps the
same globals; `bind_lineages()` fills them from the data, as the CLI body does (LRF:515-577).
"""
import argparse
import random
from warnings import warn

import numpy as np

from . import ops

__all__ = ["calcHPD", "get_br", "precompute_events", "create_bins", "get_rate_index", "BD_lik_Keiding",
           "BDI_partial_lik", "BD_partial_lik", "get_BDlik", "update_multiplier_proposal",
           "update_multiplier_proposal_vec", "update_multiplier_freq", "add_shift_RJ_weighted_mean",
           "remove_shift_RJ_weighted_mean", "prior_gamma", "Poisson_prior", "parse_ts_te", "core_arguments",
           "set_seed", "calculate_r_squared", "print_empirical_rates", "random_choice", "bind_lineages"]

# ---- module globals the likelihood operators close over (LRF:150-156, 566-574) ----
ts = te = None
sp_events_bin = ex_events_bin = br_length_bin = None
ex_events_bin_dead = br_length_bin_dead = None
n_bins = 0
model_BDI = 0
only_dead = 0
start_time = end_time = 0.0
_t0 = 0.0
_ts_dev = _te_dev = None


def _host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def bind_lineages(ts_, te_, model=0):
    """What the CLI body does between parsing and runMCMC (LRF:473-476, 515-577): keep the
    lineage arrays in HBM and bin them into unit windows [i, i+1], i in range(int(min ts), int(max te))."""
    global ts, te, sp_events_bin, ex_events_bin, br_length_bin, ex_events_bin_dead, br_length_bin_dead
    global n_bins, model_BDI, only_dead, start_time, end_time, _t0, _ts_dev, _te_dev
    torch = ops._torch()
    ts, te = np.asarray(ts_, dtype=float), np.asarray(te_, dtype=float)
    _ts_dev, _te_dev = ops._dev(ts, torch.float64), ops._dev(te, torch.float64)
    start_time, end_time = float(np.min(ts)), float(np.max(te))
    _t0 = float(int(start_time))
    n_bins = int(end_time) - int(start_time)
    lo = _t0 + np.arange(n_bins, dtype=float)
    sp, ex, br = ops.bin_events(_ts_dev, _te_dev, lo, lo + 1.0)
    sp_events_bin, ex_events_bin, br_length_bin = _host(sp), _host(ex), _host(br)
    model_BDI, only_dead = int(model), int(model == 3)
    if only_dead:
        keep = te < end_time
        _, exd, brd = ops.bin_events(ts[keep], te[keep], lo, lo + 1.0)
        ex_events_bin_dead, br_length_bin_dead = _host(exd), _host(brd)
    return sp_events_bin, ex_events_bin, br_length_bin


# ---- log parsing helper (lib:25-41) ----
def calcHPD(data, level=0.95):
    assert (0 < level < 1)
    d = np.sort(np.asarray(list(data), dtype=float))
    nIn = int(round(level * len(d)))
    if nIn < 2:
        raise RuntimeError("not enough data")
    i = int(np.argmin(d[nIn - 1:] - d[:len(d) - nIn + 1]))
    return np.array([d[i], d[i + nIn - 1]])


def random_choice(vector):
    ind = np.random.choice(range(len(vector)))
    return [vector[ind], ind]


# ---- sufficient statistics (lib:74-85, 231-257) ----
def get_br(ts, te, t0, t1):
    return float(ops.bin_events(ts, te, [t0], [t1])[2][0])


def precompute_events(ts, te, t0, t1):
    sp, ex, br = ops.bin_events(ts, te, [t0], [t1])
    return int(sp[0]), int(ex[0]), float(br[0])


def create_bins(origin, present, ts, te, rm_first_bin):
    bins = np.arange(origin, present + 1)
    sp, ex, br = ops.bin_events(ts, te, bins[:-1], bins[1:])
    n_spec, n_exti, Dt = _host(sp)[:-1], _host(ex)[:-1], _host(br)[:-1]   # always drop last bin (lib:243-245)
    if rm_first_bin:
        n_spec, n_exti, Dt = n_spec[1:], n_exti[1:], Dt[1:]
        origin += 1
    n_time_bins = len(Dt)
    return origin, present, n_spec, n_exti, Dt, n_time_bins, np.arange(n_time_bins).astype(float)


def get_rate_index(times):
    """LRF:125-135 (integer host logic; the device twin is lr_expand_rates)."""
    if len(times) == 2:
        return np.zeros(n_bins).astype(int)
    t = np.round(np.asarray(times, dtype=float) + 0)
    return np.repeat(np.arange(len(t) - 1), np.abs(np.diff(t)).astype(int))


# ---- likelihood operators: calc_likelihood(L_acc_vec, M_acc_vec) (LRF:137-162, 430-431) ----
def _loglik(L_acc_vec, M_acc_vec, model):
    if _ts_dev is None:
        raise NameError("lineage data not bound: call bind_lineages(ts, te) first")
    L, M = np.asarray(L_acc_vec, dtype=float), np.asarray(M_acc_vec, dtype=float)
    if L.shape[-1] != n_bins or M.shape[-1] != n_bins:
        print(L.shape[-1], M.shape[-1], len(sp_events_bin))     # the reference's diagnostic (LRF:145-147)
        raise SystemExit
    out = ops.bd_loglik_batch(_ts_dev, _te_dev, _t0, L, M, model, br_length_bin, end_time)
    out = _host(out)
    return np.float64(out[0]) if L.ndim == 1 else out


def BD_lik_Keiding(L_acc_vec, M_acc_vec):
    return _loglik(L_acc_vec, M_acc_vec, 3 if only_dead else 2)


def BDI_partial_lik(L_acc_vec, M_acc_vec):
    return _loglik(L_acc_vec, M_acc_vec, 1 if model_BDI == 1 else 0)


def BD_partial_lik(arg):
    """[t0, t1, rate, par]: log(rate)*#events - rate*lineage-time in the window (BDIx:124-137),
    forward time, on the bound lineages."""
    [t0, t1, rate, par] = arg
    n_sp, n_ex, br = precompute_events(_ts_dev, _te_dev, t0, t1)
    return np.log(rate) * (n_sp if par == "l" else n_ex) - rate * br


def get_BDlik(times, rates, par):
    """BDIx:139-146: all segments in one launch."""
    times = np.asarray(times, dtype=float)
    sp, ex, br = [_host(x) for x in ops.bin_events(_ts_dev, _te_dev, times[:-1], times[1:])]
    ev = sp if par == "l" else ex
    return float(np.sum(np.log(np.asarray(rates, dtype=float)) * ev - np.asarray(rates) * br))


# ---- proposals: numpy draws in the reference's order, scored on the device ----
def _score(rates, times, move, index, draws):
    k = len(rates)
    kmax = max(k + 1, 2)
    R = np.zeros((1, kmax)); T = np.zeros((1, kmax + 1)); D = np.zeros((1, 2 * kmax))
    R[0, :k] = rates
    if times is not None:
        T[0, :k + 1] = times
    draws(D[0], kmax)
    r, t, kk, s = ops.rj_propose_score(R, T, [k], [move], [index], D)
    kk = int(kk[0])
    return _host(r)[0, :kk], _host(t)[0, :kk + 1], float(s[0])


def update_multiplier_proposal_vec(q, d=1.1, f=0.75):
    q = np.asarray(q, dtype=float)
    S = np.shape(q)
    ff = np.random.binomial(1, f, S)
    u = np.random.uniform(0, 1, S)

    def fill(D, kmax):
        D[:len(q)], D[kmax:kmax + len(q)] = ff, u
    R = np.zeros((1, max(len(q) + 1, 2))); R[0, :len(q)] = q
    D = np.zeros((1, 2 * R.shape[1])); fill(D[0], R.shape[1])
    r, _, _, s = ops.rj_propose_score(R, np.zeros((1, R.shape[1] + 1)), [len(q)], [0], [0], D, d)
    return _host(r)[0, :len(q)], float(s[0])


update_multiplier_freq = update_multiplier_proposal_vec


def update_multiplier_proposal(q, d=1.1):
    u = np.random.random()
    R = np.zeros((1, 2)); R[0, 0] = q
    D = np.zeros((1, 4)); D[0, 0], D[0, 2] = 1.0, u
    r, _, _, s = ops.rj_propose_score(R, np.zeros((1, 3)), [1], [0], [0], D, d)
    return float(r[0, 0]), float(s[0])


def add_shift_RJ_weighted_mean(rates, times):
    times = np.asarray(times, dtype=float)
    r_time, ind = random_choice(np.diff(times))
    delta = np.random.uniform(0, r_time)
    u = np.random.beta(10., 10.)

    def fill(D, kmax):
        D[0], D[1] = delta, u
    return _score(rates, times, 1, ind, fill)


def remove_shift_RJ_weighted_mean(rates, times):
    idx = np.random.choice(range(1, len(times) - 1))
    return _score(rates, times, 2, idx, lambda D, kmax: None)


# ---- priors (LRF:198-202) ----
def prior_gamma(L, a=2, b=2):
    L = np.atleast_1d(np.asarray(L, dtype=float))
    R = np.ones((1, max(len(L), 1))); R[0, :len(L)] = L
    return float(ops.log_priors(R, [len(L)], a, [b])[0])


def Poisson_prior(k, rate):
    R = np.ones((1, max(k, 1)))
    base = ops.log_priors(R, [k], 2.0, [1.0])
    return float((ops.log_priors(R, [k], 2.0, [1.0], [rate]) - base)[0])


# ---- set-up (lib:196-229, 260-308) ----
def parse_ts_te(input_file, TBP, first_year, last_year, death_jitter):
    import pandas as pd
    t_file = pd.read_csv(input_file, delimiter='\t').to_numpy()
    if t_file.shape[1] == 4:
        warn('Four column (with clade) LiteRate input is deprecated. Use three columns.', FutureWarning)
        ts_years, te_years = t_file[:, 2], t_file[:, 3]
    else:
        ts_years, te_years = t_file[:, 1], t_file[:, 2]
    if TBP:
        if first_year != -1:
            te_years = te_years[ts_years <= first_year]
            ts_years = ts_years[ts_years <= first_year]
        if last_year != -1:
            ts_years = ts_years[ts_years >= last_year]
            te_years = te_years[ts_years >= last_year]
            te_years[te_years < last_year] = last_year
        ts_, te_ = max(ts_years) - ts_years, max(ts_years) - te_years
    else:
        if first_year != -1:
            te_years = te_years[ts_years >= first_year]
            ts_years = ts_years[ts_years >= first_year]
        if last_year != -1:
            te_years = te_years[ts_years <= last_year]
            ts_years = ts_years[ts_years <= last_year]
            te_years[te_years > last_year] = last_year
        ts_, te_ = ts_years, te_years
    te_ = te_ + death_jitter
    return ts_, te_, max(te_), min(ts_)


def print_empirical_rates(n_spec, n_exti, Dt):
    print("EMPIRICAL BIRTH RATES:")
    print(n_spec / Dt)
    print("EMPIRICAL DEATH RATES:")
    print(n_exti / Dt)
    return (n_spec / Dt, n_exti / Dt)


def calculate_r_squared(emp_birth, emp_death, est_birth, est_death):
    """lib:268-279 (log column, host): through-origin regression in closed form."""
    x = np.concatenate([emp_birth, emp_death])
    y = np.concatenate([est_birth, est_death])
    coeff = np.sum(x * y) / np.sum(x * x)
    fitted = coeff * x
    resid = y - fitted
    r2 = 1 - np.sum(resid ** 2) / np.sum(y ** 2)
    var_fitted = np.var(fitted, ddof=1)
    return coeff, r2, var_fitted / (var_fitted + np.var(resid, ddof=1))


def set_seed(seed):
    rseed = np.random.randint(0, 9999) if seed == -1 else seed
    random.seed(rseed)
    np.random.seed(rseed)
    return rseed


def core_arguments():
    p = argparse.ArgumentParser()
    p.add_argument('-v', action='version', version='%(prog)s')
    p.add_argument('-d', type=str, help='data file', default="", metavar="")
    p.add_argument('-n', type=int, help='n. MCMC iterations', default=10000000, metavar=10000000)
    p.add_argument('-p', type=int, help='print frequency', default=1000, metavar=1000)
    p.add_argument('-s', type=int, help='sampling frequency', default=1000, metavar=1000)
    p.add_argument('-seed', type=int, help='seed (set to -1 to make it random)', default=-1, metavar=-1)
    p.add_argument('-TBP', help='Default is AD. Include for TBP.', default=False, action='store_true')
    p.add_argument('-first_year', type=int, help='first year of the dataset (unspecified for TBP)', default=-1, metavar=-1)
    p.add_argument('-last_year', type=int, help='last year of the dataset (unspecified for TBP)', default=-1, metavar=-1)
    p.add_argument('-death_jitter', type=float, help='amount added to death times', default=.5, metavar=.5)
    p.add_argument('-rm_first_bin', type=float, help='if set to 1 it removes the first time bin', default=0, metavar=0)
    p.add_argument('-print_emp', help='Prints empirical rates', default=False, action='store_true')
    return p
