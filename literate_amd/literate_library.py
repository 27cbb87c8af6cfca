"""Host-side mirror of the reference's literate_library.py function surface, backed by the HIP
kernels (no CPU compute path: every numeric entry point needs libliterate_hip.so and a GPU).

Same names, argument meaning and return types as the reference (lib = literate_library.py,
LRF = LiteRateForward.py in /root/reference).  Where the reference's functions read module
globals (`sp_events_bin`, `br_length_bin`, `model_BDI`, ... LRF:150-156) this module keeps the
same globals; `bind_lineages()` fills them from the data, as the CLI body does (LRF:515-577).
"""
# The reference's callers do `from literate_library import *` (DDRate.py:15, trend_rate.py) and rely on the names
# its own import block brings along (lib:8-21): numpy's namespace, np, scipy, stats, pd, csv, random, warn ...  So no
# __all__ here, and the same names at module level.
import argparse, os, sys   # noqa: E401
from numpy import *        # noqa: F401,F403
import numpy as np
import csv                 # noqa: F401
import random
import scipy.stats         # noqa: F401
from scipy import stats    # noqa: F401
from scipy.special import gamma, gdtr, gdtrix, betainc   # noqa: F401
from scipy.special import beta as f_beta                 # noqa: F401
from scipy.special import gammaln, betaln, xlogy, xlog1py
from warnings import warn
import pandas as pd        # noqa: F401

from . import ops
import builtins as _py      # the star import above shadows max / min / round / ... with numpy's, as in the reference

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
_sessions = {}          # (states per call, model) -> ops.LoglikSession on the bound lineages


def _host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def bind_lineages(ts_, te_, model=0):
    """What the CLI body does between parsing and runMCMC (LRF:473-476, 515-577): keep the
    lineage arrays in HBM and bin them into unit windows [i, i+1], i in range(int(min ts), int(max te))."""
    global ts, te, sp_events_bin, ex_events_bin, br_length_bin, ex_events_bin_dead, br_length_bin_dead
    global n_bins, model_BDI, only_dead, start_time, end_time, _t0, _ts_dev, _te_dev
    torch = ops._torch()
    _sessions.clear()
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
    nIn = int(_py.round(level * len(d)))
    if nIn < 2:
        raise RuntimeError("not enough data")
    i = int(np.argmin(d[nIn - 1:] - d[:len(d) - nIn + 1]))
    return np.array([d[i], d[i + nIn - 1]])


def random_choice(vector):
    """lib:69-72: [a uniformly chosen element, its position] - one np.random.choice draw, as the reference makes it."""
    where = np.random.choice(len(vector))
    return [vector[where], where]


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
    # one prepared session per (states per call, model): the per-iteration call of runMCMC (LRF:305-308) then costs one
    # upload of the rates, the launches, one read-back and one synchronisation - nothing is allocated or re-uploaded
    C = 1 if L.ndim == 1 else L.shape[0]
    # (br_length_bin is a module global a caller may edit or rebind, as in the reference: it goes up with every call.  The
    # device evaluates the per-lineage form: br_length_bin gives the constants log k_b and the k_b > 0 mask of LRF:150-162,
    # the exposure itself comes from the bound lineages - identical to the reference whenever the global is what
    # precompute_events returned for them)
    key = (C, model, int(n_bins), float(end_time), br_length_bin is None)
    ses = _sessions.get(key)
    if ses is None:
        if len(_sessions) >= 8:
            _sessions.clear()
        ses = _sessions[key] = ops.LoglikSession(_ts_dev, _te_dev, _t0, n_bins, C, model, br_length_bin, end_time)
    out = ses(L, M, br_length_bin)
    return np.float64(out[0]) if L.ndim == 1 else out.copy()


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
    kmax = _py.max(k + 1, 2)
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
    R = np.zeros((1, _py.max(len(q) + 1, 2))); R[0, :len(q)] = q
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


# ---- proposals on scalars / vectors that never touch the lineages: host numpy, the reference's draw order (lib:124-154) ----
def _slide(value, width):
    """the window move both sliding proposals share (lib:124-134): one uniform draw, centred, `width` wide"""
    return value + (np.random.random() - .5) * width


def update_sliding_win(i, m=0, M=1, d=0.05):
    """lib:124-128: reflected at the upper bound M; a lower bound of 0 reflects by sign."""
    moved = _slide(i, d)
    over = moved - M
    if over > 0:
        moved = M - over
    return abs(moved) if m == 0 else moved


def update_sliding_win_log(i, m=1, M=np.e, d=0.05):
    """lib:130-134: reflected at either bound."""
    moved = _slide(i, d)
    if moved > M:
        return M - (moved - M)
    if moved < m:
        return m + (m - moved)
    return moved


def update_normal_nobound(i, d=0.05):
    return i + np.random.normal(0, d)


def update_normal_nobound_vec(i, d=0.05, f=.75):
    """lib:140-145: every element moves by a normal step with probability f (the draws in the reference's order: the
    Bernoulli mask, then the steps); symmetric, so the Hastings term is 0."""
    shape = np.shape(i)
    moves = np.random.binomial(1, f, shape) != 0
    steps = np.random.normal(0, d, shape)
    return i + np.where(moves, steps, 0.), 0


def approx_log_fact(n):
    """Stirling's series as the reference writes it (lib:61-63)."""
    return np.log(np.sqrt((2 * n + 1. / 3) * np.pi)) + n * np.log(n) - n


def get_log_factorial(n):
    """lib:65-67; below 100 the reference takes log(n!) from scipy.misc.factorial, which no longer exists in scipy
    (SURVEY 8c): log Gamma(n + 1) is that value."""
    if n < 100:
        return float(gammaln(n + 1.0))
    return approx_log_fact(n)


def logPoisson_pmf(x, l):
    return (x * np.log(l) - l) - get_log_factorial(x)


def update_poisson_proposal(kt):
    ktp = np.random.poisson(kt)
    if ktp == 0:
        return kt, 0
    return ktp, logPoisson_pmf(kt, ktp) - logPoisson_pmf(ktp, kt)


# ---- priors.  lib's forms (lib:178-193: log-densities, element-wise) under the reference's names; the summed
# Gamma / Poisson priors of the CLI (LRF:198-202) run on the device under their own names ----
def prior_gamma(x, a, s, l):
    """scipy.stats.gamma.logpdf(x, a, scale=s, loc=l) in closed form (mean = a*s)."""
    y = (np.asarray(x, dtype=float) - l) / s
    with np.errstate(all="ignore"):
        out = np.where(y >= 0, xlogy(a - 1.0, y) - y - gammaln(a) - np.log(s), -np.inf)
    return out if np.ndim(x) else np.float64(out)


def prior_norm(x, l=0, s=1):
    y = (np.asarray(x, dtype=float) - l) / s
    out = -0.5 * y * y - 0.5 * np.log(2 * np.pi) - np.log(s)
    return out if np.ndim(x) else np.float64(out)


def prior_beta(x, a, b):
    x_ = np.asarray(x, dtype=float)
    with np.errstate(all="ignore"):
        out = np.where((x_ >= 0) & (x_ <= 1), xlog1py(b - 1.0, -x_) + xlogy(a - 1.0, x_) - betaln(a, b), -np.inf)
    return out if np.ndim(x) else np.float64(out)


def prior_sym_beta(x, a):
    return prior_beta(x, a, a)


def prior_gamma_LRF(L, a=2, b=2):
    """LiteRateForward.py's prior_gamma(L, a, b) (LRF:201-202): summed Gamma(a, rate b) log-density, on the device."""
    L = np.atleast_1d(np.asarray(L, dtype=float))
    R = np.ones((1, _py.max(len(L), 1))); R[0, :len(L)] = L
    return float(ops.log_priors(R, [len(L)], a, [b])[0])


def Poisson_prior(k, rate):
    """LRF:198-199 (k = number of rates), on the device."""
    R = np.ones((1, _py.max(k, 1)))
    base = ops.log_priors(R, [k], 2.0, [1.0])
    return float((ops.log_priors(R, [k], 2.0, [1.0], [rate]) - base)[0])


def print_R_vec(name, v):
    """lib:43-58: an R vector literal, NaN as NA."""
    vals = ["NA" if (isinstance(x, float) and np.isnan(x)) else x for x in v] if len(v) > 2 else list(v)
    if len(v) <= 2:
        return "%s=c(%s)" % (name, ",".join(str(x) for x in vals))
    return "%s=c(%s, %s%s)" % (name, vals[0], "".join("%s," % x for x in vals[1:-1]), vals[-1])


# ---- set-up (lib:196-229, 260-308) ----
def parse_ts_te(input_file, TBP, first_year, last_year, death_jitter):
    """lib:196-229.  The file's time columns -> (ts, te, PRESENT, ORIGIN) on a forward time axis, te with the jitter.

    The window [first_year, last_year] is given in the FILE's years: on a time-before-present axis the first year is the
    larger number.  Lineages born before the window are dropped, lineages born after its last year too, and deaths
    beyond the last year are cut to it.  One reference quirk is kept on purpose, because callers see it: on a TBP axis
    the reference filters the births by -last_year first and then indexes the deaths with a mask built from the
    ALREADY FILTERED births (lib:211-212) - as soon as that filter removes a lineage the mask no longer fits the
    deaths and numpy raises IndexError (tests/golden/parse_paths.npz holds the reference's own outcomes)."""
    table = pd.read_csv(input_file, delimiter='\t').to_numpy()
    if table.shape[1] == 4:
        warn('Four column (with clade) LiteRate input is deprecated. Use three columns.', FutureWarning)
        born, died = table[:, 2], table[:, 3]
    else:
        born, died = table[:, 1], table[:, 2]
    # "year a is not later than year b" on the file's axis: ages count down towards the present, calendar years up
    not_later = (lambda a, b: a >= b) if TBP else (lambda a, b: a <= b)
    if first_year != -1:
        inside = not_later(first_year, born)
        born, died = born[inside], died[inside]
    if last_year != -1:
        inside = not_later(born, last_year)
        if TBP:
            born = born[inside]
            died = died[not_later(born, last_year)]      # the quirk: a mask of the filtered births on the unfiltered deaths
        else:
            born, died = born[inside], died[inside]
        died = np.array(died)
        # still alive at the end of the window: a STRICT comparison, as the reference's (lib:212, 222) - a NaN death stays NaN
        later = (died < last_year) if TBP else (died > last_year)
        died[later] = last_year
    if TBP:
        oldest = max(born)
        ts_, te_ = oldest - born, oldest - died
    else:
        ts_, te_ = born, died
    te_ = te_ + death_jitter
    return ts_, te_, max(te_), min(ts_)


def print_empirical_rates(n_spec, n_exti, Dt):
    """lib:259-264: events per unit of lineage-time in every bin, printed under the reference's two headings."""
    rates = tuple(events / Dt for events in (n_spec, n_exti))
    for heading, r in zip(("EMPIRICAL BIRTH RATES:", "EMPIRICAL DEATH RATES:"), rates):
        print(heading)
        print(r)
    return rates


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
