"""numpy restatement of the LiteRate hot path (binning, likelihoods, proposals, priors).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows: LRF = LiteRateForward.py, lib = literate_library.py,
DD = DDRate.py, BDIx = other/LiteRateBDI_ext.py (all under /root/reference).
Randomness is never drawn here: every proposal takes its draws as arguments so
that the HIP scorers can be fed the identical values.
"""
import math

import numpy as np

# hard-coded hyper-parameters of the reference CLI (LRF:586-590, LRF:101-102, LRF:165)
SHAPE_BETA_RJ = 10.0
MIN_ALLOWED_T = 1.0
GAMMA_SHAPE = 2.0
HP_GAMMA_SHAPE = 1.2
HP_GAMMA_RATE = 0.1
RJHP_GAMMA_SHAPE = 2.0
RJHP_GAMMA_RATE = 1.0
MULTIPLIER_D = 1.1
SMALL_NUMBER = 0.000000000000001  # DD:47


# ----------------------------------------------------------------------------
# A1/A2  sufficient statistics of one window  (lib:74-85, LRF:111-123)
# ----------------------------------------------------------------------------
def get_br(ts, te, t0, t1):
    """Total lineage-time inside [t0, t1]: sum of positive clipped overlaps (lib:74-79)."""
    dt = np.minimum(te, t1) - np.maximum(ts, t0)
    return np.sum(dt[dt > 0])


def precompute_events(ts, te, t0, t1):
    """(births in [t0,t1), deaths in (t0,t1], lineage-time) for one window (lib:81-85)."""
    n_spec = int(np.count_nonzero((ts >= t0) & (ts < t1)))
    n_exti = int(np.count_nonzero((te > t0) & (te <= t1)))
    return n_spec, n_exti, get_br(ts, te, t0, t1)


def bin_events_cli(ts, te):
    """Unit-bin statistics exactly as the CLI builds them (LRF:515-523, 566-568).

    Windows are [i, i+1] for i in range(int(min ts), int(max te)); the trailing
    partial bin is therefore dropped and extant lineages never count as deaths.
    Returns (t0, sp_events[int64], ex_events[int64], br_length[float64]).
    """
    lo, hi = int(np.min(ts)), int(np.max(te))
    sp, ex, br = [], [], []
    for i in range(lo, hi):
        a, b, c = precompute_events(ts, te, i, i + 1)
        sp.append(a), ex.append(b), br.append(c)
    return lo, np.array(sp, dtype=np.int64), np.array(ex, dtype=np.int64), np.array(br, dtype=np.float64)


def bin_events_dead(ts, te, end_time):
    """Death counts / lineage-time of the te<end_time subset (model_BDI 3; LRF:529-546)."""
    keep = te < end_time
    s, e = ts[keep], te[keep]
    lo, hi = int(np.min(ts)), int(np.max(te))
    ex, br = [], []
    for i in range(lo, hi):
        _, b, c = precompute_events(s, e, i, i + 1)
        ex.append(b), br.append(c)
    return np.array(ex, dtype=np.int64), np.array(br, dtype=np.float64)


def create_bins(origin, present, ts, te, rm_first_bin):
    """lib:231-257: windows from arange(origin, present+1), last bin always dropped."""
    edges = np.arange(origin, present + 1)
    sp, ex, br = [], [], []
    for i in range(len(edges) - 1):
        a, b, c = precompute_events(ts, te, edges[i], edges[i + 1])
        sp.append(a), ex.append(b), br.append(c)
    sp = np.array(sp)[:-1]
    ex = np.array(ex)[:-1]
    br = np.array(br)[:-1]
    if rm_first_bin:
        sp, ex, br = sp[1:], ex[1:], br[1:]
        origin += 1
    n_time_bins = len(br)
    return origin, present, sp, ex, br, n_time_bins, np.arange(n_time_bins).astype(float)


def parse_ts_te_array(t_file, TBP, first_year, last_year, death_jitter):
    """lib:196-229 applied to an already-loaded table (file IO is host-side, not oracle)."""
    if t_file.shape[1] == 4:
        ts_years, te_years = t_file[:, 2], t_file[:, 3]
    else:
        ts_years, te_years = t_file[:, 1], t_file[:, 2]
    ts_years = np.array(ts_years, dtype=float)
    te_years = np.array(te_years, dtype=float)
    if TBP:
        if first_year != -1:
            te_years = te_years[ts_years <= first_year]
            ts_years = ts_years[ts_years <= first_year]
        if last_year != -1:
            ts_years = ts_years[ts_years >= last_year]   # lib:210-211 filters te by the
            te_years = te_years[ts_years >= last_year]   # already-filtered ts (mirrored)
            te_years[te_years < last_year] = last_year
        ts = max(ts_years) - ts_years
        te = max(ts_years) - te_years
    else:
        if first_year != -1:
            te_years = te_years[ts_years >= first_year]
            ts_years = ts_years[ts_years >= first_year]
        if last_year != -1:
            te_years = te_years[ts_years <= last_year]
            ts_years = ts_years[ts_years <= last_year]
            te_years[te_years > last_year] = last_year
        ts, te = ts_years, te_years
    te = te + death_jitter
    return ts, te, max(te), min(ts)


# ----------------------------------------------------------------------------
# A3  rate index  (LRF:125-135)
# ----------------------------------------------------------------------------
def get_rate_index(times, n_bins):
    """Segment index of every unit bin; round() then integer bin widths (LRF:125-135)."""
    times = np.asarray(times, dtype=float)
    if len(times) == 2:
        return np.zeros(n_bins, dtype=int)
    widths = np.abs(np.diff(np.round(times + 0))).astype(int)
    return np.repeat(np.arange(len(widths)), widths)


# ----------------------------------------------------------------------------
# A4/A5  binned likelihoods  (LRF:137-162)
# ----------------------------------------------------------------------------
def BD_lik_Keiding(L_vec, M_vec, sp_events, ex_events, br_length, ex_dead=None, br_dead=None):
    """LRF:137-148; the *_dead pair replaces the death half for model_BDI 3."""
    blik = np.sum(np.log(L_vec) * sp_events - L_vec * br_length)
    if ex_dead is not None:
        dlik = np.sum(np.log(M_vec) * ex_dead - M_vec * br_dead)
    else:
        dlik = np.sum(np.log(M_vec) * ex_events - M_vec * br_length)
    return blik + dlik


def BDI_partial_lik(L_vec, M_vec, sp_events, ex_events, br_length, model_BDI):
    """LRF:150-162 with Tk = 1 (LRF:574); bins without lineage-time are dropped."""
    L = L_vec * (1 - model_BDI)
    I = L_vec * model_BDI
    M = M_vec
    k = br_length
    ok = k > 0
    lik = (sp_events[ok] * np.log(k[ok] * L[ok] + I[ok]) + ex_events[ok] * np.log(M[ok] * k[ok])
           - 1.0 * (k[ok] * (L[ok] + M[ok]) + I[ok]))
    return np.sum(lik)


def calc_likelihood(model_BDI, L_vec, M_vec, stats):
    """Dispatch of LRF:430-431.  stats: dict(sp, ex, br[, ex_dead, br_dead])."""
    if model_BDI <= 1:
        return BDI_partial_lik(L_vec, M_vec, stats["sp"], stats["ex"], stats["br"], model_BDI)
    if model_BDI == 3:
        return BD_lik_Keiding(L_vec, M_vec, stats["sp"], stats["ex"], stats["br"],
                              stats["ex_dead"], stats["br_dead"])
    return BD_lik_Keiding(L_vec, M_vec, stats["sp"], stats["ex"], stats["br"])


# ----------------------------------------------------------------------------
# A6  per-lineage form  (BDIx:124-160, forward-time orientation, LRF boundary rules)
# ----------------------------------------------------------------------------
def BD_partial_lik(ts, te, t0, t1, rate, par):
    """log(rate)*#events - rate*sum_i overlap_i for one window (BDIx:124-137).

    Forward time; births counted in [t0,t1), deaths in (t0,t1] as LRF:120-121 does
    (the Python-2 original runs on ages and uses (lo,up] for both).
    """
    if par == "l":
        n = np.count_nonzero((ts >= t0) & (ts < t1))
    else:
        n = np.count_nonzero((te > t0) & (te <= t1))
    return math.log(rate) * n - rate * get_br(ts, te, t0, t1)


def get_BDlik(ts, te, times, rates, par):
    """Sum of BD_partial_lik over the segments of a piecewise-constant rate (BDIx:139-146)."""
    return sum(BD_partial_lik(ts, te, times[i], times[i + 1], rates[i], par) for i in range(len(rates)))


def per_lineage_tables(lam_bins, mu_bins, model, br_length=None):
    """Per-bin event log-weights, exposure rates and the chain constant for a model.

    model 2/3: logB=log lam, logD=log mu, R=lam+mu                       (LRF:140-144)
    model 0  : bins with k>0: logB=log(k*lam), logD=log(mu*k), R=lam+mu  (LRF:160, I=0)
    model 1  : bins with k>0: logB=log(lam),   logD=log(mu*k), R=mu, const=-sum lam (L=0, I=lam)
    Bins with k==0 contribute nothing in models 0/1 (the k>0 mask of LRF:160).
    """
    lam = np.asarray(lam_bins, dtype=float)
    mu = np.asarray(mu_bins, dtype=float)
    const = 0.0
    if model >= 2:
        return np.log(lam), np.log(mu), lam + mu, const
    k = np.asarray(br_length, dtype=float)
    ok = k > 0
    logB = np.zeros_like(lam)
    logD = np.zeros_like(lam)
    R = np.zeros_like(lam)
    if model == 0:
        logB[ok] = np.log(k[ok] * lam[ok] + 0.0)
        R[ok] = lam[ok] + mu[ok]
    else:
        logB[ok] = np.log(k[ok] * 0.0 + lam[ok])
        R[ok] = mu[ok]
        const = -np.sum(lam[ok])
    logD[ok] = np.log(mu[ok] * k[ok])
    return logB, logD, R, const


def per_lineage_loglik_dense(ts, te, t0, lam_bins, mu_bins, model=2, br_length=None, chunk=4096):
    """Definition-level per-lineage log-likelihood: O(N*n_bins) overlap matrix.

    For lineage i and unit bin b=[t0+b, t0+b+1]:
        birth_ib = ts_i in [lo_b, hi_b) ; death_ib = te_i in (lo_b, hi_b] ;
        overlap_ib = max(0, min(te_i,hi_b) - max(ts_i,lo_b))
    loglik = sum_i sum_b birth_ib*logB_b + death_ib*logD_b - overlap_ib*R_b  (+ const)
    which is get_BDlik (BDIx:139-146) summed over the 'l' and 'm' processes at
    unit-bin resolution.  Lineages are processed in chunks to bound memory.
    """
    logB, logD, R, const = per_lineage_tables(lam_bins, mu_bins, model, br_length)
    n_bins = len(logB)
    lo = t0 + np.arange(n_bins, dtype=float)
    hi = lo + 1.0
    total = 0.0
    for a in range(0, len(ts), chunk):
        s = ts[a:a + chunk, None]
        e = te[a:a + chunk, None]
        birth = (s >= lo) & (s < hi)
        death = (e > lo) & (e <= hi)
        ov = np.minimum(e, hi) - np.maximum(s, lo)
        ov = np.where(ov > 0, ov, 0.0)
        total += np.sum(birth * logB + death * logD - ov * R)
    return total + const


def lineage_bins(ts, te, t0, n_bins):
    """Table index (0 = before window, 1..n_bins = unit bins, n_bins+1 = after) and
    in-bin fraction of every birth and death time, with the reference's boundary
    rules: births [lo,hi) -> floor, deaths (lo,hi] -> ceil-1."""
    bs = np.floor(ts - t0)
    # exact re-check against the integer-offset edges (ts - t0 may round)
    bs = np.where(t0 + bs > ts, bs - 1, bs)
    bs = np.where(t0 + bs + 1 <= ts, bs + 1, bs)
    be = np.ceil(te - t0) - 1
    be = np.where(t0 + be >= te, be - 1, be)
    be = np.where(t0 + be + 1 < te, be + 1, be)
    fs = ts - (t0 + bs)
    fe = te - (t0 + be)
    js = np.clip(bs + 1, 0, n_bins + 1).astype(np.int64)
    je = np.clip(be + 1, 0, n_bins + 1).astype(np.int64)
    fs = np.where((js == 0) | (js == n_bins + 1), 0.0, fs)
    fe = np.where((je == 0) | (je == n_bins + 1), 0.0, fe)
    return js, fs, je, fe


def per_lineage_loglik(ts, te, t0, lam_bins, mu_bins, model=2, br_length=None, pre=None):
    """O(N) gather form of the same sum (what the HIP scan kernel evaluates):
    contribution_i = (logB+cum)[js] + fs*R[js] + (logD-cum)[je] - fe*R[je],
    cum_b = sum_{b'<b} R_b'.  This is also bench.py's numpy CPU baseline."""
    logB, logD, R, const = per_lineage_tables(lam_bins, mu_bins, model, br_length)
    n_bins = len(logB)
    cum = np.concatenate(([0.0], np.cumsum(R)))
    SA = np.concatenate(([0.0], logB + cum[:-1], [cum[-1]]))
    SR = np.concatenate(([0.0], R, [0.0]))
    EA = np.concatenate(([0.0], logD - cum[:-1], [-cum[-1]]))
    js, fs, je, fe = pre if pre is not None else lineage_bins(ts, te, t0, n_bins)
    return np.sum(SA[js] + fs * SR[js] + EA[je] - fe * SR[je]) + const


# ----------------------------------------------------------------------------
# A7  multiplier proposals  (LRF:165-176, lib:156-173)
# ----------------------------------------------------------------------------
def update_multiplier_freq(q, ff, u, d=MULTIPLIER_D):
    """LRF:165-176 with the binomial mask ff and the uniforms u given."""
    m = np.exp(2 * np.log(d) * (np.asarray(u, dtype=float) - .5))
    m[np.asarray(ff) == 0] = 1.
    return q * m, np.sum(np.log(m))


def update_multiplier_proposal(q, u, d=MULTIPLIER_D):
    """lib:167-173 with its single uniform given."""
    m = np.exp(2 * np.log(d) * (u - .5))
    return q * m, np.log(m)


# ----------------------------------------------------------------------------
# A8  reversible-jump moves  (LRF:29-97)
# ----------------------------------------------------------------------------
def log_beta_sym_pdf(x, a=SHAPE_BETA_RJ):
    """scipy.stats.beta.logpdf(x, a, a) in closed form (LRF:22-23)."""
    return (a - 1.0) * np.log1p(-x) + (a - 1.0) * np.log(x) - (2.0 * math.lgamma(a) - math.lgamma(2.0 * a))


def add_shift_RJ_weighted_mean(rates, times, ind, delta_t, u):
    """LRF:29-47 with draws (interval index, offset inside it, Beta(10,10) variate) given."""
    rates = np.asarray(rates, dtype=float)
    times = np.asarray(times, dtype=float)
    r_time = times[ind + 1] - times[ind]
    t_prime = times[ind] + delta_t
    times_prime = np.sort(np.append(times, t_prime))
    t1, t2 = times[ind], times[ind + 1]
    p1 = (t1 - t_prime) / (t1 - t2)
    p2 = (t_prime - t2) / (t1 - t2)
    rate_i = rates[ind]
    logit = np.log((1 - u) / u)
    r1 = np.exp(np.log(rate_i) - p2 * logit)
    r2 = np.exp(np.log(rate_i) + p1 * logit)
    rates_prime = np.insert(rates, ind + 1, r2)
    rates_prime[ind] = r1
    log_q = np.log(abs(r_time)) - log_beta_sym_pdf(u)
    jac = 2 * np.log(r1 + r2) - np.log(rate_i)
    return rates_prime, times_prime, log_q + jac


def remove_shift_RJ_weighted_mean(rates, times, idx):
    """LRF:49-69 with the removed shift index (1..K-1) given.  Deletion is BY VALUE
    (LRF:56, 63): every entry equal to the removed time / rate goes."""
    rates = np.asarray(rates, dtype=float)
    times = np.asarray(times, dtype=float)
    t_prime = times[idx]
    t1, t2 = times[idx - 1], times[idx + 1]
    dT = abs(t2 - t1)
    times_prime = times[times != t_prime]
    p1 = (t1 - t_prime) / (t1 - t2)
    p2 = (t_prime - t2) / (t1 - t2)
    r1, r2 = rates[idx - 1], rates[idx]
    rate_prime = np.exp(p1 * np.log(r1) + p2 * np.log(r2))
    rates_prime = rates[rates != rates[idx]]
    rates_prime[idx - 1] = rate_prime
    u = 1. / (1 + r2 / r1)
    log_q = -np.log(dT) + log_beta_sym_pdf(u)
    jac = np.log(rate_prime) - 2 * np.log(r1 + r2)
    return rates_prime, times_prime, log_q + jac


# ----------------------------------------------------------------------------
# A10  priors  (LRF:198-213, 99-108)
# ----------------------------------------------------------------------------
def Poisson_prior(k, rate):
    """LRF:198-199; k is the number of RATES."""
    return k * np.log(rate) - rate - np.sum(np.log(np.arange(1, k + 1)))


def prior_gamma(L, a=2.0, b=2.0):
    """sum of Gamma(shape a, rate b) log-densities (LRF:201-202), evaluated the way
    scipy does: y = x/scale; (a-1) log y - y - lgamma(a) - log(scale)."""
    L = np.asarray(L, dtype=float)
    scale = 1. / b
    y = L / scale
    return np.sum((a - 1.0) * np.log(y) - y - math.lgamma(a) - np.log(scale))


def rj_hp_posterior_params(K_l, K_m):
    """Shape and scale of the Gibbs draw for the Poisson rate (LRF:99-108)."""
    return RJHP_GAMMA_SHAPE + K_l + K_m, 1. / (RJHP_GAMMA_RATE + 2)


def rate_hp_posterior_params(rates):
    """Shape and scale of the Gibbs draw for a Gamma-prior rate (LRF:210-213)."""
    rates = np.asarray(rates, dtype=float)
    return HP_GAMMA_SHAPE + GAMMA_SHAPE * len(rates), 1. / (HP_GAMMA_RATE + np.sum(rates))


# ----------------------------------------------------------------------------
# adequacy statistics written to the mcmc log (lib:268-279)
# ----------------------------------------------------------------------------
def calculate_r_squared(emp_birth, emp_death, est_birth, est_death):
    """Through-origin regression of estimated on empirical rates (lib:268-279),
    closed form of the single-column lstsq."""
    x = np.concatenate([emp_birth, emp_death])
    y = np.concatenate([est_birth, est_death])
    coeff = np.sum(x * y) / np.sum(x * x)
    fitted = coeff * x
    resid = y - fitted
    ssres = np.sum(resid ** 2)
    r2 = 1 - ssres / np.sum(y ** 2)
    var_fitted = np.var(fitted, ddof=1)
    gelman_r2 = var_fitted / (var_fitted + np.var(resid, ddof=1))
    return coeff, r2, gelman_r2


def calcHPD(data, level=0.95):
    """Narrowest interval holding `level` of the samples (lib:25-41)."""
    d = np.sort(np.asarray(data, dtype=float))
    n_in = int(round(level * len(d)))
    if n_in < 2:
        raise RuntimeError("not enough data")
    widths = d[n_in - 1:] - d[:len(d) - n_in + 1]
    i = int(np.argmin(widths))
    return np.array([d[i], d[i + n_in - 1]])


def marginal_rates_from_rows(rows, start_age, end_age, burnin=0.2):
    """Posterior marginal rate per unit bin (plotRJforward.v3.py:92-139).

    rows: list of 1-D arrays [K rates, K-1 shift times] (one *_rates.log row each).
    Returns (mean[nbins], hpd_lo, hpd_hi, matrix[samples, nbins]); bins are stored
    most-recent-first exactly as the plotting script does (:121).
    """
    nbins = abs(int(end_age - start_age))
    edges = np.arange(end_age, start_age)
    if burnin < 1:
        burnin = min(int(burnin * len(rows)), int(0.9 * len(rows)))
    out = []
    for row in rows[burnin:]:
        row = np.asarray(row, dtype=float)
        if len(row) == 1:
            out.append(np.zeros(nbins) + row[0])
            continue
        nr = int(np.ceil(len(row) / 2.))
        h = np.histogram(row[nr:], bins=edges)[0]
        out.append(row[:nr][np.cumsum(h)][::-1])
    mat = np.array(out)
    lo, hi = [], []
    for i in range(mat.shape[1]):
        a, b = calcHPD(mat[:, i], 0.95)
        lo.append(a), hi.append(b)
    return mat.mean(axis=0), np.array(lo), np.array(hi), mat


# ----------------------------------------------------------------------------
# A12  DDRate likelihood  (DD:55-107)
# ----------------------------------------------------------------------------
def dd_rates(args, DT, time_range, m_birth=2, m_death=2):
    """Per-bin birth/death rates, niche and niche fraction (DD:71-100)."""
    l_max, k, x0, div_0, L, m_max, nuB, nuD = args
    n = len(DT)

    def logistic():
        return div_0 + L / ((1 + np.exp(-k * (time_range - x0))) ** (1 / 1))

    def floor_rates(r):
        r = r.copy()
        r[r <= 0] = SMALL_NUMBER
        return r

    niche = np.ones(n)
    niche_frac = np.ones(n)
    if m_birth == 0:
        birth = np.ones(n) * l_max
    else:
        niche = np.ones(n) * (L + div_0) if m_birth == 1 else logistic()
        niche_frac = DT / niche
        birth = floor_rates(l_max - l_max * (niche_frac ** nuB))
    if m_death <= 0:
        death = np.ones(n) * m_max
    else:
        niche = np.ones(n) * (L + div_0) if m_death == 1 else logistic()
        niche_frac = DT / niche
        death = floor_rates(m_max + m_max * (niche_frac ** nuD))
    return birth, death, niche, niche_frac


def dd_likelihood_function(args, N_SPEC, N_EXTI, DT, time_range, m_birth=2, m_death=2):
    """DD:71-107: [lik[2], birth_rates, death_rates, niche, niche_frac]."""
    birth, death, niche, niche_frac = dd_rates(args, DT, time_range, m_birth, m_death)
    birth_lik = np.sum(np.log(birth) * N_SPEC - birth * DT)
    death_lik = np.sum(np.log(death) * N_EXTI - death * DT)
    return [np.array([birth_lik, death_lik]), birth, death, niche, niche_frac]


def _gamma1_logpdf(x, scale):
    """scipy.stats.gamma.logpdf(x, 1, scale=scale, loc=0) = -x/scale - log(scale) (x>=0)."""
    if x < 0:
        return -np.inf
    return -x / scale - np.log(scale)


def _norm_logpdf(x):
    return -0.5 * x * x - 0.5 * math.log(2 * math.pi)


def dd_calc_prior(args, prior_k0_l, origin, present):
    """DD:110-122."""
    p = _gamma1_logpdf(args[0], 10)
    p += _norm_logpdf(args[1])
    p += _gamma1_logpdf(args[5], 10)
    p += _gamma1_logpdf(args[3], prior_k0_l)
    p += _gamma1_logpdf(args[4], prior_k0_l)
    p += _norm_logpdf(args[6])
    p += _norm_logpdf(args[7])
    if origin + args[2] >= present:
        p = -np.inf
    return p


# ----------------------------------------------------------------------------
# the other rate maps that feed the same per-bin likelihood
# ----------------------------------------------------------------------------
def ddv2_rates(args, DT, time_range, m_birth=2, m_death=2):
    """DDRatev2.py:79-104 (get_logistic :55-56 with nu = 1, get_const_K :58-59, get_brates :61-65, get_drates
    :67-71): rates move linearly in niche_frac**nu between a floor rate and a multiple of it.  As written there:
    the death map is built from l_f (not a death parameter), m_death <= 0 gives death rates of exactly 1, and
    m_birth == 0 gives l_f * l_mul."""
    l_f, l_mul, k, x0, div_0, L, m_mul, nuB, nuD = args
    n = len(DT)

    def logistic():
        return div_0 + L / ((1 + np.exp(-k * (time_range - x0))) ** (1 / 1))

    def floor_rates(r):
        r = np.array(r, dtype=float)
        r[r <= 0] = SMALL_NUMBER
        return r

    niche = np.ones(n)
    niche_frac = np.ones(n)
    if m_birth == 0:
        birth = np.ones(n) * l_f * l_mul
    else:
        niche = np.ones(n) * (L + div_0) if m_birth == 1 else logistic()
        niche_frac = DT / niche
        rate_max = l_f + l_f * l_mul
        birth = floor_rates(rate_max - (rate_max - l_f) * (niche_frac ** nuB))
    if m_death <= 0:
        death = np.ones(n)
    else:
        niche = np.ones(n) * (L + div_0) if m_death == 1 else logistic()
        niche_frac = DT / niche
        rate_min = l_f - l_f * m_mul
        death = floor_rates(rate_min + (l_f - rate_min) * (niche_frac ** nuD))
    return birth, death, niche, niche_frac


def ddv2_likelihood_function(args, N_SPEC, N_EXTI, DT, time_range, m_birth=2, m_death=2):
    """DDRatev2.py:73-111: [lik[2], birth_rates, death_rates, niche, niche_frac]."""
    birth, death, niche, niche_frac = ddv2_rates(args, DT, time_range, m_birth, m_death)
    birth_lik = np.sum(np.log(birth) * N_SPEC - birth * DT)
    death_lik = np.sum(np.log(death) * N_EXTI - death * DT)
    return [np.array([birth_lik, death_lik]), birth, death, niche, niche_frac]


def normalise_trend(trend, rm_first_bin=0):
    """parse_trend_data, trend_rate.py:58-69, after the column has been read: drop the last bin (and the first with
    rm_first_bin), min-max scale to [0, 1], zeros become SMALL_NUMBER."""
    t = np.array(trend, dtype=float)[:-1]
    if rm_first_bin:
        t = t[1:]
    t = (t - np.min(t)) / (np.max(t) - np.min(t))
    t[t == 0] = SMALL_NUMBER
    return t


def trend_rates(args, TREND, const_birth=False, const_death=False):
    """trend_rate.py:73-88: rate_b = r_min + slope * TREND_b ** exponent, floored to SMALL_NUMBER."""
    l_min, m_min, alpha, beta, delta, gamma = args
    n = len(TREND)
    if const_birth:
        birth = np.ones(n) * l_min
    else:
        birth = l_min + alpha * TREND ** delta
        birth[birth <= 0.0] = SMALL_NUMBER
    if const_death:
        death = np.ones(n) * m_min
    else:
        death = m_min + beta * TREND ** gamma
        death[death <= 0.0] = SMALL_NUMBER
    return birth, death


def _gamma_logpdf(x, a, scale, loc=0.0):
    """scipy.stats.gamma.logpdf(x, a, scale=scale, loc=loc) in closed form."""
    y = (x - loc) / scale
    if y < 0:
        return -np.inf
    if y == 0:
        return -np.log(scale) if a == 1 else (-np.inf if a > 1 else np.inf)
    return (a - 1.0) * np.log(y) - y - math.lgamma(a) - np.log(scale)


def trend_calc_prior(args):
    """trend_rate.py:93-100."""
    p = _gamma_logpdf(args[0], 1, 10, .001)
    p += _gamma_logpdf(args[1], 1, 10, .001)
    p += _norm_logpdf(args[2] / 5.0) - np.log(5.0)
    p += _norm_logpdf(args[3] / 5.0) - np.log(5.0)
    p += _gamma_logpdf(args[4], 3, .5)
    p += _gamma_logpdf(args[5], 3, .5)
    return p


def trend_likelihood_function(args, N_SPEC, N_EXTI, DT, TREND, const_birth=False, const_death=False):
    """trend_rate.py:72-91: [lik[2], birth_rates, death_rates]."""
    birth, death = trend_rates(args, TREND, const_birth, const_death)
    birth_lik = np.sum(np.log(birth) * N_SPEC - birth * DT)
    death_lik = np.sum(np.log(death) * N_EXTI - death * DT)
    return [np.array([birth_lik, death_lik]), birth, death]
