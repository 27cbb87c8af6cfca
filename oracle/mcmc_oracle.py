"""Restatement of the reference RJMCMC driver, runMCMC (LiteRateForward.py:216-373).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The loop body follows the
reference statement by statement, including its quirks (SURVEY.md section 8a:
A9 no-op time move, A10 stale Poisson prior, initial prior with rate 2,
delete-by-value).  Randomness comes from a pluggable draw source:

* ``NumpyLegacyDraws`` consumes numpy's global MT19937 stream in the reference's
  own call order, so with the reference's seed this loop reproduces the
  reference's trajectory (that is how the restatement is pinned, see
  tests/golden/make_golden.py);
* ``PhiloxDraws`` uses the addressed Philox stream of oracle/philox.py, i.e.
  the randomness the HIP engine uses, so device trajectories can be compared
  step by step.
"""
import math

import numpy as np

from . import literate_oracle as lo
from . import philox as px


class Settings:
    """CLI-level switches the loop reads as module globals (LRF:411-437, 586-590)."""

    def __init__(self, model_BDI=0, const_rates=0, const_death_rate=0, use_rate_HP=1,
                 Poisson_HP=0.0, update_fraction=0.75, calc_adequacy=1):
        self.model_BDI = model_BDI
        self.const_rates = const_rates
        self.const_death_rate = const_death_rate
        self.use_rate_HP = use_rate_HP
        self.Poisson_HP = Poisson_HP
        self.update_fraction = update_fraction
        self.calc_adequacy = calc_adequacy


class NumpyLegacyDraws:
    """np.random.* in the reference's call order (LRF:30-38, 50, 74, 106, 167-168,
    181, 190, 212, 234, 313, 580-581)."""

    def init_rates(self):
        L = np.random.gamma(2, 2, 1)
        M = np.random.gamma(2, 2, 1)
        return L, M

    def move(self, it):
        r = np.random.random(2)
        return r[0], r[1]

    def multiplier(self, it, K, f):
        ff = np.random.binomial(1, f, K)
        u = np.random.uniform(0, 1, K)
        return ff, u

    def times_move(self, it, K):
        idx = np.random.choice(range(1, K))
        r = np.random.random()
        return idx, r

    def rj_select(self, it):
        r = np.random.random(2)
        return r[0], r[1]

    def rj_add(self, it, times):
        ind = np.random.choice(range(len(times) - 1))
        delta = np.random.uniform(0, np.diff(times)[ind])
        u = np.random.beta(lo.SHAPE_BETA_RJ, lo.SHAPE_BETA_RJ)
        return ind, delta, u

    def rj_remove(self, it, K):
        return np.random.choice(range(1, K))

    def gibbs_poi(self, it, shape, scale):
        return np.random.gamma(shape, scale)

    def gibbs_rate(self, it, side, shape, scale):
        return np.random.gamma(shape=shape, scale=scale)

    def accept(self, it):
        return np.random.random()


class PhiloxDraws:
    """The device's addressed stream (literate_amd/csrc/lr_device.h) for one chain."""

    def __init__(self, seed, chain):
        self.s = px.Stream(seed, chain)

    def init_rates(self):
        L = np.array([self.s.gamma(0, px.P_INIT, 0, 2.0) * 2.0])
        M = np.array([self.s.gamma(0, px.P_INIT, 64, 2.0) * 2.0])
        return L, M

    def move(self, it):
        return self.s.pair(it, px.P_MOVE, 0)

    def multiplier(self, it, K, f):
        ff = np.zeros(K, dtype=int)
        u = np.zeros(K)
        for j in range(K):
            ub, uu = self.s.pair(it, px.P_MULT, j)
            ff[j] = 1 if ub < f else 0
            u[j] = uu
        return ff, u

    def times_move(self, it, K):
        uc, ur = self.s.pair(it, px.P_TIMES, 0)
        return 1 + min(int(uc * (K - 1)), K - 2), ur

    def rj_select(self, it):
        return self.s.pair(it, px.P_RJ, 0)

    def rj_add(self, it, times):
        K = len(times) - 1
        uc, ud = self.s.pair(it, px.P_RJ, 1)
        ind = min(int(uc * K), K - 1)
        delta = ud * (times[ind + 1] - times[ind])
        ga = self.s.gamma(it, px.P_BETA_A, 0, lo.SHAPE_BETA_RJ)
        gb = self.s.gamma(it, px.P_BETA_B, 0, lo.SHAPE_BETA_RJ)
        return ind, delta, ga / (ga + gb)

    def rj_remove(self, it, K):
        uc, _ = self.s.pair(it, px.P_RJ, 1)
        return 1 + min(int(uc * (K - 1)), K - 2)

    def gibbs_poi(self, it, shape, scale):
        return self.s.gamma(it, px.P_GIBBS_POI, 0, shape) * scale

    def gibbs_rate(self, it, side, shape, scale):
        return self.s.gamma(it, px.P_GIBBS_L if side == 0 else px.P_GIBBS_M, 0, shape) * scale

    def accept(self, it):
        return self.s.pair(it, px.P_ACCEPT, 0)[0]


def _rjmcmc(draws, it, L, M, timesL, timesM, sample_shift_mu, k_max):
    """Dispatcher of LRF:71-97.  k_max caps the number of rates (device limit; the
    reference has none): an add at K == k_max is returned as forced-reject."""
    r0, r1 = draws.rj_select(it)
    newL, newtL, qL = L, timesL, 0
    newM, newtM, qM = M, timesM, 0
    overflow = False
    if r0 > sample_shift_mu:
        if r1 > 0.5:
            if k_max is not None and len(L) >= k_max:
                overflow = True
            else:
                newL, newtL, qL = lo.add_shift_RJ_weighted_mean(L, timesL, *draws.rj_add(it, timesL))
        elif len(L) > 1:
            newL, newtL, qL = lo.remove_shift_RJ_weighted_mean(L, timesL, draws.rj_remove(it, len(L)))
        update_L = 1
    else:
        if r1 > 0.5:
            if k_max is not None and len(M) >= k_max:
                overflow = True
            else:
                newM, newtM, qM = lo.add_shift_RJ_weighted_mean(M, timesM, *draws.rj_add(it, timesM))
        elif len(M) > 1:
            newM, newtM, qM = lo.remove_shift_RJ_weighted_mean(M, timesM, draws.rj_remove(it, len(M)))
        update_L = 0
    return newL, newtL, newM, newtM, qL + qM, update_L, overflow


def run_mcmc(stats, start_time, end_time, settings, draws, n_iterations, s_freq,
             init=None, emp=None, k_max=None, lik_fn=None):
    """runMCMC (LRF:216-373).  Returns dict(mcmc=[rows], sp=[rows], ex=[rows]).

    stats: dict(sp, ex, br[, ex_dead, br_dead]); n_bins = len(stats['sp']).
    init : optional (L, M, timesL, timesM); default = CLI init (LRF:580-583).
    emp  : (B_EMP, D_EMP) for the adequacy columns, or None to skip them.
    lik_fn(L_vec, M_vec): override of calc_likelihood (e.g. the per-lineage form).
    """
    n_bins = len(stats["sp"])
    st = settings
    if lik_fn is None:
        def lik_fn(Lv, Mv):
            return lo.calc_likelihood(st.model_BDI, Lv, Mv, stats)

    if init is None:
        L_acc, M_acc = draws.init_rates()
        timesLA = np.array([start_time, end_time], dtype=float)
        timesMA = np.array([start_time, end_time], dtype=float)
    else:
        L_acc, M_acc, timesLA, timesMA = [np.array(x, dtype=float) for x in init]

    Poi_lambda_rjHP = 1 if st.Poisson_HP == 0 else st.Poisson_HP
    Gamma_rate = [1., 1.]
    indLA = lo.get_rate_index(timesLA, n_bins)
    indMA = lo.get_rate_index(timesMA, n_bins)
    likA = lik_fn(L_acc[indLA], M_acc[indMA])
    priorA = lo.prior_gamma(L_acc) + lo.prior_gamma(M_acc)          # rate 2 here (LRF:227)
    priorA += -np.log(end_time - start_time) * (len(L_acc) - 1 + len(M_acc) - 1)
    priorPoiA = lo.Poisson_prior(len(L_acc), Poi_lambda_rjHP) + lo.Poisson_prior(len(M_acc), Poi_lambda_rjHP)
    priorA += priorPoiA

    out = {"mcmc": [], "sp": [], "ex": []}
    for it in range(n_iterations):
        r0, r1 = draws.move(it)
        L, timesL = L_acc + 0, timesLA + 0
        M, timesM = M_acc + 0, timesMA + 0
        indL, indM = indLA, indMA
        hasting = 0
        gibbs = 0
        priorPoi = 0
        forced_reject = False

        if st.const_death_rate:
            sample_shift_mu, b_freq, d_freq = 0, 0.7, 0.8
            fL, fM = st.update_fraction, 1
        else:
            sample_shift_mu, b_freq, d_freq = 0.5, 0.4, 0.8
            fL, fM = st.update_fraction, st.update_fraction

        if r0 < b_freq:
            if r1 < .5 or len(L_acc) == 1:
                L, hasting = lo.update_multiplier_freq(L_acc, *draws.multiplier(it, len(L_acc), fL))
            else:
                draws.times_move(it, len(L_acc))            # A9: consumed, state unchanged
                timesL = np.sort(timesLA + 0.)
                indL = lo.get_rate_index(np.floor(timesL), n_bins)
        elif r0 < d_freq:
            if r1 < .5 or len(M_acc) == 1:
                M, hasting = lo.update_multiplier_freq(M_acc, *draws.multiplier(it, len(M_acc), fM))
            else:
                draws.times_move(it, len(M_acc))
                timesM = np.sort(timesMA + 0.)
                indM = lo.get_rate_index(np.floor(timesM), n_bins)
        elif r0 < 0.999 and st.const_rates == 0:
            L, timesL, M, timesM, hasting, update_L, forced_reject = _rjmcmc(
                draws, it, L_acc, M_acc, timesLA, timesMA, sample_shift_mu, k_max)
            if update_L == 1:
                indL = lo.get_rate_index(np.floor(timesL), n_bins)
            else:
                indM = lo.get_rate_index(np.floor(timesM), n_bins)
            priorPoi = lo.Poisson_prior(len(L), Poi_lambda_rjHP) + lo.Poisson_prior(len(M), Poi_lambda_rjHP)
        else:
            if st.Poisson_HP == 0:
                Poi_lambda_rjHP = draws.gibbs_poi(it, *lo.rj_hp_posterior_params(len(L_acc), len(M_acc)))
            if st.use_rate_HP:
                Gamma_rate = [draws.gibbs_rate(it, 0, *lo.rate_hp_posterior_params(L_acc)),
                              draws.gibbs_rate(it, 1, *lo.rate_hp_posterior_params(M_acc))]
            gibbs = 1

        if (min(abs(np.diff(timesL))) <= lo.MIN_ALLOWED_T or min(abs(np.diff(timesM))) <= lo.MIN_ALLOWED_T
                or forced_reject):
            prior = -np.inf
            lik = -np.inf
        else:
            prior = lo.prior_gamma(L, lo.GAMMA_SHAPE, Gamma_rate[0]) + lo.prior_gamma(M, lo.GAMMA_SHAPE, Gamma_rate[1])
            prior += -np.log(end_time - start_time) * (len(L) - 1 + len(M) - 1)
            if priorPoi != 0:
                prior += priorPoi
            else:
                prior += priorPoiA
                priorPoi = priorPoiA
            lik = lik_fn(L[indL], M[indM]) if gibbs == 0 else likA

        u = draws.accept(it)
        with np.errstate(divide="ignore", invalid="ignore"):
            ok = (lik - likA + prior - priorA + hasting >= np.log(u)) or gibbs == 1
        if ok:
            L_acc, M_acc, timesLA, timesMA = L, M, timesL, timesM
            likA, priorA = lik, prior
            indLA, indMA = indL, indM
            priorPoiA = priorPoi

        if it % s_freq == 0:
            row = [it, likA + priorA, likA, priorA, np.mean(L_acc), np.mean(M_acc),
                   len(L_acc), len(M_acc), start_time, end_time,
                   Gamma_rate[0], Gamma_rate[1], Poi_lambda_rjHP]
            if emp is not None:
                row += list(lo.calculate_r_squared(emp[0], emp[1], L_acc[indLA], M_acc[indMA]))
            out["mcmc"].append(np.array(row, dtype=float))
            out["sp"].append(np.concatenate([L_acc, timesLA[1:len(timesLA) - 1]]))
            out["ex"].append(np.concatenate([M_acc, timesMA[1:len(timesMA) - 1]]))
    out["final"] = (L_acc, M_acc, timesLA, timesMA, likA, priorA)
    return out
