"""Restatement of the trend_rate.py Metropolis-Hastings driver, __main__ (trend_rate.py:102-196).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Statement by statement: one uniform picks the move (33 %: additive
normal step d = 0.001 on the slopes alpha / beta, update_normal_nobound_vec lib:140-146; else the vector multiplier on
l_min, m_min and the exponents, lib:156-165), likelihood_function (trend_rate.py:72-91), calc_prior (:93-100), `>`
in the acceptance rule and the unconditional acceptance of iteration 0 (:176).  Draw sources as in mcmc_oracle.
"""
import numpy as np

from . import literate_oracle as lo
from . import philox as px

P_TR_MOVE, P_TR_NORM, P_TR_MULT, P_TR_ACCEPT = 32, 33, 34, 35


class NumpyLegacyDraws:
    """np.random.* in trend_rate.py's call order (:165, lib:142-143 / lib:158-159, :176)."""

    def move(self, it):
        return np.random.random(1)[0]

    def normal_vec(self, it, f, d):
        ff = np.random.binomial(1, f, np.shape(f))
        m = np.random.normal(0, d, np.shape(f))
        return ff, m

    def multiplier(self, it, f):
        ff = np.random.binomial(1, f, np.shape(f))
        u = np.random.uniform(0, 1, np.shape(f))
        return ff, u

    def accept(self, it):
        return np.random.random()


class PhiloxDraws:
    def __init__(self, seed, chain):
        self.s = px.Stream(seed, chain)

    def move(self, it):
        return self.s.pair(it, P_TR_MOVE, 0)[0]

    def normal_vec(self, it, f, d):
        n = len(f)
        ff, m = np.zeros(n, dtype=int), np.zeros(n)
        for j in range(n):
            ff[j] = 1 if self.s.pair(it, P_TR_MULT, j)[0] < f[j] else 0
            m[j] = self.s.normal(it, P_TR_NORM, j) * d
        return ff, m

    def multiplier(self, it, f):
        n = len(f)
        ff, u = np.zeros(n, dtype=int), np.zeros(n)
        for j in range(n):
            ub, uu = self.s.pair(it, P_TR_MULT, j)
            ff[j] = 1 if ub < f[j] else 0
            u[j] = uu
        return ff, u

    def accept(self, it):
        return self.s.pair(it, P_TR_ACCEPT, 0)[0]


def update_masks(const_birth, const_death):
    """trend_rate.py:122-135: (multiplier mask, normal mask), each normalised to sum 1."""
    um = np.array([1, 1, 0, 0, 1, 1], dtype=float)
    un = np.array([0, 0, 1, 1, 0, 0], dtype=float)
    if const_birth:
        um[4], un[2] = 0, 0
    if const_death:
        um[5], un[3] = 0, 0
    with np.errstate(all="ignore"):
        return um / np.sum(um), un / np.sum(un)


def run_trend_mcmc(N_SPEC, N_EXTI, DT, TREND, draws, n_iterations, s_freq, const_birth=False, const_death=False,
                   emp=None, lik_fn=None):
    """Rows as trend_rate.py writes them (:183-189): [it, posterior, lik, lik_birth, lik_death, prior, args[6],
    birth[n], death[n] (+ adequacy[3])]."""
    if lik_fn is None:
        def lik_fn(a):
            return lo.trend_likelihood_function(a, N_SPEC, N_EXTI, DT, TREND, const_birth, const_death)
    um, un = update_masks(const_birth, const_death)
    argsA = np.array([.1, .1, 0, 0, 1, 1], dtype=float)
    with np.errstate(all="ignore"):
        lik_res = lik_fn(argsA)
    likA = np.sum(lik_res[0])
    likBirthA, likDeathA = lik_res[0][0], lik_res[0][1]
    birth, death = lik_res[1], lik_res[2]
    priorA = lo.trend_calc_prior(argsA)
    rows = []
    for it in range(n_iterations):
        args = argsA + 0.
        rr = draws.move(it)
        if rr < .33:
            ff, m = draws.normal_vec(it, un, .001)
            m = np.array(m, dtype=float)
            m[ff == 0] = 0.
            args, hastings = args + m, 0
        else:
            ff, u = draws.multiplier(it, um)
            m = np.exp(2 * np.log(1.1) * (u - .5))
            m[ff == 0] = 1.
            args, hastings = args * m, np.sum(np.log(m))
        with np.errstate(all="ignore"):
            lik_res = lik_fn(args)
            lik = np.sum(lik_res[0])
            prior = lo.trend_calc_prior(args)
            u_acc = draws.accept(it)
            ok = ((lik - likA) + (prior - priorA) + hastings > np.log(u_acc)) or it == 0
        if ok:
            argsA, priorA, likA = args, prior, lik
            likBirthA, likDeathA = lik_res[0][0], lik_res[0][1]
            birth, death = lik_res[1], lik_res[2]
        if it % s_freq == 0:
            row = [it, likA + priorA, likA, likBirthA, likDeathA, priorA] + list(argsA) + list(birth) + list(death)
            if emp is not None:
                with np.errstate(all="ignore"):
                    row += list(lo.calculate_r_squared(emp[0], emp[1], birth, death))
            rows.append(np.array(row, dtype=float))
    return rows
