"""Restatement of the DDRate.py Metropolis-Hastings driver, __main__ (DDRate.py:124-241).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The loop body follows the reference statement by
statement: one sliding-window move on the logistic midpoint in 10 % of the iterations of the niche models
(DD:198-203), otherwise a vector multiplier move over the parameters the model frees (DD:166-181, 205);
likelihood_function (DD:71-107) and calc_prior (DD:110-122); `>` in the acceptance rule and the
unconditional acceptance of iteration 0 (DD:211).  Randomness is pluggable like in mcmc_oracle:
``NumpyLegacyDraws`` replays numpy's global stream in the reference's call order (pins the restatement on
reference runs), ``PhiloxDraws`` is the device's addressed stream.
"""
import numpy as np

from . import literate_oracle as lo
from . import philox as px

# draw purposes of the DD sampler in the addressed stream (distinct from the RJ sampler's, oracle/philox.py)
P_DD_MOVE, P_DD_SLIDE, P_DD_MULT, P_DD_ACCEPT = 16, 17, 18, 19


class NumpyLegacyDraws:
    """np.random.* in DDRate.py's call order (DD:197, lib:125, lib:140, lib:158-159, DD:211)."""

    def move(self, it):
        rr = np.random.random(2)
        return rr[0], rr[1]

    def slide(self, it):
        return np.random.random()

    def normal(self, it, d):
        return np.random.normal(0, d)

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
        return self.s.pair(it, P_DD_MOVE, 0)

    def slide(self, it):
        return self.s.pair(it, P_DD_SLIDE, 0)[0]

    def normal(self, it, d):
        return self.s.normal(it, P_DD_SLIDE, 1) * d

    def multiplier(self, it, f):
        n = len(f)
        ff, u = np.zeros(n, dtype=int), np.zeros(n)
        for j in range(n):
            ub, uu = self.s.pair(it, P_DD_MULT, j)
            ff[j] = 1 if ub < f[j] else 0
            u[j] = uu
        return ff, u

    def accept(self, it):
        return self.s.pair(it, P_DD_ACCEPT, 0)[0]


def update_multiplier_mask(m_birth, m_death):
    """Per-parameter update probabilities of the vector multiplier move (DD:166-181)."""
    if m_birth == 0 and m_death <= 0:
        um = np.array([1., 0, 0, 0, 0, 1, 0, 0])
    elif m_birth == 2 or m_death == 2:
        um = np.array([1., 1, 0, 1, 1, 1, 1, 1])
    else:
        um = np.array([1., 0, 0, 0, 1, 1, 1, 1])
    if m_death == -1:
        um = um * np.array([1., 0, 1, 1, 1, 0, 1, 1])
    if m_death == -2:
        um = um * np.array([1., 0, 1, 1, 1, 0, 0, 0])
    return um / np.sum(um)


def initial_args(origin, present, init_death=0.1):
    """DD:151-161: [l_max, k, x0, div_0, L, m_max, nuB, nuD]."""
    return np.array([0.5, 1.5, present - np.mean([origin, present]), 10, 20000, init_death, 1., 1.])


def run_dd_mcmc(N_SPEC, N_EXTI, DT, time_range, origin, present, m_birth, m_death, draws, n_iterations, s_freq,
                init_death=0.1, emp=None, lik_fn=None):
    """Returns the rows DDRate.py writes (DD:219-235) without the per-bin columns' formatting: each row is
    [it, posterior, lik, lik_birth, lik_death, prior, args_logged[8], birth[n], death[n], niche[n], frac[n]
     (+ adequacy[3] when emp=(B_EMP, D_EMP))]."""
    prior_k0_l = np.max(DT)
    if lik_fn is None:
        def lik_fn(a):
            return lo.dd_likelihood_function(a, N_SPEC, N_EXTI, DT, time_range, m_birth, m_death)
    um = update_multiplier_mask(m_birth, m_death)
    argsA = initial_args(origin, present, init_death)
    with np.errstate(all="ignore"):
        lik_res = lik_fn(argsA)
    likA = np.sum(lik_res[0])
    likBirthA, likDeathA = lik_res[0][0], lik_res[0][1]
    birth, death, niche, frac = lik_res[1], lik_res[2], lik_res[3], lik_res[4]
    priorA = lo.dd_calc_prior(argsA, prior_k0_l, origin, present)
    rows = []
    for it in range(n_iterations):
        args = argsA + 0.
        hastings = 0
        rr0, rr1 = draws.move(it)
        if rr1 < 0.1 and (m_birth == 2 or m_death == 2):
            res = argsA + 0
            ii = res[2] + (draws.slide(it) - .5) * 1.5               # update_sliding_win(m=0, M=PRESENT, d=1.5), lib:124-128
            if ii > present:
                ii = present - (ii - present)
            res[2] = abs(ii)
            if m_death == -1:
                res[1] = res[1] + draws.normal(it, 0.2)                # update_normal_nobound, lib:136-138
            args, hastings = res, 0
        else:
            ff, u = draws.multiplier(it, um)
            m = np.exp(2 * np.log(1.1) * (u - .5))
            m[ff == 0] = 1.
            args = args * m
            hastings = np.sum(np.log(m))
        with np.errstate(all="ignore"):
            lik_res = lik_fn(args)
            lik = np.sum(lik_res[0])
            prior = lo.dd_calc_prior(args, prior_k0_l, origin, present)
            u_acc = draws.accept(it)
            ok = ((lik - likA) + (prior - priorA) + hastings > np.log(u_acc)) or it == 0
        if ok:
            argsA, priorA, likA = args, prior, lik
            likBirthA, likDeathA = lik_res[0][0], lik_res[0][1]
            birth, death, niche, frac = lik_res[1], lik_res[2], lik_res[3], lik_res[4]
        if it % s_freq == 0:
            argsO = np.array(argsA, dtype=float)
            argsO[2] += origin
            argsO[4] += argsO[3]
            row = [it, likA + priorA, likA, likBirthA, likDeathA, priorA] + list(argsO) + list(birth) + list(death) \
                + list(niche) + list(frac)
            if emp is not None:
                with np.errstate(all="ignore"):
                    row += list(lo.calculate_r_squared(emp[0], emp[1], birth, death))
            rows.append(np.array(row, dtype=float))
    return rows
