// lr_dd.h - DDRate (diversity-dependent rates) device arithmetic: SURVEY section 8a row A12 and the sampler
// around it (DDRate.py:55-122, 124-241).  Shared by the stand-alone rate kernel (lr_stats.hip) and the
// engine's DD chain step (lr_mcmc.hip), so that both evaluate the very same expressions.
#pragma once
#include "lr_chain.h"

#define LR_DD_NPAR 8          /* [l_max, k, x0, div_0, L, m_max, nuB, nuD] (DD:161) */
#define LR_DD_SMALL 0.000000000000001 /* SMALL_NUMBER, DD:47 */
// draw purposes of the DD sampler in the addressed stream (oracle/dd_mcmc_oracle.py)
#define LR_P_DD_MOVE 16
#define LR_P_DD_SLIDE 17
#define LR_P_DD_MULT 18
#define LR_P_DD_ACCEPT 19

struct lr_dd_params {
    double l_max, k, x0, div_0, L, m_max, nuB, nuD;
};

// likelihood_function's rate half for ONE bin (DD:71-100): x = bin index (TIME_RANGE, lib:255), dt = DT[b].
// get_logistic (DD:55-56) raises its denominator to 1/nu with nu = 1: x ** 1.0 is x, so no pow here.
__device__ __forceinline__ void lr_dd_bin_rates(const lr_dd_params& p, double x, double dt, int m_birth, int m_death,
                                                double* br_, double* dr_, double* niche_, double* frac_) {
    // frac ** nu as exp(nu * log(frac)) with the logarithm shared by the two processes when they see the same niche
    // (a pow call is a log and an exp plus its own special-case handling; 0 and negative fractions behave as in
    // numpy: 0 ** nu = 0, negative ** nu = nan)
    double niche = 1.0, frac = 1.0, br, dr, lfrac = 0.0;
    int niche_model = 0;
    if (m_birth == 0) {
        br = 1.0 * p.l_max;
    } else {
        niche = (m_birth == 1) ? 1.0 * (p.L + p.div_0) : p.div_0 + p.L / (1.0 + exp(-p.k * (x - p.x0)));
        frac = dt / niche;
        lfrac = lr_log(frac), niche_model = m_birth;
        br = p.l_max - p.l_max * exp(p.nuB * lfrac);
        if (br <= 0.0) br = LR_DD_SMALL;
    }
    if (m_death <= 0) {
        dr = 1.0 * p.m_max;
    } else {
        if (m_death != niche_model) {
            niche = (m_death == 1) ? 1.0 * (p.L + p.div_0) : p.div_0 + p.L / (1.0 + exp(-p.k * (x - p.x0)));
            frac = dt / niche;
            lfrac = lr_log(frac);
        }
        dr = p.m_max + p.m_max * exp(p.nuD * lfrac);
        if (dr <= 0.0) dr = LR_DD_SMALL;
    }
    *br_ = br, *dr_ = dr, *niche_ = niche, *frac_ = frac;
}

// lane j < 8 holds parameter j -> wave-uniform struct
__device__ __forceinline__ lr_dd_params lr_dd_unpack(double v) {
    lr_dd_params p;
    p.l_max = lr_bcast(v, 0), p.k = lr_bcast(v, 1), p.x0 = lr_bcast(v, 2), p.div_0 = lr_bcast(v, 3);
    p.L = lr_bcast(v, 4), p.m_max = lr_bcast(v, 5), p.nuB = lr_bcast(v, 6), p.nuD = lr_bcast(v, 7);
    return p;
}

// per-parameter update probability of the vector multiplier move (DD:166-181), lane j < 8
__device__ __forceinline__ double lr_dd_update_freq(int m_birth, int m_death, int lane) {
    //                                l_max k  x0 div_0 L  m_max nuB nuD
    double um = 0.0;
    const int j = lane;
    if (m_birth == 0 && m_death <= 0) um = (j == 0 || j == 5) ? 1.0 : 0.0;
    else if (m_birth == 2 || m_death == 2) um = (j == 2) ? 0.0 : 1.0;
    else um = (j == 0 || j >= 4) ? 1.0 : 0.0;
    if (m_death == -1) um *= (j == 1 || j == 5) ? 0.0 : 1.0;
    if (m_death == -2) um *= (j == 1 || j >= 5) ? 0.0 : 1.0;
    if (j >= LR_DD_NPAR) um = 0.0;
    const double tot = lr_wave_sum(um);
    return um / tot;
}

// calc_prior (DD:110-122): Gamma(1, scale) and standard-normal log-densities; -inf once the midpoint passes PRESENT
__device__ __forceinline__ double lr_dd_prior(double v, double origin, double present, double k0, double log_k0,
                                              int lane) {
    const double LOG10 = 2.302585092994046;           // log(10)
    const double HALF_LOG_2PI = 0.9189385332046727;   // 0.5 log(2 pi)
    double t = 0.0;
    if (lane == 0 || lane == 5) t = (v < 0.0) ? -INFINITY : -v / 10.0 - LOG10;
    if (lane == 3 || lane == 4) t = (v < 0.0) ? -INFINITY : -v / k0 - log_k0;
    if (lane == 1 || lane == 6 || lane == 7) t = -0.5 * v * v - HALF_LOG_2PI;
    if (lane >= LR_DD_NPAR) t = 0.0;
    double p = lr_wave_sum(t);
    if (origin + lr_bcast(v, 2) >= present) p = -INFINITY;
    return p;
}

// Lookup tables (Keiding form, lr_bin_terms model >= 2) straight from per-bin rates given by `rates(b, &br, &dr)`;
// same table formats as lr_build_tables_segments_wave (general double2 entries, or unit-resolution entries `es`
// doubles apart).  Lane l owns bins [l*P, (l+1)*P), P <= LR_DD_MAXP.
#define LR_DD_MAXP 4
template <int CS = 2, class F>
__device__ __forceinline__ void lr_rates_build_tables_wave(F rates, int n_bins, int H, double2* __restrict__ tab, int lane,
                                                           int unit, double fs0, double fe0, int es) {
    double* tabd = reinterpret_cast<double*>(tab);
    auto put_S = [&](int j, double v, double R) { lr_put_S<CS>(tabd, unit, es, j, v, R, fs0); };
    auto put_E = [&](int j, double v, double R) { lr_put_E<CS>(tabd, unit, es, j, v, R, fe0); };
    const int P = (n_bins + LR_WAVE - 1) / LR_WAVE;
    const int b0 = min(lane * P, n_bins), b1 = min(b0 + P, n_bins);
    double br[LR_DD_MAXP], dr[LR_DD_MAXP];
    double sumR = 0.0;
#pragma unroll
    for (int i = 0; i < LR_DD_MAXP; ++i) {
        const int b = b0 + i;
        br[i] = 1.0, dr[i] = 1.0;
        if (i < P && b < b1) {
            rates(b, &br[i], &dr[i]);
            sumR += br[i] + dr[i];
        }
    }
    double totR;
    double cum = lr_wave_exclusive_scan(sumR, lane, &totR);
#pragma unroll
    for (int i = 0; i < LR_DD_MAXP; ++i) {
        const int b = b0 + i;
        if (i < P && b < b1) {
            const double logB = lr_log(br[i]), logD = lr_log(dr[i]), R = br[i] + dr[i];
            put_S(b + 1, logB + cum, R);
            put_E(H + b + 1, logD - cum, R);
            cum += R;
        }
    }
    if (lane == 0) {
        put_S(0, 0.0, 0.0), put_E(H, 0.0, 0.0);
        put_S(n_bins + 1, totR, 0.0), put_E(H + n_bins + 1, -totR, 0.0);
    }
}

template <int CS = 2>
__device__ inline void lr_dd_build_tables_wave(const lr_dd_params& p, const double* __restrict__ DT, int m_birth,
                                               int m_death, int n_bins, int H, double2* __restrict__ tab, int lane,
                                               int unit, double fs0, double fe0, int es) {
    lr_rates_build_tables_wave<CS>(
        [&](int b, double* br, double* dr) {
            double ni, fr;
            lr_dd_bin_rates(p, (double)b, DT[b], m_birth, m_death, br, dr, &ni, &fr);
        },
        n_bins, H, tab, lane, unit, fs0, fe0, es);
}

// ---- trend_rate.py: rates driven by a per-bin covariate ---------------------------------------
#define LR_TR_NPAR 6          /* [l_min, m_min, alpha, beta, delta, gamma] (trend_rate.py:74) */
#define LR_P_TR_MOVE 32
#define LR_P_TR_NORM 33
#define LR_P_TR_MULT 34
#define LR_P_TR_ACCEPT 35

struct lr_trend_params {
    double l_min, m_min, alpha, beta, delta, gamma;
};

// likelihood_function's rate half for ONE bin (trend_rate.py:73-88); t = TREND[b]
__device__ __forceinline__ void lr_trend_bin_rates(const lr_trend_params& p, double t, int const_birth, int const_death,
                                                   double* br_, double* dr_) {
    // TREND ** exponent as exp(exponent * log(TREND)), one logarithm for both processes (TREND is in (0, 1])
    double br = 1.0 * p.l_min, dr = 1.0 * p.m_min;
    const double lt = (const_birth && const_death) ? 0.0 : lr_log(t);
    if (!const_birth) {
        br = p.l_min + p.alpha * exp(p.delta * lt);
        if (br <= 0.0) br = LR_DD_SMALL;
    }
    if (!const_death) {
        dr = p.m_min + p.beta * exp(p.gamma * lt);
        if (dr <= 0.0) dr = LR_DD_SMALL;
    }
    *br_ = br, *dr_ = dr;
}

__device__ __forceinline__ lr_trend_params lr_trend_unpack(double v) {
    lr_trend_params p;
    p.l_min = lr_bcast(v, 0), p.m_min = lr_bcast(v, 1), p.alpha = lr_bcast(v, 2), p.beta = lr_bcast(v, 3);
    p.delta = lr_bcast(v, 4), p.gamma = lr_bcast(v, 5);
    return p;
}

// update probabilities of the two vector moves (trend_rate.py:122-135), lane j < 6; 0/0 = nan when a mask is empty,
// as in the reference (a nan probability never fires)
__device__ __forceinline__ void lr_trend_update_freq(int const_birth, int const_death, int lane, double* f_mult,
                                                     double* f_norm) {
    double um = (lane == 0 || lane == 1 || lane == 4 || lane == 5) ? 1.0 : 0.0;
    double un = (lane == 2 || lane == 3) ? 1.0 : 0.0;
    if (const_birth && lane == 4) um = 0.0;
    if (const_birth && lane == 2) un = 0.0;
    if (const_death && lane == 5) um = 0.0;
    if (const_death && lane == 3) un = 0.0;
    if (lane >= LR_TR_NPAR) um = 0.0, un = 0.0;
    *f_mult = um / lr_wave_sum(um);
    *f_norm = un / lr_wave_sum(un);
}

// calc_prior (trend_rate.py:93-100): Gamma(1, scale 10, loc .001) on the floors, Normal(0, 5) on the slopes,
// Gamma(3, scale .5) on the exponents; one packed log
__device__ __forceinline__ double lr_trend_prior(double v, int lane) {
    const double LOG10 = 2.302585092994046, HALF_LOG_2PI = 0.9189385332046727, LOG5 = 1.6094379124341003;
    const double LOG2 = 0.6931471805599453;
    const double y = (lane <= 1) ? (v - .001) / 10.0 : (v / .5);
    const double ly = lr_log((lane == 4 || lane == 5) && y > 0.0 ? y : 1.0);
    double t = 0.0;
    if (lane <= 1) t = (y < 0.0) ? -INFINITY : -y - LOG10;
    if (lane == 2 || lane == 3) t = -0.5 * (v / 5.0) * (v / 5.0) - HALF_LOG_2PI - LOG5;
    if (lane == 4 || lane == 5) t = (y <= 0.0) ? -INFINITY : 2.0 * ly - y - LOG2 + LOG2;   // - lgamma(3) - log(.5)
    if (lane >= LR_TR_NPAR) t = 0.0;
    return lr_wave_sum(t);
}
