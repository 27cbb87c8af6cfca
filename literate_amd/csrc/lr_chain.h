// lr_chain.h - wave-level proposal / prior arithmetic of one chain.
//
// One wave owns one chain; lane j holds element j of each state vector (rate j, shift time j),
// scalars are wave-uniform.  The same functions serve the explicit-draw scorer
// (lr_rj_propose_score, checked against reference-generated vectors) and the fused engine.
#pragma once
#include "lr_device.h"

#define LR_SHAPE_BETA_RJ 10.0   /* LRF:586 */
#define LR_MIN_ALLOWED_T 1.0    /* LRF:587 */
#define LR_GAMMA_SHAPE 2.0      /* LRF:588 */
#define LR_HP_GAMMA_SHAPE 1.2   /* LRF:589 */
#define LR_HP_GAMMA_RATE 0.1    /* LRF:590 */
#define LR_RJHP_SHAPE 2.0       /* LRF:101 */
#define LR_RJHP_RATE 1.0        /* LRF:102 */
#define LR_MULT_D 1.1           /* LRF:165 */

// scipy.stats.beta.logpdf(x, a, a), a = 10 (LRF:22-23)
__device__ __forceinline__ double lr_log_beta_sym_pdf(double x) {
    const double a = LR_SHAPE_BETA_RJ;
    const double betaln = 2.0 * lgamma(a) - lgamma(2.0 * a);
    return (a - 1.0) * log1p(-x) + (a - 1.0) * log(x) - betaln;
}

// update_multiplier_freq (LRF:165-176): lane j < K scales its rate by exp(2 log d (u-.5)) if ff
__device__ __forceinline__ double lr_wave_multiplier(double& R, int K, bool ff, double u, double d, int lane) {
    const double l = 2.0 * log(d);
    double m = exp(l * (u - .5));
    if (!ff) m = 1.0;
    const bool active = lane < K;
    if (active) R = R * m;
    return lr_wave_sum(active ? log(m) : 0.0);
}

// add_shift_RJ_weighted_mean (LRF:29-47).  ind = interval, delta = offset inside it, u ~ Beta(10,10).
__device__ __forceinline__ double lr_wave_add_shift(double& R, double& T, int& K, int ind, double delta, double u,
                                                    int lane) {
    const double t_i1 = __shfl(T, ind, LR_WAVE);
    const double t_i2 = __shfl(T, ind + 1, LR_WAVE);
    const double rate_i = __shfl(R, ind, LR_WAVE);
    const double Tup = __shfl_up(T, 1, LR_WAVE);
    const double Rup = __shfl_up(R, 1, LR_WAVE);
    const double r_time = t_i2 - t_i1;
    const double t_prime = t_i1 + delta;
    const double p1 = (t_i1 - t_prime) / (t_i1 - t_i2);
    const double p2 = (t_prime - t_i2) / (t_i1 - t_i2);
    const double logit = log((1 - u) / u);
    const double r1 = exp(log(rate_i) - p2 * logit);
    const double r2 = exp(log(rate_i) + p1 * logit);
    // sorted insert of t_prime in [t_i1, t_i2): position ind+1
    if (lane == ind + 1) T = t_prime;
    else if (lane > ind + 1) T = Tup;
    if (lane == ind) R = r1;
    else if (lane == ind + 1) R = r2;
    else if (lane > ind + 1) R = Rup;
    K += 1;
    const double log_q = log(fabs(r_time)) - lr_log_beta_sym_pdf(u);
    const double jac = 2 * log(r1 + r2) - log(rate_i);
    return log_q + jac;
}

// remove_shift_RJ_weighted_mean (LRF:49-69).  idx = removed shift, 1..K-1.  The reference deletes
// by VALUE (LRF:56, 63); that equals deletion by index unless two rates / times are bit-identical.
__device__ __forceinline__ double lr_wave_remove_shift(double& R, double& T, int& K, int idx, int lane) {
    const double t_prime = __shfl(T, idx, LR_WAVE);
    const double t_i1 = __shfl(T, idx - 1, LR_WAVE);
    const double t_i2 = __shfl(T, idx + 1, LR_WAVE);
    const double r1 = __shfl(R, idx - 1, LR_WAVE);
    const double r2 = __shfl(R, idx, LR_WAVE);
    const double Tdn = __shfl_down(T, 1, LR_WAVE);
    const double Rdn = __shfl_down(R, 1, LR_WAVE);
    const double dT = fabs(t_i2 - t_i1);
    const double p1 = (t_i1 - t_prime) / (t_i1 - t_i2);
    const double p2 = (t_prime - t_i2) / (t_i1 - t_i2);
    const double rate_prime = exp(p1 * log(r1) + p2 * log(r2));
    if (lane >= idx) T = Tdn;
    if (lane == idx - 1) R = rate_prime;
    else if (lane >= idx) R = Rdn;
    K -= 1;
    const double u = 1. / (1 + r2 / r1);
    const double log_q = -log(dT) + lr_log_beta_sym_pdf(u);
    const double jac = log(rate_prime) - (2 * log(r1 + r2));
    return log_q + jac;
}

// prior_gamma (LRF:201-202) the way scipy evaluates it: y = x/scale; (a-1) log y - y - lgamma(a) - log scale
__device__ __forceinline__ double lr_wave_prior_gamma(double R, int K, double a, double b, int lane) {
    const double scale = 1. / b;
    const double y = R / scale;
    const double v = (a - 1.0) * log(y) - y - lgamma(a) - log(scale);
    return lr_wave_sum(lane < K ? v : 0.0);
}

// Poisson_prior (LRF:198-199): k log(rate) - rate - sum_{i<=k} log i, k = number of rates
__device__ __forceinline__ double lr_wave_poisson_prior(int k, double rate, int lane) {
    const double lf = lr_wave_sum((lane >= 1 && lane <= k) ? log((double)lane) : 0.0);
    return k * log(rate) - rate - lf;
}

// integer bin edge of shift time j relative to the first one: floor (LRF:262 etc.) or round (LRF:129)
__device__ __forceinline__ int lr_wave_edges(double T, int mode) {
    const double e = mode ? rint(T) : floor(T);
    const double e0 = __shfl(e, 0, LR_WAVE);
    return (int)(e - e0);
}

// min_j |T[j+1]-T[j]| over j < K (guard of LRF:290)
__device__ __forceinline__ double lr_wave_min_segment(double T, int K, int lane) {
    const double Tn = __shfl_down(T, 1, LR_WAVE);
    return lr_wave_min(lane < K ? fabs(Tn - T) : 1e300);
}
