// lr_chain.h - wave-level proposal / prior arithmetic of one chain.
//
// One wave owns one chain; lane j holds element j of each state vector (rate j, shift time j),
// scalars are wave-uniform.  The same functions serve the explicit-draw scorer
// (lr_rj_propose_score, checked against reference-generated vectors) and the fused engine.
//
// The chain step is latency bound (one wave, a serial chain of fp64 transcendentals), so
// independent scalar log/exp evaluations are packed into different lanes of ONE call and
// broadcast back, and the rejection loop of the Gamma sampler is evaluated for 32 attempts at
// once (lane a = attempt a; the first accepted attempt wins, exactly as the serial loop).
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
#define LR_BETALN_10_10 (-13.736229227036555) /* scipy.special.betaln(10, 10) */

// broadcast of lane `src` (wave-uniform index) through v_readlane: no trip through the LDS crossbar
__device__ __forceinline__ double lr_bcast(double v, int src) { return lr_readlane_f64(v, src); }
__device__ __forceinline__ int lr_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// log(k!) summed the way the reference does, np.sum(np.log(np.arange(1, k+1))) (LRF:199)
__device__ __forceinline__ double lr_log_factorial(int k) {
    static const double T[LR_KMAX + 2] = {
        0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479453, 4.787491742782046, 6.579251212010101,
        8.525161361065415, 10.604602902745249, 12.801827480081469, 15.104412573075514, 17.502307845873887,
        19.98721449566189, 22.552163853123425, 25.191221182738683, 27.899271383840894, 30.671860106080672,
        33.50507345013689, 36.39544520803305, 39.339884187199495, 42.335616460753485, 45.38013889847691,
        48.47118135183523, 51.60667556776438, 54.78472939811232, 58.00360522298052, 61.261701761002,
        64.55753862700632, 67.88974313718153, 71.257038967168, 74.65823634883016, 78.0922235533153,
        81.55795945611504, 85.05446701758153};
    return T[k];
}

// update_multiplier_freq (LRF:165-176): lane j < K scales its rate by m = exp(2 log d (u-.5)) if ff.
// Hastings term = sum log m (LRF:175); log(exp(x)) is taken as x (differs by < 1 ulp of x).
__device__ __forceinline__ double lr_wave_multiplier(double& R, int K, bool ff, double u, double l, int lane) {
    const bool active = (lane < K) && ff;
    const double x = active ? l * (u - .5) : 0.0;
    const double m = exp(x);
    if (active) R = R * m;
    return lr_wave_sum(x);
}

// add_shift_RJ_weighted_mean (LRF:29-47).  ind = interval, delta = offset inside it, u ~ Beta(10,10).
__device__ __forceinline__ double lr_wave_add_shift(double& R, double& T, int& K, int ind, double delta, double u,
                                                    int lane) {
    const double t_i1 = lr_bcast(T, ind);
    const double t_i2 = lr_bcast(T, ind + 1);
    const double rate_i = lr_bcast(R, ind);
    const double Tup = lr_dpp_zero<0x138 /* wave_shr:1 */, 0xf, 0xf>(T);
    const double Rup = lr_dpp_zero<0x138, 0xf, 0xf>(R);
    const double r_time = t_i2 - t_i1;
    const double t_prime = t_i1 + delta;
    const double p1 = (t_i1 - t_prime) / (t_i1 - t_i2);
    const double p2 = (t_prime - t_i2) / (t_i1 - t_i2);
    // one packed log: lane0 (1-u)/u, lane1 rate_i, lane2 |r_time|, lane3 u, lane4 1-u
    double x = 1.0;
    if (lane == 0) x = (1 - u) / u;
    if (lane == 1) x = rate_i;
    if (lane == 2) x = fabs(r_time);
    if (lane == 3) x = u;
    if (lane == 4) x = 1 - u;
    const double lx = lr_log(x);
    const double logit = lr_bcast(lx, 0), log_rate = lr_bcast(lx, 1), log_rt = lr_bcast(lx, 2), log_u = lr_bcast(lx, 3);
    const double log_1mu = lr_bcast(lx, 4);
    // one packed exp: lane0 -> r1, lane1 -> r2
    const double ex = exp(lane == 0 ? log_rate - p2 * logit : (lane == 1 ? log_rate + p1 * logit : 0.0));
    const double r1 = lr_bcast(ex, 0), r2 = lr_bcast(ex, 1);
    // sorted insert of t_prime in [t_i1, t_i2): position ind+1
    if (lane == ind + 1) T = t_prime;
    else if (lane > ind + 1) T = Tup;
    if (lane == ind) R = r1;
    else if (lane == ind + 1) R = r2;
    else if (lane > ind + 1) R = Rup;
    K += 1;
    const double a = LR_SHAPE_BETA_RJ;
    const double log_beta = (a - 1.0) * log_1mu + (a - 1.0) * log_u - LR_BETALN_10_10;   // LRF:22-23
    const double log_q = log_rt - log_beta;
    const double jac = 2 * lr_log(r1 + r2) - log_rate;
    return log_q + jac;
}

// remove_shift_RJ_weighted_mean (LRF:49-69).  idx = removed shift, 1..K-1.  The reference deletes
// by VALUE (LRF:56, 63); that equals deletion by index unless two rates / times are bit-identical.
__device__ __forceinline__ double lr_wave_remove_shift(double& R, double& T, int& K, int idx, int lane) {
    const double t_prime = lr_bcast(T, idx);
    const double t_i1 = lr_bcast(T, idx - 1);
    const double t_i2 = lr_bcast(T, idx + 1);
    const double r1 = lr_bcast(R, idx - 1);
    const double r2 = lr_bcast(R, idx);
    const double Tdn = lr_dpp_zero<0x130 /* wave_shl:1 */, 0xf, 0xf>(T);
    const double Rdn = lr_dpp_zero<0x130, 0xf, 0xf>(R);
    const double dT = fabs(t_i2 - t_i1);
    const double p1 = (t_i1 - t_prime) / (t_i1 - t_i2);
    const double p2 = (t_prime - t_i2) / (t_i1 - t_i2);
    const double u = 1. / (1 + r2 / r1);
    // packed log: lane0 r1, lane1 r2, lane2 dT, lane3 r1+r2, lane4 u, lane5 1-u
    double x = 1.0;
    if (lane == 0) x = r1;
    if (lane == 1) x = r2;
    if (lane == 2) x = dT;
    if (lane == 3) x = r1 + r2;
    if (lane == 4) x = u;
    if (lane == 5) x = 1 - u;
    const double lx = lr_log(x);
    const double l1 = lr_bcast(lx, 0), l2 = lr_bcast(lx, 1), log_dT = lr_bcast(lx, 2), log_sum = lr_bcast(lx, 3);
    const double log_u = lr_bcast(lx, 4), log_1mu = lr_bcast(lx, 5);
    const double rate_prime = exp(p1 * l1 + p2 * l2);
    if (lane >= idx) T = Tdn;
    if (lane == idx - 1) R = rate_prime;
    else if (lane >= idx) R = Rdn;
    K -= 1;
    const double a = LR_SHAPE_BETA_RJ;
    const double log_beta = (a - 1.0) * log_1mu + (a - 1.0) * log_u - LR_BETALN_10_10;
    const double log_q = -log_dT + log_beta;
    const double jac = lr_log(rate_prime) - (2 * log_sum);
    return log_q + jac;
}

// prior_gamma (LRF:201-202), general shape, the way scipy evaluates it:
// y = x/scale; (a-1) log y - y - lgamma(a) - log scale
__device__ __forceinline__ double lr_wave_prior_gamma(double R, int K, double a, double b, int lane) {
    const double scale = 1. / b;
    const double y = R / scale;
    const double v = (a - 1.0) * lr_log(y) - y - lgamma(a) - lr_log(scale);
    return lr_wave_sum(lane < K ? v : 0.0);
}

// same for shape 2 (lgamma(2) = 0) with log R and log b already known: no transcendental left
__device__ __forceinline__ double lr_wave_prior_gamma2(double R, double logR, int K, double b, double logb, int lane) {
    const double v = (logR + logb) - R * b + logb;
    return lr_wave_sum(lane < K ? v : 0.0);
}

// Poisson_prior (LRF:198-199): k log(rate) - rate - log k!, k = number of rates
__device__ __forceinline__ double lr_poisson_prior(int k, double rate, double log_rate) {
    return k * log_rate - rate - lr_log_factorial(k);
}
__device__ __forceinline__ double lr_wave_poisson_prior(int k, double rate, int lane) {
    const double lf = lr_wave_sum((lane >= 1 && lane <= k) ? lr_log((double)lane) : 0.0);
    return k * lr_log(rate) - rate - lf;
}

// integer bin edge of shift time j relative to the first one: floor (LRF:262 etc.) or round (LRF:129)
__device__ __forceinline__ int lr_wave_edges(double T, int mode) {
    const double e = mode ? rint(T) : floor(T);
    const double e0 = lr_bcast(e, 0);
    return (int)(e - e0);
}

// min_j |T[j+1]-T[j]| over j < K (guard of LRF:290)
__device__ __forceinline__ double lr_wave_min_segment(double T, int K, int lane) {
    const double Tn = lr_dpp_zero<0x130 /* wave_shl:1 */, 0xf, 0xf>(T);
    return lr_wave_min(lane < K ? fabs(Tn - T) : 1e300);
}

// Two standard Gamma variates at once (shapes >= 1): lanes 0..31 evaluate attempts 0..31 of the first,
// lanes 32..63 of the second; identical to lr_gamma(.., base 0, ..) for each (first accepted attempt).
__device__ __forceinline__ void lr_wave_gamma2(const lr_stream& s, uint64_t it, uint32_t purpose_a, double shape_a,
                                               uint32_t purpose_b, double shape_b, int lane, double* ga, double* gb) {
    const bool hi = lane >= 32;
    const int a = lane & 31;
    const uint32_t purpose = hi ? purpose_b : purpose_a;
    const double shape = hi ? shape_b : shape_a;
    const double d = shape - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    const double x = lr_normal(s, it, purpose, 2 * a);
    const double t = 1.0 + c * x;
    const double v = t * t * t;
    const double u = lr_pair(s, it, purpose, 2 * a + 1).a;
    const double val0 = d * v;
    // Marsaglia-Tsang squeeze: u < 1 - 0.0331 x^4 implies the log test below.  When attempt 0 of BOTH variates
    // passes it (~85 % of the calls) the two logarithms are not needed at all.
    const double x2 = x * x;
    const unsigned long long sq = __ballot(v > 0.0 && u < 1.0 - 0.0331 * x2 * x2);
    if ((sq & 1ull) && (sq & (1ull << 32))) {
        *ga = lr_bcast(val0, 0), *gb = lr_bcast(val0, 32);
        return;
    }
    bool ok = false;
    if (v > 0.0) ok = (u <= 0.0) || (lr_log(u) < 0.5 * x * x + d - d * v + d * lr_log(v));
    const unsigned long long m = __ballot(ok);
    const unsigned lo = (unsigned)(m & 0xffffffffull), hi_m = (unsigned)(m >> 32);
    const double val = d * v;
    const double ra = lo ? lr_bcast(val, __ffs(lo) - 1) : lr_bcast(d, 0);
    const double rb = hi_m ? lr_bcast(val, 32 + __ffs(hi_m) - 1) : lr_bcast(d, 32);
    *ga = ra, *gb = rb;
}
