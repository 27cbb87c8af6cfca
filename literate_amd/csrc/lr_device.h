// lr_device.h - device-side building blocks shared by the LiteRate HIP kernels (gfx950, wave64).
//
//  * wave-level reductions / scans with a fixed association order (bitwise reproducible);
//  * Philox4x32-10 addressed draws, Box-Muller normals, Marsaglia-Tsang gamma variates
//    (restated on the CPU in oracle/philox.py, purposes must match);
//  * the per-chain table builder that turns per-bin rates into the two lookup tables the
//    lineage scan gathers from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/literate_hip.h"
#include "lr_math.h"

#define LR_WAVE 64

// ---------------------------------------------------------------------------------------
// wave helpers on DPP (data-parallel primitives: the cross-lane move happens inside the VALU
// instruction, no trip through the LDS crossbar as ds_bpermute / __shfl would make).  Fixed
// association order, so results are bitwise reproducible; every lane gets the same total.
// Scan pattern (AMD GCN cross-lane ops): row_shr 1,2,3 of the input, row_shr 4 / 8 of the
// running value under bank masks, then row_bcast15 / row_bcast31 under row masks.
// ALL 64 lanes must be active.
// ---------------------------------------------------------------------------------------
#define LR_DPP_ROW_SHR(n) (0x110 | (n))
#define LR_DPP_ROW_BCAST15 0x142
#define LR_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double lr_dpp_zero(double src) {  // src moved by DPP, 0 where masked / out of range
    int lo = __double2loint(src), hi = __double2hiint(src);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double lr_dpp_self(double src) {  // src moved by DPP, own value where masked / out of range
    int lo = __double2loint(src), hi = __double2hiint(src);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, BANK_MASK, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ long long lr_dpp_zero_i64(long long src) {
    int lo = (int)(src & 0xffffffffll), hi = (int)(src >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double lr_wave_inclusive_scan(double x) {
    double v = x;
    v += lr_dpp_zero<LR_DPP_ROW_SHR(1), 0xf, 0xf>(x);
    v += lr_dpp_zero<LR_DPP_ROW_SHR(2), 0xf, 0xf>(x);
    v += lr_dpp_zero<LR_DPP_ROW_SHR(3), 0xf, 0xf>(x);
    v += lr_dpp_zero<LR_DPP_ROW_SHR(4), 0xf, 0xe>(v);
    v += lr_dpp_zero<LR_DPP_ROW_SHR(8), 0xf, 0xc>(v);
    v += lr_dpp_zero<LR_DPP_ROW_BCAST15, 0xa, 0xf>(v);
    v += lr_dpp_zero<LR_DPP_ROW_BCAST31, 0xc, 0xf>(v);
    return v;
}
// Every lane l gets v[l & 31] (lr_half_lo) / v[32 + (l & 31)] (lr_half_hi): gfx950's v_permlane32_swap exchanges the
// upper 32 lanes of one register with the lower 32 of another inside the VALU - no trip through the LDS crossbar as a
// ds_bpermute (__shfl) would make, and the chain step is one wave's instruction stream.
__device__ __forceinline__ double lr_half_lo(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ double lr_half_hi(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double lr_readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double lr_wave_sum(double v) { return lr_readlane_f64(lr_wave_inclusive_scan(v), 63); }
__device__ __forceinline__ double lr_wave_min(double x) {
    double v = x;
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_SHR(1), 0xf, 0xf>(v));
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_SHR(2), 0xf, 0xf>(v));
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_SHR(4), 0xf, 0xf>(v));
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_SHR(8), 0xf, 0xf>(v));     // lane 15 of each row: row minimum
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_BCAST15, 0xa, 0xf>(v));    // lane 31 / 63: min of rows 0-1 / 2-3
    v = fmin(v, lr_dpp_self<LR_DPP_ROW_BCAST31, 0xc, 0xf>(v));    // lane 63: min of all
    return lr_readlane_f64(v, 63);
}
__device__ __forceinline__ long long lr_wave_sum_i64(long long x) {
    long long v = x;
    v += lr_dpp_zero_i64<LR_DPP_ROW_SHR(1), 0xf, 0xf>(x);
    v += lr_dpp_zero_i64<LR_DPP_ROW_SHR(2), 0xf, 0xf>(x);
    v += lr_dpp_zero_i64<LR_DPP_ROW_SHR(3), 0xf, 0xf>(x);
    v += lr_dpp_zero_i64<LR_DPP_ROW_SHR(4), 0xf, 0xe>(v);
    v += lr_dpp_zero_i64<LR_DPP_ROW_SHR(8), 0xf, 0xc>(v);
    v += lr_dpp_zero_i64<LR_DPP_ROW_BCAST15, 0xa, 0xf>(v);
    v += lr_dpp_zero_i64<LR_DPP_ROW_BCAST31, 0xc, 0xf>(v);
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((long long)hi << 32) | (unsigned int)lo;
}
// exclusive prefix sum over lanes; *total = sum over all lanes
__device__ __forceinline__ double lr_wave_exclusive_scan(double v, int lane, double* total) {
    const double incl = lr_wave_inclusive_scan(v);
    *total = lr_readlane_f64(incl, 63);
    (void)lane;
    return lr_dpp_zero<0x138 /* wave_shr:1 */, 0xf, 0xf>(incl);   // lane l <- incl[l-1], lane 0 <- 0
}

// exclusive prefix sum of 32-bit integers over the lanes (same DPP pattern)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int lr_dpp_zero_i32(int src) {
    return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ int lr_wave_exclusive_scan_i32(int x) {
    int v = x;
    v += lr_dpp_zero_i32<LR_DPP_ROW_SHR(1), 0xf, 0xf>(x);
    v += lr_dpp_zero_i32<LR_DPP_ROW_SHR(2), 0xf, 0xf>(x);
    v += lr_dpp_zero_i32<LR_DPP_ROW_SHR(3), 0xf, 0xf>(x);
    v += lr_dpp_zero_i32<LR_DPP_ROW_SHR(4), 0xf, 0xe>(v);
    v += lr_dpp_zero_i32<LR_DPP_ROW_SHR(8), 0xf, 0xc>(v);
    v += lr_dpp_zero_i32<LR_DPP_ROW_BCAST15, 0xa, 0xf>(v);
    v += lr_dpp_zero_i32<LR_DPP_ROW_BCAST31, 0xc, 0xf>(v);
    return lr_dpp_zero_i32<0x138 /* wave_shr:1 */, 0xf, 0xf>(v);
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and the draws built on it
// ---------------------------------------------------------------------------------------
#define LR_P_MOVE 0
#define LR_P_MULT 1
#define LR_P_TIMES 2
#define LR_P_RJ 3
#define LR_P_BETA_A 4
#define LR_P_BETA_B 5
#define LR_P_GIBBS_POI 6
#define LR_P_GIBBS_L 7
#define LR_P_GIBBS_M 8
#define LR_P_ACCEPT 9
#define LR_P_INIT 10
#define LR_GAMMA_MAX_ATTEMPTS 32

struct lr_u2 {
    double a, b;
};

__device__ __forceinline__ void lr_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                          uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: integer
        // multiplies run at quarter rate, and a step spends two Philox blocks on them
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

struct lr_stream {
    uint32_t k0, k1;
};

__device__ __forceinline__ lr_u2 lr_pair(const lr_stream& s, uint64_t it, uint32_t purpose, uint32_t idx) {
    uint32_t w[4];
    lr_philox((uint32_t)it, (uint32_t)(it >> 32), purpose, idx, s.k0, s.k1, w);
    lr_u2 r;
    r.a = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
    r.b = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
    return r;
}

__device__ __forceinline__ double lr_normal(const lr_stream& s, uint64_t it, uint32_t purpose, uint32_t idx) {
    const lr_u2 u = lr_pair(s, it, purpose, idx);
    return sqrt(-2.0 * lr_log(1.0 - u.a)) * cospi(2.0 * u.b);   // cos(2 pi u_b) without the range reduction
}

// standard Gamma(shape >= 1): attempt a uses idx base+2a (normal) and base+2a+1 (uniform)
__device__ inline double lr_gamma(const lr_stream& s, uint64_t it, uint32_t purpose, uint32_t base, double shape) {
    const double d = shape - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    for (int a = 0; a < LR_GAMMA_MAX_ATTEMPTS; ++a) {
        const double x = lr_normal(s, it, purpose, base + 2 * a);
        const double t = 1.0 + c * x;
        const double v = t * t * t;
        if (v <= 0.0) continue;
        const double u = lr_pair(s, it, purpose, base + 2 * a + 1).a;
        if (u <= 0.0) return d * v;
        if (lr_log(u) < 0.5 * x * x + d - d * v + d * lr_log(v)) return d * v;
    }
    return d;
}


// ---------------------------------------------------------------------------------------
// table layouts (`mode`) the builders write and the scans read
//   0  chain-major general: entry j of a chain = double2 (value, slope); the launch-based scans take the in-bin
//      fractions from ts / te themselves (fs = ts - floor ts, fe = te - (ceil te - 1))
//   1  unit resolution: every lineage has the same fractions (fs0, fe0), folded into 8-byte entries; pair tables: the two
//      chains of a pair side by side, i.e. a chain's consecutive entries 2 doubles apart
//   2  pair-general (persistent engines on general times): a VALUE plane of 2H entries (v chain 0, v chain 1) and, `so` =
//      4H doubles behind it, a SLOPE plane of 2H entries (s chain 0, s chain 1) - both with the 16-byte entry stride of
//      the unit layout, so that a wave's 16-byte gathers spread over all LDS banks (interleaved 32-byte entries would
//      leave half of the banks idle on every read).  The packed lineages carry fs and fe' = ceil te - te = 1 - fe as
//      32-bit fixed-point fractions, so the slopes are stored times 2^-32 and the death side is kept as
//      E' = (value - R, +R):  value + fe (-R) = (value - R) + fe' R.
// ---------------------------------------------------------------------------------------
#define LR_TAB_GENERAL 0
#define LR_TAB_UNIT 1
#define LR_TAB_PAIRGEN 2
#define LR_FRAC_SCALE 0x1p-32
// `so` argument of the builders: doubles from a chain's value to its slope in the pair-general layout (4H), 2 otherwise
// (the value doubles as the layout switch of the kernels that know their layout at compile time)
__host__ __device__ __forceinline__ int lr_tab_es(int mode, int H) { return mode == LR_TAB_PAIRGEN ? 4 * H : 2; }

// Unit bins one lane of the one-pass table builder handles for the instantiated table sizes H = 40 / 72 / 136 / 264 / 520
// (lr_plan_scan picks H so that n_bins <= 64 * lr_bins_per_lane(H)): a kernel instantiated for H needs one builder only.
__host__ __device__ constexpr int lr_bins_per_lane(int H) { return H <= 40 ? 1 : (H <= 136 ? 2 : (H <= 264 ? 4 : 8)); }

// CS = doubles between consecutive entries of ONE chain in the packed layouts: 2 in a pair table (the two chains of a
// pair side by side), 1 in a column of its own (the speculative kernel keeps every candidate's column apart and lets the
// scanner waves interleave the two that are selected)
// birth-side entry j: value v = logB + cum, exposure rate R
template <int CS = 2>
__device__ __forceinline__ void lr_put_S(double* tabd, int mode, int so, int j, double v, double R, double fs0) {
    if (mode == LR_TAB_UNIT) tabd[CS * j] = v + fs0 * R;
    else if (mode == LR_TAB_PAIRGEN) tabd[CS * j] = v, tabd[CS * j + so] = R * LR_FRAC_SCALE;
    else reinterpret_cast<double2*>(tabd)[j] = make_double2(v, R);
}
// death-side entry j (H + bin + 1): value v = logD - cum, exposure rate R
template <int CS = 2>
__device__ __forceinline__ void lr_put_E(double* tabd, int mode, int so, int j, double v, double R, double fe0) {
    if (mode == LR_TAB_UNIT) tabd[CS * j] = v - fe0 * R;
    else if (mode == LR_TAB_PAIRGEN) tabd[CS * j] = v - R, tabd[CS * j + so] = R * LR_FRAC_SCALE;
    else reinterpret_cast<double2*>(tabd)[j] = make_double2(v, -R);
}

// ---------------------------------------------------------------------------------------
// lookup tables of one chain
//
// Unit bins b = 0..n_bins-1 cover [t0+b, t0+b+1).  Entry j = b+1; j = 0 is "before the
// window", j = n_bins+1 "after it".  With cum_b = sum_{b'<b} R_b' the scan evaluates
//     contribution_i = S[js].x + fs*S[js].y + E[je].x + fe*E[je].y
// where  S[b+1] = (logB_b + cum_b,  R_b)      birth event + exposure integral up to ts
//        E[b+1] = (logD_b - cum_b, -R_b)      death event - exposure integral up to te
//        S[0] = E[0] = (0,0),  S[n_bins+1] = (cum_total, 0),  E[n_bins+1] = (-cum_total, 0).
// Layout per chain: [n_cls][2 (S,E)][H] double2, H >= n_bins+2 (padded so that the fast scan kernel
// can use immediate LDS offsets); class 1 (model 3 only) carries the
// birth process alone and is used by extant lineages (LRF:141-142: death half on te<end_time).
// Model conventions (LRF:137-162):
//   2/3: logB = log lam, logD = log mu, R = lam+mu
//   0  : k>0: logB = log(k*lam), logD = log(mu*k), R = lam+mu ; k==0: all 0
//   1  : k>0: logB = log(lam),   logD = log(mu*k), R = mu, const -= lam ; k==0: all 0
// One wave builds one chain's tables; lane l owns the contiguous bins [l*P, (l+1)*P).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void lr_bin_terms(int model, double lam, double mu, double k, double* logB, double* logD,
                                             double* R, double* Rl, double* cterm) {
    *cterm = 0.0;
    if (model >= 2) {
        *logB = lr_log(lam), *logD = lr_log(mu), *R = lam + mu, *Rl = lam;
        return;
    }
    if (!(k > 0.0)) {
        *logB = 0.0, *logD = 0.0, *R = 0.0, *Rl = 0.0;
        return;
    }
    if (model == 0) {
        *logB = lr_log(k * lam + 0.0), *R = lam + mu;
    } else {
        *logB = lr_log(k * 0.0 + lam), *R = mu, *cterm = -lam;
    }
    *logD = lr_log(mu * k);
    *Rl = 0.0;
}

__device__ inline double lr_build_tables_wave(const double* __restrict__ lam_bins, const double* __restrict__ mu_bins,
                                              const double* __restrict__ br_length, int model, int n_bins, int n_cls,
                                              int H, double2* __restrict__ tab, int lane) {
    const int P = (n_bins + LR_WAVE - 1) / LR_WAVE;
    const int b0 = lane * P;
    const int b1 = min(b0 + P, n_bins);
    double sumR = 0.0, sumRl = 0.0, csum = 0.0;
    for (int b = b0; b < b1; ++b) {
        double logB, logD, R, Rl, ct;
        lr_bin_terms(model, lam_bins[b], mu_bins[b], model < 2 ? br_length[b] : 1.0, &logB, &logD, &R, &Rl, &ct);
        sumR += R, sumRl += Rl, csum += ct;
    }
    double totR, totRl;
    double cum = lr_wave_exclusive_scan(sumR, lane, &totR);
    double cuml = 0.0;
    if (n_cls == 2) cuml = lr_wave_exclusive_scan(sumRl, lane, &totRl);
    for (int b = b0; b < b1; ++b) {
        double logB, logD, R, Rl, ct;
        lr_bin_terms(model, lam_bins[b], mu_bins[b], model < 2 ? br_length[b] : 1.0, &logB, &logD, &R, &Rl, &ct);
        tab[b + 1] = make_double2(logB + cum, R);
        tab[H + b + 1] = make_double2(logD - cum, -R);
        cum += R;
        if (n_cls == 2) {
            tab[2 * H + b + 1] = make_double2(logB + cuml, Rl);
            tab[3 * H + b + 1] = make_double2(-cuml, -Rl);
            cuml += Rl;
        }
    }
    if (lane == 0) {
        tab[0] = make_double2(0.0, 0.0);
        tab[H] = make_double2(0.0, 0.0);
        tab[n_bins + 1] = make_double2(totR, 0.0);
        tab[H + n_bins + 1] = make_double2(-totR, 0.0);
        if (n_cls == 2) {
            tab[2 * H] = make_double2(0.0, 0.0);
            tab[3 * H] = make_double2(0.0, 0.0);
            tab[2 * H + n_bins + 1] = make_double2(totRl, 0.0);
            tab[3 * H + n_bins + 1] = make_double2(-totRl, 0.0);
        }
    }
    return lr_wave_sum(csum);
}


