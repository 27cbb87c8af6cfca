// lr_stats.hip - sufficient statistics (A1/A2), rate-index expansion (A3) and DDRate rates (A12).
#include "lr_device.h"
#include "lr_dd.h"
#include "lr_internal.h"

// ------------------------------------------------------------------------------------------
// A1/A2: events and lineage-time of n_windows windows.  Block = (lineage tile, group of BW
// windows); each lineage is loaded once per block and tested against the BW windows held in
// registers.  Partials [tile][window] are summed in tile order by lr_bin_final_kernel.
// ------------------------------------------------------------------------------------------
#define LR_BW 8
#define LR_BIN_THREADS 256

__global__ __launch_bounds__(LR_BIN_THREADS) void lr_bin_partial_kernel(
    const double* __restrict__ ts, const double* __restrict__ te, long long n, const double* __restrict__ win_lo,
    const double* __restrict__ win_hi, int n_windows, long long chunk, long long* __restrict__ p_sp,
    long long* __restrict__ p_ex, double* __restrict__ p_br) {
    __shared__ long long s_sp[LR_BIN_THREADS / LR_WAVE][LR_BW];
    __shared__ long long s_ex[LR_BIN_THREADS / LR_WAVE][LR_BW];
    __shared__ double s_br[LR_BIN_THREADS / LR_WAVE][LR_BW];
    const int tid = threadIdx.x, tile = blockIdx.x, w0 = blockIdx.y * LR_BW;
    double lo[LR_BW], hi[LR_BW], br[LR_BW];
    int csp[LR_BW], cex[LR_BW];
#pragma unroll
    for (int w = 0; w < LR_BW; ++w) {
        const bool ok = w0 + w < n_windows;
        // an empty window (lo > hi) matches nothing
        lo[w] = ok ? win_lo[w0 + w] : 1.0;
        hi[w] = ok ? win_hi[w0 + w] : 0.0;
        br[w] = 0.0, csp[w] = 0, cex[w] = 0;
    }
    const long long start = (long long)tile * chunk, end = min(start + chunk, n);
    for (long long i = start + tid; i < end; i += LR_BIN_THREADS) {
        const double s = ts[i], e = te[i];
#pragma unroll
        for (int w = 0; w < LR_BW; ++w) {
            csp[w] += (s >= lo[w]) & (s < hi[w]);
            cex[w] += (e > lo[w]) & (e <= hi[w]);
            const double d = fmin(e, hi[w]) - fmax(s, lo[w]);
            br[w] += (d > 0.0) ? d : 0.0;
        }
    }
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
#pragma unroll
    for (int w = 0; w < LR_BW; ++w) {
        const long long a = lr_wave_sum_i64(csp[w]);
        const long long b = lr_wave_sum_i64(cex[w]);
        const double c = lr_wave_sum(br[w]);
        if (lane == 0) s_sp[wave][w] = a, s_ex[wave][w] = b, s_br[wave][w] = c;
    }
    __syncthreads();
    if (tid < LR_BW && w0 + tid < n_windows) {
        long long a = 0, b = 0;
        double c = 0.0;
        for (int v = 0; v < LR_BIN_THREADS / LR_WAVE; ++v) a += s_sp[v][tid], b += s_ex[v][tid], c += s_br[v][tid];
        const size_t o = (size_t)tile * n_windows + w0 + tid;
        p_sp[o] = a, p_ex[o] = b, p_br[o] = c;
    }
}

__global__ void lr_bin_final_kernel(const long long* __restrict__ p_sp, const long long* __restrict__ p_ex,
                                    const double* __restrict__ p_br, int tiles, int n_windows,
                                    long long* __restrict__ sp, long long* __restrict__ ex, double* __restrict__ br) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_windows) return;
    long long a = 0, b = 0;
    double c = 0.0;
    for (int t = 0; t < tiles; ++t) {
        const size_t o = (size_t)t * n_windows + w;
        a += p_sp[o], b += p_ex[o], c += p_br[o];
    }
    sp[w] = a, ex[w] = b, br[w] = c;
}

static void lr_bin_plan(long long n, int n_windows, int* tiles, long long* chunk) {
    const int groups = (n_windows + LR_BW - 1) / LR_BW;
    long long t = (1024 + groups - 1) / groups;
    const long long max_t = (n + 4 * LR_BIN_THREADS - 1) / (4 * LR_BIN_THREADS);
    if (t > max_t) t = max_t;
    if (t < 1) t = 1;
    *chunk = lr_align_up64((n + t - 1) / t, LR_BIN_THREADS);
    *tiles = (int)((n + *chunk - 1) / *chunk);
}

extern "C" int64_t lr_bin_events_workspace_bytes(int64_t n, int32_t n_windows) {
    if (n < 1 || n_windows < 1) return LR_ERR_SIZE;
    int tiles;
    long long chunk;
    lr_bin_plan(n, n_windows, &tiles, &chunk);
    return 3 * lr_align_up64((long long)tiles * n_windows * 8, 256);
}

extern "C" int lr_bin_events(const double* ts, const double* te, int64_t n, const double* win_lo, const double* win_hi,
                             int32_t n_windows, int64_t* sp_events, int64_t* ex_events, double* br_length,
                             void* workspace, int64_t workspace_bytes, void* stream_) {
    if (!ts || !te || !win_lo || !win_hi || !sp_events || !ex_events || !br_length || !workspace) return LR_ERR_NULL;
    if (n < 1 || n_windows < 1 || n_windows > 65535 * LR_BW) return LR_ERR_SIZE;
    int tiles;
    long long chunk;
    lr_bin_plan(n, n_windows, &tiles, &chunk);
    const long long seg = lr_align_up64((long long)tiles * n_windows * 8, 256);
    if (3 * seg > workspace_bytes) return LR_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    long long* p_sp = (long long*)ws;
    long long* p_ex = (long long*)(ws + seg);
    double* p_br = (double*)(ws + 2 * seg);
    dim3 grid(tiles, (n_windows + LR_BW - 1) / LR_BW);
    hipLaunchKernelGGL(lr_bin_partial_kernel, grid, dim3(LR_BIN_THREADS), 0, stream, ts, te, (long long)n, win_lo,
                       win_hi, n_windows, chunk, p_sp, p_ex, p_br);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    hipLaunchKernelGGL(lr_bin_final_kernel, dim3((n_windows + 127) / 128), dim3(128), 0, stream, p_sp, p_ex, p_br,
                       tiles, n_windows, (long long*)sp_events, (long long*)ex_events, br_length);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// A3: rate index expansion
// ------------------------------------------------------------------------------------------
__global__ void lr_expand_rates_kernel(const double* __restrict__ rates, const double* __restrict__ times,
                                       const int* __restrict__ K, int kmax, int n_chains, int n_bins, int mode,
                                       double* __restrict__ rate_bins) {
    const int c = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const double* t = times + (size_t)c * (kmax + 1);
    const int k = K[c];
    const double e0 = mode ? rint(t[0]) : floor(t[0]);
    int seg = 0;
    for (int j = 1; j < k; ++j) {
        const double ej = mode ? rint(t[j]) : floor(t[j]);
        if ((int)(ej - e0) <= b) seg = j;
    }
    rate_bins[(size_t)c * n_bins + b] = rates[(size_t)c * kmax + seg];
}

extern "C" int lr_expand_rates(const double* rates, const double* times, const int32_t* K, int32_t kmax,
                               int32_t n_chains, int32_t n_bins, int32_t mode, double* rate_bins, void* stream_) {
    if (!rates || !times || !K || !rate_bins) return LR_ERR_NULL;
    if (kmax < 1 || n_chains < 1 || n_chains > 65535 || n_bins < 1) return LR_ERR_SIZE;
    dim3 grid((n_bins + 127) / 128, n_chains);
    hipLaunchKernelGGL(lr_expand_rates_kernel, grid, dim3(128), 0, (hipStream_t)stream_, rates, times, K, kmax,
                       n_chains, n_bins, mode, rate_bins);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// A12: DDRate per-bin rates (DD:55-100); x = bin index (TIME_RANGE, lib:255)
// ------------------------------------------------------------------------------------------
__global__ void lr_dd_rates_kernel(const double* __restrict__ args, const double* __restrict__ DT, int n_bins,
                                   int n_chains, int m_birth, int m_death, double* __restrict__ birth,
                                   double* __restrict__ death, double* __restrict__ niche_o,
                                   double* __restrict__ frac_o) {
    const int c = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const double* a = args + (size_t)c * LR_DD_NPAR;
    const lr_dd_params p{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]};
    double niche, frac, br, dr;
    lr_dd_bin_rates(p, (double)b, DT[b], m_birth, m_death, &br, &dr, &niche, &frac);   // shared with the engine's DD step
    const size_t o = (size_t)c * n_bins + b;
    birth[o] = br, death[o] = dr, niche_o[o] = niche, frac_o[o] = frac;
}

extern "C" int lr_dd_rates(const double* args, const double* DT, int32_t n_bins, int32_t n_chains, int32_t m_birth,
                           int32_t m_death, double* birth_rates, double* death_rates, double* niche,
                           double* niche_frac, void* stream_) {
    if (!args || !DT || !birth_rates || !death_rates || !niche || !niche_frac) return LR_ERR_NULL;
    if (n_bins < 1 || n_chains < 1 || n_chains > 65535) return LR_ERR_SIZE;
    if (m_birth < 0 || m_birth > 2 || m_death < -2 || m_death > 2) return LR_ERR_MODEL;
    dim3 grid((n_bins + 127) / 128, n_chains);
    hipLaunchKernelGGL(lr_dd_rates_kernel, grid, dim3(128), 0, (hipStream_t)stream_, args, DT, n_bins, n_chains,
                       m_birth, m_death, birth_rates, death_rates, niche, niche_frac);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// the other rate maps of the reference that feed the same per-bin likelihood.
// DDRatev2.py:55-104 (9 parameters) and trend_rate.py:73-88 (6 parameters + a per-bin covariate).
// ------------------------------------------------------------------------------------------
__global__ void lr_ddv2_rates_kernel(const double* __restrict__ args, const double* __restrict__ DT, int n_bins,
                                     int m_birth, int m_death, double* __restrict__ birth, double* __restrict__ death,
                                     double* __restrict__ niche_o, double* __restrict__ frac_o) {
    const int c = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const double* a = args + (size_t)c * 9;
    const double l_f = a[0], l_mul = a[1], k = a[2], x0 = a[3], div_0 = a[4], L = a[5], m_mul = a[6], nuB = a[7],
                 nuD = a[8];
    const double x = (double)b, dt = DT[b];
    const double SMALL = 0.000000000000001;
    double niche = 1.0, frac = 1.0, br, dr;
    if (m_birth == 0) {
        br = 1.0 * l_f * l_mul;                                            // DDRatev2.py:77
    } else {
        niche = (m_birth == 1) ? 1.0 * (L + div_0) : div_0 + L / (1.0 + exp(-k * (x - x0)));
        frac = dt / niche;
        const double rate_max = l_f + l_f * l_mul;                         // get_brates, DDRatev2.py:61-65
        br = rate_max - (rate_max - l_f) * pow(frac, nuB);
        if (br <= 0.0) br = SMALL;
    }
    if (m_death <= 0) {
        dr = 1.0;                                                          // np.ones, DDRatev2.py:91
    } else {
        niche = (m_death == 1) ? 1.0 * (L + div_0) : div_0 + L / (1.0 + exp(-k * (x - x0)));
        frac = dt / niche;
        const double rate_min = l_f - l_f * m_mul;                         // get_drates on l_f, DDRatev2.py:67-71, 99
        dr = rate_min + (l_f - rate_min) * pow(frac, nuD);
        if (dr <= 0.0) dr = SMALL;
    }
    const size_t o = (size_t)c * n_bins + b;
    birth[o] = br, death[o] = dr, niche_o[o] = niche, frac_o[o] = frac;
}

extern "C" int lr_ddv2_rates(const double* args, const double* DT, int32_t n_bins, int32_t n_chains, int32_t m_birth,
                             int32_t m_death, double* birth_rates, double* death_rates, double* niche,
                             double* niche_frac, void* stream_) {
    if (!args || !DT || !birth_rates || !death_rates || !niche || !niche_frac) return LR_ERR_NULL;
    if (n_bins < 1 || n_chains < 1 || n_chains > 65535) return LR_ERR_SIZE;
    if (m_birth < 0 || m_birth > 2 || m_death < -2 || m_death > 2) return LR_ERR_MODEL;
    dim3 grid((n_bins + 127) / 128, n_chains);
    hipLaunchKernelGGL(lr_ddv2_rates_kernel, grid, dim3(128), 0, (hipStream_t)stream_, args, DT, n_bins, m_birth,
                       m_death, birth_rates, death_rates, niche, niche_frac);
    return (int)hipGetLastError();
}

__global__ void lr_trend_rates_kernel(const double* __restrict__ args, const double* __restrict__ trend, int n_bins,
                                      int const_birth, int const_death, double* __restrict__ birth,
                                      double* __restrict__ death) {
    const int c = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const double* a = args + (size_t)c * LR_TR_NPAR;
    const lr_trend_params p{a[0], a[1], a[2], a[3], a[4], a[5]};
    double br, dr;
    lr_trend_bin_rates(p, trend[b], const_birth, const_death, &br, &dr);   // shared with the engine's trend step
    const size_t o = (size_t)c * n_bins + b;
    birth[o] = br, death[o] = dr;
}

extern "C" int lr_trend_rates(const double* args, const double* trend, int32_t n_bins, int32_t n_chains,
                              int32_t const_birth, int32_t const_death, double* birth_rates, double* death_rates,
                              void* stream_) {
    if (!args || !trend || !birth_rates || !death_rates) return LR_ERR_NULL;
    if (n_bins < 1 || n_chains < 1 || n_chains > 65535) return LR_ERR_SIZE;
    dim3 grid((n_bins + 127) / 128, n_chains);
    hipLaunchKernelGGL(lr_trend_rates_kernel, grid, dim3(128), 0, (hipStream_t)stream_, args, trend, n_bins,
                       const_birth, const_death, birth_rates, death_rates);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Binned Keiding halves, sum_b log(rate_b) * events_b - rate_b * DT_b (DD:86, 101; trend_rate.py:82, 89): the
// `likelihood_birth` / `likelihood_death` log columns of the DDRate-family samplers (the engine itself scores the
// proposal per lineage and carries only the sum).  One wave per parameter vector, fixed summation order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LR_WAVE) void lr_binned_keiding_kernel(const double* __restrict__ birth,
                                                                    const double* __restrict__ death,
                                                                    const long long* __restrict__ n_spec,
                                                                    const long long* __restrict__ n_exti,
                                                                    const double* __restrict__ DT, int n_bins,
                                                                    double* __restrict__ out_birth,
                                                                    double* __restrict__ out_death) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double sb = 0.0, sd = 0.0;
    for (int b = lane; b < n_bins; b += LR_WAVE) {
        const double lb = birth[(size_t)c * n_bins + b], ld = death[(size_t)c * n_bins + b];
        sb += lr_log(lb) * (double)n_spec[b] - lb * DT[b];
        sd += lr_log(ld) * (double)n_exti[b] - ld * DT[b];
    }
    sb = lr_wave_sum(sb), sd = lr_wave_sum(sd);
    if (lane == 0) out_birth[c] = sb, out_death[c] = sd;
}

extern "C" int lr_binned_keiding(const double* birth_rates, const double* death_rates, const int64_t* n_spec,
                                 const int64_t* n_exti, const double* DT, int32_t n_bins, int32_t n_chains,
                                 double* out_birth, double* out_death, void* stream_) {
    if (!birth_rates || !death_rates || !n_spec || !n_exti || !DT || !out_birth || !out_death) return LR_ERR_NULL;
    if (n_bins < 1 || n_chains < 1) return LR_ERR_SIZE;
    hipLaunchKernelGGL(lr_binned_keiding_kernel, dim3(n_chains), dim3(LR_WAVE), 0, (hipStream_t)stream_, birth_rates,
                       death_rates, (const long long*)n_spec, (const long long*)n_exti, DT, n_bins, out_birth, out_death);
    return (int)hipGetLastError();
}

extern "C" int lr_version(void) { return 101; }
