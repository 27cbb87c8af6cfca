// lr_scan.h - the fast lineage-scan block body, shared by the stand-alone scan kernel
// (lr_loglik.hip) and the fused scan+chain-step kernel of the engine (lr_mcmc.hip).
#pragma once
#include "lr_device.h"
#include "lr_internal.h"

// ------------------------------------------------------------------------------------------
// fast path (one table class, H a template constant): every LDS gather address is
// lane_offset + immediate, the index math is integer (cvt + med3), the next pair of lineages
// is prefetched while the current one is scored.
// ------------------------------------------------------------------------------------------
template <int CB, int H>
__device__ __forceinline__ void lr_score_lineage_fast(double s, double e, double t0, int n_bins,
                                                      const char* __restrict__ lds, double (&acc)[CB]) {
    const double fl = floor(s);
    const double ce = ceil(e);
    // v_cvt_i32_f64 saturates, so far-away times clamp correctly before the med3
    const int a = min(max(__double2int_rz(fl - t0), -1), n_bins);          // birth bin, -1 / n_bins = outside
    const int b = min(max(__double2int_rz(ce - t0), 0), n_bins + 1);       // death entry index
    const double fs = s - fl;
    const double fe = (e - ce) + 1.0;
    const char* pS = lds + ((a + 1) << 4);
    const char* pE = lds + (b << 4) + H * 16;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const double2 S = *reinterpret_cast<const double2*>(pS + c * (2 * H * 16));
        const double2 E = *reinterpret_cast<const double2*>(pE + c * (2 * H * 16));
        double t = S.x + E.x;
        t = fma(fs, S.y, t);
        t = fma(fe, E.y, t);
        acc[c] += t;
    }
}

// Per-chain sums of one wave in log2 steps that HALVE the number of live accumulators: at offset 32
// the lower half-wave keeps chains [0,CB/2), the upper half chains [CB/2,CB), and so on; afterwards a
// plain butterfly over the remaining offsets.  10 exchanges instead of 48 for CB = 8, fixed order.
template <int CB>
__device__ __forceinline__ void lr_wave_reduce_chains(double (&acc)[CB], int lane, double* out /* [CB] */) {
    int off = 32;
#pragma unroll
    for (int n = CB; n > 1; n >>= 1) {
        const int h = n >> 1;
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int k = 0; k < h; ++k) {
            const double send = hi ? acc[k] : acc[k + h];
            const double keep = hi ? acc[k + h] : acc[k];
            acc[k] = keep + __shfl_xor(send, off, LR_WAVE);
        }
        off >>= 1;
    }
    const int group = 2 * off;  // lanes sharing one chain
    for (; off > 0; off >>= 1) acc[0] += __shfl_xor(acc[0], off, LR_WAVE);
    int chain = 0, o = 32;
#pragma unroll
    for (int n = CB; n > 1; n >>= 1) {
        chain = chain * 2 + ((lane & o) ? 1 : 0);
        o >>= 1;
    }
    if ((lane & (group - 1)) == 0) out[chain] = acc[0];
}

// One block: tile `tile` of the lineages x chains [chain0, chain0+CB) of the `n_chains` whose tables start at
// `tables`; partial sums go to partials[tile * partial_stride + chain].
template <int CB, int H>
__device__ __forceinline__ void lr_scan_fast_body(double2* lds, int tile, int chain0, const double* __restrict__ ts,
                                                  const double* __restrict__ te, long long n, double t0, int n_bins,
                                                  const double2* __restrict__ tables, int n_chains, long long chunk,
                                                  double* __restrict__ partials, int partial_stride) {
    constexpr int STRIDE = 2 * H;
    const int tid = threadIdx.x;
    const int nvalid = min(CB, n_chains - chain0);
    const long long start = (long long)tile * chunk;
    const long long end = min(start + chunk, n);
    const bool aligned = ((((uintptr_t)ts) | ((uintptr_t)te)) & 15) == 0;
    long long i = start + 2 * tid;
    // first pair of lineages: in flight while the tables are staged
    double2 s2 = make_double2(0.0, 0.0), e2 = make_double2(0.0, 0.0);
    if (aligned && i + 1 < end) {
        s2 = *reinterpret_cast<const double2*>(ts + i);
        e2 = *reinterpret_cast<const double2*>(te + i);
    }
    {
        // stage the CB tables: all 16-byte global loads are issued back to back (one latency), then written
        const double2* src = tables + (size_t)chain0 * STRIDE;
        const int n_valid_entries = nvalid * STRIDE;
        constexpr int NI = (CB * STRIDE + LR_SCAN_THREADS - 1) / LR_SCAN_THREADS;
        double2 buf[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = tid + k * LR_SCAN_THREADS;
            buf[k] = src[min(i, n_valid_entries - 1)];
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = tid + k * LR_SCAN_THREADS;
            if (i < CB * STRIDE) lds[i] = (i < n_valid_entries) ? buf[k] : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    double acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.0;
    const char* lbase = reinterpret_cast<const char*>(lds);
    if (aligned) {
        while (i + 1 < end) {
            const double2 sc = s2, ec = e2;
            const long long nx = i + 2 * LR_SCAN_THREADS;
            if (nx + 1 < end) {  // prefetch the next pair
                s2 = *reinterpret_cast<const double2*>(ts + nx);
                e2 = *reinterpret_cast<const double2*>(te + nx);
            }
            lr_score_lineage_fast<CB, H>(sc.x, ec.x, t0, n_bins, lbase, acc);
            lr_score_lineage_fast<CB, H>(sc.y, ec.y, t0, n_bins, lbase, acc);
            i = nx;
        }
        if (i < end) lr_score_lineage_fast<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
    } else {
        for (; i < end; i += 2 * LR_SCAN_THREADS) {
            lr_score_lineage_fast<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
            if (i + 1 < end) lr_score_lineage_fast<CB, H>(ts[i + 1], te[i + 1], t0, n_bins, lbase, acc);
        }
    }

    __syncthreads();
    double* red = reinterpret_cast<double*>(lds);
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    lr_wave_reduce_chains<CB>(acc, lane, red + wave * CB);
    __syncthreads();
    if (tid < nvalid) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < LR_SCAN_THREADS / LR_WAVE; ++w) t += red[w * CB + tid];
        partials[(size_t)tile * partial_stride + chain0 + tid] = t;
    }
}


template <int CB, int H>
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_scan_fast_kernel(const double* __restrict__ ts,
                                                                       const double* __restrict__ te, long long n,
                                                                       double t0, int n_bins,
                                                                       const double2* __restrict__ tables,
                                                                       int n_chains, long long chunk,
                                                                       double* __restrict__ partials,
                                                                       int partial_stride) {
    extern __shared__ double2 lds[];
    lr_scan_fast_body<CB, H>(lds, blockIdx.x, blockIdx.y * CB, ts, te, n, t0, n_bins, tables, n_chains, chunk, partials,
                             partial_stride);
}
