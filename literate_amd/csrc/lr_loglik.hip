// lr_loglik.hip - batched per-lineage birth-death log-likelihood.
//
// Hot kernel: lr_scan_kernel<CB>.  One 256-thread block owns a tile of lineages and CB chains.
// The chains' lookup tables (lr_device.h) are staged in LDS once; every lineage (ts, te) is then
// loaded from HBM/L2 once with 16-byte coalesced loads, turned into two table indices and two
// in-bin fractions, and scored against all CB chains with two 16-byte LDS gathers + 4 fp64 ops
// per (lineage, chain).  Per-chain sums are reduced lane -> wave (shuffle butterfly) -> block
// (LDS, fixed order) -> one partial per (tile, chain); partials are summed in tile order by the
// consumer, so results are bitwise reproducible.  No MFMA: this is a gather/scan/reduce.
#include <cstdlib>

#include "lr_device.h"
#include "lr_internal.h"
#include "lr_scan.h"

// ------------------------------------------------------------------------------------------
// tables from per-bin rates: one wave per chain
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LR_WAVE) void lr_build_tables_kernel(const double* __restrict__ lam_bins,
                                                                  const double* __restrict__ mu_bins,
                                                                  const double* __restrict__ br_length, int model,
                                                                  int n_bins, int n_cls, int H, int tab_stride,
                                                                  double2* __restrict__ tables,
                                                                  double* __restrict__ consts) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const double cst = lr_build_tables_wave(lam_bins + (size_t)c * n_bins, mu_bins + (size_t)c * n_bins, br_length,
                                            model, n_bins, n_cls, H, tables + (size_t)c * tab_stride, lane);
    if (lane == 0) consts[c] = cst;
}

// ------------------------------------------------------------------------------------------
// the lineage scan
// ------------------------------------------------------------------------------------------
template <int CB>
__device__ __forceinline__ void lr_score_lineage(double s, double e, double t0, double nb1, int H, int n_cls,
                                                 double end_time, const double2* __restrict__ lds, int tab_stride,
                                                 double (&acc)[CB]) {
    const double fl = floor(s);
    const double ce = ceil(e);
    // table index: births [lo,hi) -> floor, deaths (lo,hi] -> ceil-1; 0 / n_bins+1 = outside
    const int js = (int)fmin(fmax(fl - t0 + 1.0, 0.0), nb1);
    const int je = (int)fmin(fmax(ce - t0, 0.0), nb1);
    const double fs = s - fl;
    const double fe = e - (ce - 1.0);
    int base = 0;
    if (n_cls == 2 && e >= end_time) base = 2 * H;
    const int offS = base + js;
    const int offE = base + H + je;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const double2 S = lds[c * tab_stride + offS];
        const double2 E = lds[c * tab_stride + offE];
        double t = S.x + E.x;
        t = fma(fs, S.y, t);
        t = fma(fe, E.y, t);
        acc[c] += t;
    }
}

template <int CB>
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_scan_kernel(const double* __restrict__ ts,
                                                                  const double* __restrict__ te, long long n,
                                                                  double t0, int n_bins, int n_cls, int H,
                                                                  double end_time,
                                                                  const double2* __restrict__ tables, int tab_stride,
                                                                  int n_chains, long long chunk,
                                                                  double* __restrict__ partials, int partial_stride) {
    extern __shared__ double2 lds[];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int chain0 = blockIdx.y * CB;
    const int nvalid = min(CB, n_chains - chain0);

    // stage the CB tables (16-byte loads, contiguous in global memory)
    {
        const double2* src = tables + (size_t)chain0 * tab_stride;
        const int n_valid_entries = nvalid * tab_stride;
        for (int i = tid; i < CB * tab_stride; i += LR_SCAN_THREADS)
            lds[i] = (i < n_valid_entries) ? src[i] : make_double2(0.0, 0.0);
    }
    __syncthreads();

    double acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.0;

    const long long start = (long long)tile * chunk;
    const long long end = min(start + chunk, n);
    const double nb1 = (double)(n_bins + 1);
    const bool aligned = ((((uintptr_t)ts) | ((uintptr_t)te)) & 15) == 0;  // chunk is even, so start is
    if (aligned) {
        for (long long i = start + 2 * tid; i < end; i += 2 * LR_SCAN_THREADS) {
            if (i + 1 < end) {
                const double2 s2 = *reinterpret_cast<const double2*>(ts + i);
                const double2 e2 = *reinterpret_cast<const double2*>(te + i);
                lr_score_lineage<CB>(s2.x, e2.x, t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
                lr_score_lineage<CB>(s2.y, e2.y, t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
            } else {
                lr_score_lineage<CB>(ts[i], te[i], t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
            }
        }
    } else {
        // same lineage -> thread assignment as the aligned path (identical summation order)
        for (long long i = start + 2 * tid; i < end; i += 2 * LR_SCAN_THREADS) {
            lr_score_lineage<CB>(ts[i], te[i], t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
            if (i + 1 < end)
                lr_score_lineage<CB>(ts[i + 1], te[i + 1], t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
        }
    }

    // lane -> wave -> block, fixed order
    __syncthreads();  // everyone is done reading the tables: reuse the start of LDS as scratch
    double* red = reinterpret_cast<double*>(lds);
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const double w = lr_wave_sum(acc[c]);
        if (lane == 0) red[wave * CB + c] = w;
    }
    __syncthreads();
    if (tid < nvalid) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < LR_SCAN_THREADS / LR_WAVE; ++w) t += red[w * CB + tid];
        partials[(size_t)(chain0 + tid) * partial_stride + tile] = t;
    }
}

// Small inputs (the calc_likelihood seam: LRF:305-308 evaluates ONE state per MCMC iteration on a few thousand lineages):
// one launch instead of three.  A 1024-thread block per chain: wave 0 builds the chain's table in LDS, every thread scores
// lineages tid, tid + 1024, ..., the wave sums (fixed order inside a wave) are added in wave order, out[c] = sum + constant.
#define LR_SMALL_THREADS 1024
__global__ __launch_bounds__(LR_SMALL_THREADS) void lr_loglik_small_kernel(
    const double* __restrict__ ts, const double* __restrict__ te, long long n, double t0, int n_bins, int n_cls, int H,
    double end_time, const double* __restrict__ lam_bins, const double* __restrict__ mu_bins,
    const double* __restrict__ br_length, int model, double* __restrict__ out) {
    extern __shared__ double2 lds[];
    __shared__ double red[LR_SMALL_THREADS / LR_WAVE];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    double cst = 0.0;
    if (wave == 0)
        cst = lr_build_tables_wave(lam_bins + (size_t)c * n_bins, mu_bins + (size_t)c * n_bins, br_length, model, n_bins, n_cls,
                                   H, lds, lane);
    __syncthreads();
    double acc[1] = {0.0};
    const double nb1 = (double)(n_bins + 1);
    const int tab_stride = n_cls * 2 * H;
    for (long long i = tid; i < n; i += LR_SMALL_THREADS)
        lr_score_lineage<1>(ts[i], te[i], t0, nb1, H, n_cls, end_time, lds, tab_stride, acc);
    const double w = lr_wave_sum(acc[0]);
    if (lane == 0) red[wave] = w;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < LR_SMALL_THREADS / LR_WAVE; ++k) t += red[k];
        // `out` may be pinned HOST memory the host polls (ops.LoglikSession): a system-scope release store, so the value
        // leaves the device's caches whatever the coherence mode of the allocation
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + c), (unsigned long long)__double_as_longlong(t + cst),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the inputs the one-launch form takes: few states on few lineages (a block walks ALL lineages for its chain)
static inline bool lr_loglik_small(long long n, int n_chains) {
    static const int off = getenv("LR_LOGLIK_SMALL") ? atoi(getenv("LR_LOGLIK_SMALL")) == 0 : 0;
    return !off && n <= (1ll << 18) && n_chains <= 64 && n * (long long)n_chains <= (1ll << 21);
}

// out[c] = consts[c] + the sum of chain c's tile partials, in a FIXED order: one 256-thread block per chain, thread j adds
// tiles j, j + 256, ... in ascending order (eight loads in flight, absent tiles entering as + 0.0: every round of loads is
// a round trip to memory), 16 threads then add 16 of those sums each, thread 0 the 16.  (One thread per chain walking ~2000 tiles serially took 145 us - five times the scan of 1e7 lineages.)
__global__ __launch_bounds__(256) void lr_reduce_partials_kernel(const double* __restrict__ partials,
                                                                 const double* __restrict__ consts, int tiles,
                                                                 int tile_stride, double* __restrict__ out) {
    __shared__ double red[256 + 16];
    const int c = blockIdx.x, j = threadIdx.x;
    const double* row = partials + (size_t)c * tile_stride;
    double s = 0.0;
    for (int k0 = 0; k0 < tiles; k0 += 8 * 256) {     // (the planner's <= 2048 tiles: ONE round of loads)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q * 256 + j;
            v[q] = k < tiles ? row[k] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    red[j] = s;
    __syncthreads();
    if (j < 16) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[j * 16 + q];
        red[256 + j] = t;
    }
    __syncthreads();
    if (j == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[256 + q];
        out[c] = t + consts[c];
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int lr_plan_scan(long long n, int n_chains, int n_bins, int model, int unit_res, lr_scan_plan* p, int wide) {
    if (n < 1 || n_chains < 1) return LR_ERR_SIZE;
    if (n_bins < 1 || n_bins > LR_MAX_BINS) return LR_ERR_SIZE;
    if (model < 0 || model > 3) return LR_ERR_MODEL;
    p->n_cls = (model == LR_MODEL_KEIDING_DEAD) ? 2 : 1;
    // fast path: one table class and a padded half-stride the kernel is instantiated for
    static const int fast_H[] = {40, 72, 136, 264};
    p->fast = 0;
    p->H = n_bins + 2;
    if (p->n_cls == 1) {
        for (int h : fast_H)
            if (n_bins + 2 <= h && n_bins <= 64 * lr_bins_per_lane(h)) {
                p->H = h, p->fast = 1;
                break;
            }
    }
    p->unit = (unit_res && p->fast) ? 1 : 0;
    int cb = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // unit-resolution tables hold doubles, two chains per 16-byte entry: H double2 per chain
        p->tab_stride = p->unit ? p->H : p->n_cls * 2 * p->H;
        const size_t per_chain_ = (size_t)p->tab_stride * sizeof(double2);
        cb = p->unit ? 16 : 8;
        while (cb > 1 && per_chain_ * cb > LR_SCAN_LDS_BUDGET) cb >>= 1;
        // sixteen chains of general tables per pass when they fit half a CU's LDS (H <= 136) and there are chains for it
        static const int wide_off = getenv("LR_SCAN_WIDE") ? atoi(getenv("LR_SCAN_WIDE")) == 0 : 0;
        if (wide && !wide_off && !p->unit && p->fast && n_chains > 8 && per_chain_ * 16 <= LR_SCAN_LDS_WIDE) cb = 16;
        if (per_chain_ * cb > LR_SCAN_LDS_MAX) return LR_ERR_SIZE;
        while (cb > 1 && cb / 2 >= n_chains) cb >>= 1;  // do not carry empty chain slots
        if (p->unit && cb < 2) {
            p->unit = 0;  // a lone chain cannot fill a pair entry: general tables
            continue;
        }
        break;
    }
    const size_t per_chain = (size_t)p->tab_stride * sizeof(double2);
    p->cb = cb;
    p->threads = (cb == 16 && !p->unit) ? LR_SCAN_WIDE_THREADS : LR_SCAN_THREADS;
    p->groups = (n_chains + cb - 1) / cb;
    if (p->groups > 65535) return LR_ERR_SIZE;
    const long long unit = 2 * p->threads;
    // (lineage tiles aimed at: 1024 = four 256-thread blocks per CU, one round of blocks.  16 chains x 1e7 / 3e7 / 1e8 lineages
    // in the launch-based engine, per iteration: 2048 tiles 43.1 / 103.5 / 297.6 us, 1024: 42.0 / 102.2 / 290.7, 512: 42.4 / 104.7 /
    // 299.9, 4096: 48.6 / 106.1 / 293.5 - every block stages its chains' tables, and the step sums a chain's tile partials)
    static const long long target_blocks = getenv("LR_SCAN_BLOCKS") ? atoll(getenv("LR_SCAN_BLOCKS")) : 1024;
    // (the sixteen-chain general scan runs two 512-thread blocks per CU - 512 blocks, one round of them: measured 4-6 % slower)
    long long tiles = (target_blocks + p->groups - 1) / p->groups;
    // ... but with many groups of sixteen, at least 256 tiles per group (C = 256 at 1e8 lineages: 64 tiles per group 0.54 of the
    // HBM peak, 128: 0.61, 256: 0.63 - a group's tiles share an XCD, and its 64 block slots want several rounds to even out)
    if (p->threads == LR_SCAN_WIDE_THREADS && tiles < 256 && !getenv("LR_SCAN_BLOCKS")) tiles = 256;
    const long long max_tiles = (n + 4 * unit - 1) / (4 * unit);  // >= 8 lineages per thread
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    long long chunk = lr_align_up64((n + tiles - 1) / tiles, unit);
    tiles = (n + chunk - 1) / chunk;
    p->tiles = (int)tiles;
    p->chunk = chunk;
    size_t lds = per_chain * cb;
    // block reduction scratch: generic kernel [waves][cb], fast/unit kernels [cb][threads] + [threads]
    const size_t red = p->fast ? sizeof(double) * ((size_t)cb * p->threads + p->threads)
                               : sizeof(double) * (LR_SCAN_THREADS / LR_WAVE) * cb;
    if (lds < red) lds = red;
    p->lds_bytes = lds;
    return LR_OK;
}

template <int CB, int H>
static int lr_launch_scan_fast(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0,
                               int n_bins, const double2* tables, int n_chains, double* partials, int partial_stride,
                               hipStream_t stream) {
    static bool configured = false;
    if (p.lds_bytes > 64 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_scan_fast_kernel<CB, H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    dim3 grid(p.tiles, (n_chains + CB - 1) / CB);
    hipLaunchKernelGGL((lr_scan_fast_kernel<CB, H>), grid, dim3(LR_SCAN_THREADS), p.lds_bytes, stream, ts, te, n, t0,
                       n_bins, tables, n_chains, p.chunk, partials, partial_stride);
    return (int)hipGetLastError();
}

template <int H>
static int lr_launch_scan_wide(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0, int n_bins,
                               const double2* tables, int n_chains, double* partials, int partial_stride,
                               hipStream_t stream) {
    if (p.lds_bytes > 64 * 1024) {
        // (per call: the attribute belongs to the function on the CURRENT device, and a process may drive several)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_scan_wide_kernel<H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(p.tiles, (n_chains + 15) / 16);
    hipLaunchKernelGGL((lr_scan_wide_kernel<H>), grid, dim3(LR_SCAN_WIDE_THREADS), p.lds_bytes, stream, ts, te, n, t0, n_bins,
                       tables, n_chains, p.chunk, partials, partial_stride);
    return (int)hipGetLastError();
}

template <int H>
static int lr_launch_scan_fast_h(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0,
                                 int n_bins, const double2* tables, int n_chains, double* partials,
                                 int partial_stride, hipStream_t stream) {
    switch (p.cb) {
        case 16: return lr_launch_scan_wide<H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 8: return lr_launch_scan_fast<8, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 4: return lr_launch_scan_fast<4, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 2: return lr_launch_scan_fast<2, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        default: return lr_launch_scan_fast<1, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
    }
}

template <int CB>
static int lr_launch_scan_cb(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0,
                             int n_bins, double end_time, const double2* tables, int n_chains, double* partials,
                             int partial_stride, hipStream_t stream) {
    static size_t configured = 64 * 1024;
    if (p.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_scan_kernel<CB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = p.lds_bytes;
    }
    dim3 grid(p.tiles, (n_chains + CB - 1) / CB);
    hipLaunchKernelGGL(lr_scan_kernel<CB>, grid, dim3(LR_SCAN_THREADS), p.lds_bytes, stream, ts, te, n, t0, n_bins,
                       p.n_cls, p.H, end_time, tables, p.tab_stride, n_chains, p.chunk, partials, partial_stride);
    return (int)hipGetLastError();
}

template <int CB, int H>
static int lr_launch_scan_unit(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0,
                               int n_bins, const double2* tables, int n_chains, double* partials, int partial_stride,
                               hipStream_t stream) {
    dim3 grid(p.tiles, (n_chains + CB - 1) / CB);
    hipLaunchKernelGGL((lr_scan_unit_kernel<CB, H>), grid, dim3(LR_SCAN_THREADS), p.lds_bytes, stream, ts, te, n, t0,
                       n_bins, tables, n_chains, p.chunk, partials, partial_stride);
    return (int)hipGetLastError();
}

template <int H>
static int lr_launch_scan_unit_h(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0,
                                 int n_bins, const double2* tables, int n_chains, double* partials,
                                 int partial_stride, hipStream_t stream) {
    switch (p.cb) {
        case 16: return lr_launch_scan_unit<16, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 8: return lr_launch_scan_unit<8, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 4: return lr_launch_scan_unit<4, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        case 2: return lr_launch_scan_unit<2, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
        default: return lr_launch_scan_unit<1, H>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
    }
}

int lr_launch_scan(const lr_scan_plan& p, const double* ts, const double* te, long long n, double t0, int n_bins,
                   double end_time, const double2* tables, int n_chains, double* partials, int partial_stride,
                   hipStream_t stream) {
    if (p.unit) {
        switch (p.H) {
            case 40: return lr_launch_scan_unit_h<40>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 72: return lr_launch_scan_unit_h<72>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 136: return lr_launch_scan_unit_h<136>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 264: return lr_launch_scan_unit_h<264>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            default: return LR_ERR_SIZE;
        }
    }
    if (p.fast) {
        switch (p.H) {
            case 40: return lr_launch_scan_fast_h<40>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 72: return lr_launch_scan_fast_h<72>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 136: return lr_launch_scan_fast_h<136>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            case 264: return lr_launch_scan_fast_h<264>(p, ts, te, n, t0, n_bins, tables, n_chains, partials, partial_stride, stream);
            default: return LR_ERR_SIZE;
        }
    }
    switch (p.cb) {
        case 8: return lr_launch_scan_cb<8>(p, ts, te, n, t0, n_bins, end_time, tables, n_chains, partials, partial_stride, stream);
        case 4: return lr_launch_scan_cb<4>(p, ts, te, n, t0, n_bins, end_time, tables, n_chains, partials, partial_stride, stream);
        case 2: return lr_launch_scan_cb<2>(p, ts, te, n, t0, n_bins, end_time, tables, n_chains, partials, partial_stride, stream);
        default: return lr_launch_scan_cb<1>(p, ts, te, n, t0, n_bins, end_time, tables, n_chains, partials, partial_stride, stream);
    }
}

// workspace layout of lr_bd_loglik_batch: [tables | consts | partials], each 256-byte aligned
static void lr_loglik_ws(const lr_scan_plan& p, int n_chains, size_t* off_tables, size_t* off_consts,
                         size_t* off_partials, size_t* total) {
    size_t o = 0;
    *off_tables = o, o += lr_align_up64((long long)n_chains * p.tab_stride * sizeof(double2), 256);
    *off_consts = o, o += lr_align_up64((long long)n_chains * sizeof(double), 256);
    *off_partials = o, o += lr_align_up64((long long)lr_tile_stride(p.tiles) * n_chains * sizeof(double), 256);
    *total = o;
}

extern "C" int64_t lr_bd_loglik_workspace_bytes(int64_t n, int32_t n_bins, int32_t n_chains, int32_t model) {
    lr_scan_plan p;
    const int rc = lr_plan_scan(n, n_chains, n_bins, model, 0, &p, 1);
    if (rc != LR_OK) return rc;
    size_t a, b, c, total;
    lr_loglik_ws(p, n_chains, &a, &b, &c, &total);
    return (int64_t)total;
}

extern "C" int lr_bd_loglik_plan(int64_t n, int32_t n_bins, int32_t n_chains, int32_t model, int32_t* out) {
    if (!out) return LR_ERR_NULL;
    lr_scan_plan p;
    const int rc = lr_plan_scan(n, n_chains, n_bins, model, 0, &p, 1);
    if (rc != LR_OK) return rc;
    out[0] = p.cb, out[1] = p.tiles, out[2] = p.H, out[3] = p.groups;
    return LR_OK;
}

extern "C" int lr_bd_loglik_batch(const double* ts, const double* te, int64_t n, double t0, int32_t n_bins,
                                  const double* lam_bins, const double* mu_bins, int32_t n_chains, int32_t model,
                                  const double* br_length, double end_time, double* out_loglik, void* workspace,
                                  int64_t workspace_bytes, void* stream_) {
    if (!ts || !te || !lam_bins || !mu_bins || !out_loglik || !workspace) return LR_ERR_NULL;
    if ((model == LR_MODEL_BD || model == LR_MODEL_ID) && !br_length) return LR_ERR_MODEL;
    if (t0 != floor(t0)) return LR_ERR_T0;
    lr_scan_plan p;
    int rc = lr_plan_scan(n, n_chains, n_bins, model, 0, &p, 1);
    if (rc != LR_OK) return rc;
    size_t o_tab, o_cst, o_par, total;
    lr_loglik_ws(p, n_chains, &o_tab, &o_cst, &o_par, &total);
    if ((int64_t)total > workspace_bytes) return LR_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (lr_loglik_small(n, n_chains)) {
        const size_t lds = (size_t)p.n_cls * 2 * (n_bins + 2) * sizeof(double2);
        if (lds <= 60 * 1024) {
            hipLaunchKernelGGL(lr_loglik_small_kernel, dim3(n_chains), dim3(LR_SMALL_THREADS), lds, stream, ts, te, (long long)n, t0,
                               n_bins, p.n_cls, n_bins + 2, end_time, lam_bins, mu_bins, br_length, model, out_loglik);
            return (int)hipGetLastError();
        }
    }
    char* ws = (char*)workspace;
    double2* tables = (double2*)(ws + o_tab);
    double* consts = (double*)(ws + o_cst);
    double* partials = (double*)(ws + o_par);
    hipLaunchKernelGGL(lr_build_tables_kernel, dim3(n_chains), dim3(LR_WAVE), 0, stream, lam_bins, mu_bins, br_length,
                       model, n_bins, p.n_cls, p.H, p.tab_stride, tables, consts);
    rc = (int)hipGetLastError();
    if (rc) return rc;
    rc = lr_launch_scan(p, ts, te, n, t0, n_bins, end_time, tables, n_chains, partials, lr_tile_stride(p.tiles), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lr_reduce_partials_kernel, dim3(n_chains), dim3(256), 0, stream, partials, consts, p.tiles,
                       lr_tile_stride(p.tiles), out_loglik);
    return (int)hipGetLastError();
}

#ifdef LR_DIAG
extern "C" int lr_diag_dump(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_buf), (size_t)n_words * 8);
}
#endif
