// lr_stats.hip - sufficient statistics (A1/A2), rate-index expansion (A3) and DDRate rates (A12).
#include <cstdlib>

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
            // the reference's own clipping (lib:74-79: `te[te > t1] = t1`, `ts[ts < t0] = t0`): a NaN stays a NaN and
            // drops out at `br > 0` - fmin / fmax would replace it by the window edge
            const double d = (e > hi[w] ? hi[w] : e) - (s < lo[w] ? lo[w] : s);
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
// A1/A2 for UNIT windows [t0 + w, t0 + w + 1], w < n_bins (the loop LRF:519-523, create_bins lib:231-245): ONE pass over
// ts / te, 16 bytes per lineage - the HBM-bound form.  Per lineage: birth bin bs = floor(ts) - t0, death bin
// be = ceil(te) - 1 - t0, two histogram increments (the events) and, for the lineage-time, only the PARTIAL bins:
//     bs == be : te - ts in that bin;      else : (floor(ts) + 1) - ts in bin bs, te - (ceil(te) - 1) in bin be
// - the very differences get_br forms for those windows (lib:74-79) - while the bins a lineage spans completely count 1
// each: A[b] = #{bs < b} - #{be <= b} + #{bs == be == b}, prefix sums of the two event histograms.  Lineages with
// te <= ts (or a NaN) count as events but carry no time, as in the reference (`br > 0`), through a correction histogram.
// Everything is accumulated in INTEGERS, so the result does not depend on the order of the atomics: a partial term t in
// (0, 1] enters as the 52-bit integer  bits(1.0 + t) - bits(1.0)  (exact whenever t is a multiple of 2^-52: every time
// >= 1, hence every dataset; otherwise rounded to nearest once), split into a 32-bit and a 20-bit limb so that 64-bit
// counters cannot overflow, and br_length[b] = the exact integer sum A[b] * 2^52 + sum of terms, rounded to fp64 ONCE -
// the correctly rounded sum where the reference's pairwise numpy sum carries ~1e-16 * log N.
// LDS: every histogram exists in R copies, lane l using copy l % R (entry = bin * R + copy: the copies of one bin lie in
// different banks), so 64 lanes that hit the same bin - the normal case on input sorted by birth - do not serialise.
// ------------------------------------------------------------------------------------------
#define LR_UB_THREADS 1024
#define LR_UB_MAX_BLOCKS 1024

struct lr_ub_shape {
    int W;          // windows
    int rshift;     // log2 of the copies per histogram
    int n32;        // 32-bit histogram entries per copy: bs [W+2] | be [W+2] | adj [W]
    int cols;       // 64-bit partial columns per block: n32 + hi [W] + lo [W]
    size_t lds_bytes;
};

static lr_ub_shape lr_ub_plan(int W) {
    lr_ub_shape p;
    p.W = W;
    p.n32 = 3 * W + 4;
    p.cols = p.n32 + 2 * W;
    static const int rs_env = getenv("LR_UB_RSHIFT") ? atoi(getenv("LR_UB_RSHIFT")) : 5;     // (A/B runs: copies = 1 << rshift)
    p.rshift = rs_env < 0 ? 0 : (rs_env > 5 ? 5 : rs_env);
    for (;;) {
        p.lds_bytes = ((size_t)p.n32 * 4 + (size_t)2 * W * 8) << p.rshift;
        if (p.lds_bytes <= 144 * 1024 || p.rshift == 0) break;
        --p.rshift;
    }
    if (p.lds_bytes < (size_t)p.cols * 8) p.lds_bytes = (size_t)p.cols * 8;   // the last block's column totals
    p.lds_bytes = (p.lds_bytes + 15) & ~(size_t)15;                           // zeroed in 16-byte stores, all of it
    return p;
}

__device__ __forceinline__ unsigned long long lr_ub_term(double t) {
    return (unsigned long long)(__double_as_longlong(1.0 + t) - 0x3ff0000000000000ll);
}

__device__ __forceinline__ void lr_ub_add_term(unsigned long long* hi, unsigned long long* lo, int idx, double t) {
    const unsigned long long u = lr_ub_term(t);
    atomicAdd(hi + idx, u >> 20);
    const unsigned long long l = u & 0xfffffull;
    if (l) atomicAdd(lo + idx, l);          // year-resolution data: every term is a multiple of 2^-32, no low limb
}

__device__ __forceinline__ void lr_ub_lineage(double s, double e, double t0, int W, int rshift, int copy,
                                              unsigned int* h_bs, unsigned int* h_be, int* h_adj,
                                              unsigned long long* h_hi, unsigned long long* h_lo) {
    const double fl = floor(s), ce = ceil(e);
    // entry index = bin + 1; 0 = before the first window, W + 1 = past the last (v_cvt_i32_f64 saturates)
    int ia = min(max(__double2int_rz(fl - t0), -1), W) + 1;
    int ib = min(max(__double2int_rz(ce - t0), 0), W + 1);
    if (s != s) ia = W + 1;                 // a NaN is in no window (every comparison of LRF:119-120 is false)
    if (e != e) ib = W + 1;
    atomicAdd(h_bs + ((ia << rshift) | copy), 1u);
    atomicAdd(h_be + ((ib << rshift) | copy), 1u);
    if (s < e) {
        if (ia == ib) {
            if (ia >= 1 && ia <= W) {
                atomicAdd(h_adj + (((ia - 1) << rshift) | copy), 1);
                lr_ub_add_term(h_hi, h_lo, ((ia - 1) << rshift) | copy, e - s);
            }
        } else {
            if (ia >= 1 && ia <= W) lr_ub_add_term(h_hi, h_lo, ((ia - 1) << rshift) | copy, (fl + 1.0) - s);
            if (ib >= 1 && ib <= W) lr_ub_add_term(h_hi, h_lo, ((ib - 1) << rshift) | copy, e - (ce - 1.0));
        }
    } else {
        // no lineage-time: cancel what the prefix sums of the two histograms would count for it
        const int bs = ia - 1, be = ib - 1;
        const int b_lo = max(min(bs, be), 0), b_hi = min(max(bs, be), W - 1);
        for (int b = b_lo; b <= b_hi; ++b) {
            const int d = (int)(be <= b) - (int)(bs < b);
            if (d) atomicAdd(h_adj + ((b << rshift) | copy), d);
        }
    }
}

__global__ __launch_bounds__(LR_UB_THREADS) void lr_bin_unit_kernel(const double* __restrict__ ts,
                                                                   const double* __restrict__ te, long long n, double t0,
                                                                   lr_ub_shape p, long long chunk,
                                                                   long long* __restrict__ acc,
                                                                   unsigned int* __restrict__ ticket,
                                                                   long long* __restrict__ sp, long long* __restrict__ ex,
                                                                   double* __restrict__ br) {
    extern __shared__ unsigned long long ub_lds[];
    const int tid = threadIdx.x, W = p.W, rshift = p.rshift, R = 1 << rshift;
    unsigned long long* h_hi = ub_lds;
    unsigned long long* h_lo = h_hi + ((size_t)W << rshift);
    unsigned int* h_bs = reinterpret_cast<unsigned int*>(h_lo + ((size_t)W << rshift));
    unsigned int* h_be = h_bs + ((size_t)(W + 2) << rshift);
    int* h_adj = reinterpret_cast<int*>(h_be + ((size_t)(W + 2) << rshift));
    const long long start = (long long)blockIdx.x * chunk, end = min(start + chunk, n);
    const bool aligned = ((((uintptr_t)ts) | ((uintptr_t)te)) & 15) == 0;     // chunk is even, so start is
    long long i = start + 2 * tid;
    // two pairs of lineages per thread in flight (64 B): 64 KB per CU at one block of 16 waves per CU
    double2 s2 = make_double2(0.0, 0.0), e2 = s2, s3 = s2, e3 = s2;
    if (aligned && i + 1 < end) {
        s2 = *reinterpret_cast<const double2*>(ts + i);
        e2 = *reinterpret_cast<const double2*>(te + i);
        if (i + 2 * LR_UB_THREADS + 1 < end) {
            s3 = *reinterpret_cast<const double2*>(ts + i + 2 * LR_UB_THREADS);
            e3 = *reinterpret_cast<const double2*>(te + i + 2 * LR_UB_THREADS);
        }
    }
    {
        uint4* z = reinterpret_cast<uint4*>(ub_lds);
        const int n16 = (int)(p.lds_bytes >> 4);
        for (int k = tid; k < n16; k += LR_UB_THREADS) z[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int copy = tid & (R - 1);
    if (aligned) {
        while (i + 1 < end) {
            const double2 sc = s2, ec = e2;
            s2 = s3, e2 = e3;
            const long long nx2 = i + 4 * LR_UB_THREADS;
            if (nx2 + 1 < end) {
                s3 = *reinterpret_cast<const double2*>(ts + nx2);
                e3 = *reinterpret_cast<const double2*>(te + nx2);
            }
            lr_ub_lineage(sc.x, ec.x, t0, W, rshift, copy, h_bs, h_be, h_adj, h_hi, h_lo);
            lr_ub_lineage(sc.y, ec.y, t0, W, rshift, copy, h_bs, h_be, h_adj, h_hi, h_lo);
            i += 2 * LR_UB_THREADS;
        }
        if (i < end) lr_ub_lineage(ts[i], te[i], t0, W, rshift, copy, h_bs, h_be, h_adj, h_hi, h_lo);
    } else {
        for (; i < end; i += 2 * LR_UB_THREADS) {
            lr_ub_lineage(ts[i], te[i], t0, W, rshift, copy, h_bs, h_be, h_adj, h_hi, h_lo);
            if (i + 1 < end) lr_ub_lineage(ts[i + 1], te[i + 1], t0, W, rshift, copy, h_bs, h_be, h_adj, h_hi, h_lo);
        }
    }
    __syncthreads();
    // fold the R copies of every histogram entry and add the block's column sums into the call's 64-bit accumulators
    // (agent-scope integer atomics at the memory side: coherent across the XCDs' L2s, and order-free).  Thread j starts
    // at copy j % R, so the threads of a wave read different banks.
    for (int j = tid; j < p.cols; j += LR_UB_THREADS) {
        long long v = 0;
        if (j < p.n32) {
            const int* h = reinterpret_cast<const int*>(h_bs) + ((size_t)j << rshift);
            for (int r = 0; r < R; ++r) v += h[(r + j) & (R - 1)];
        } else {
            const unsigned long long* h = h_hi + ((size_t)(j - p.n32) << rshift);
            for (int r = 0; r < R; ++r) v += (long long)h[(r + j) & (R - 1)];
        }
        if (v) {
            // a RETURNING atomic, its old value consumed: the value comes back from where the add was performed (the
            // memory side, agent scope), so once it is here the add is visible to every XCD - whatever a no-return
            // atomic's completion count may mean on the way there
            const long long old = __hip_atomic_fetch_add(acc + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));
        }
    }
    // The block that takes the last ticket turns the accumulators into the outputs.  No release / acquire fences (an
    // agent-scope fence writes back and invalidates the XCD's whole L2: ~1.3 us each, serialised per XCD): every add of
    // this block has returned its old value (above), the barrier orders the block's waves before its ticket, and the
    // last block reads the accumulators with agent-scope (sc1) loads that do not hit in its own L2.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    long long* tot = reinterpret_cast<long long*>(ub_lds);          // [cols], the histograms are dead
    for (int j = tid; j < p.cols; j += LR_UB_THREADS)
        tot[j] = __hip_atomic_load(acc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long long* t_bs = tot;                 // [W + 2]
    const long long* t_be = tot + (W + 2);       // [W + 2]
    const long long* t_adj = tot + 2 * (W + 2);  // [W]
    const long long* t_hi = tot + p.n32;         // [W]
    const long long* t_lo = t_hi + W;            // [W]
    if (tid < LR_WAVE) {
        // A[b] = sum_{j <= b} (bs[j] - be[j + 1]) - be[0] + adj[b]: lane l owns the bins [l * per, (l + 1) * per)
        const int per = (W + LR_WAVE - 1) / LR_WAVE;
        const int b0 = tid * per, b1 = min(b0 + per, W);
        long long local = 0;
        for (int b = b0; b < b1; ++b) local += t_bs[b] - t_be[b + 1];
        long long incl = local;
        for (int d = 1; d < LR_WAVE; d <<= 1) {
            const long long up = __shfl_up(incl, d);
            if (tid >= d) incl += up;
        }
        long long run = incl - local - t_be[0];
        for (int b = b0; b < b1; ++b) {
            run += t_bs[b] - t_be[b + 1];
            const long long A = run + t_adj[b];
            sp[b] = t_bs[b + 1];
            ex[b] = t_be[b + 1];
            // exact total in units of 2^-52: A * 2^52 + hi * 2^20 + lo, as a 128-bit integer (q1:q0), rounded to fp64 once
            const unsigned long long hi = (unsigned long long)t_hi[b], lo = (unsigned long long)t_lo[b];
            unsigned long long q0 = (unsigned long long)A << 52, q1 = (unsigned long long)A >> 12;
            unsigned long long add = hi << 20;
            q0 += add, q1 += (hi >> 44) + (q0 < add);
            q0 += lo, q1 += (q0 < lo);
            double v;
            if (q1 == 0 && q0 < (1ull << 53)) {
                v = (double)(long long)q0;
            } else {
                const int top = q1 ? 127 - __clzll((long long)q1) : 63 - __clzll((long long)q0);   // index of the leading bit
                const int sh = top - 52;                                                    // bits to drop (>= 1)
                unsigned long long m, rem_hi, rem_lo;      // mantissa (53 bits), dropped bits left-aligned in rem_hi:rem_lo
                if (sh >= 64) {
                    m = q1 >> (sh - 64);
                    rem_hi = (sh == 64) ? q0 : ((q1 << (128 - sh)) | (q0 >> (sh - 64)));
                    rem_lo = (sh == 64) ? 0ull : (q0 << (128 - sh));
                } else {
                    m = (q1 << (64 - sh)) | (q0 >> sh);
                    rem_hi = q0 << (64 - sh);
                    rem_lo = 0ull;
                }
                const unsigned long long half = 1ull << 63;
                if (rem_hi > half || (rem_hi == half && (rem_lo != 0ull || (m & 1ull)))) ++m;   // to nearest, ties to even
                v = ldexp((double)(long long)m, sh);
            }
            br[b] = ldexp(v, -52);
        }
    }
}

extern "C" int64_t lr_bin_unit_events_workspace_bytes(int64_t n, int32_t n_bins) {
    if (n < 1 || n_bins < 1 || n_bins > LR_MAX_BINS) return LR_ERR_SIZE;
    const lr_ub_shape p = lr_ub_plan(n_bins);
    return 256 + (int64_t)p.cols * 8;
}

extern "C" int lr_bin_unit_events(const double* ts, const double* te, int64_t n, double t0, int32_t n_bins,
                                  int64_t* sp_events, int64_t* ex_events, double* br_length, void* workspace,
                                  int64_t workspace_bytes, void* stream_) {
    if (!ts || !te || !sp_events || !ex_events || !br_length || !workspace) return LR_ERR_NULL;
    if (n < 1 || n_bins < 1 || n_bins > LR_MAX_BINS) return LR_ERR_SIZE;
    if (t0 != floor(t0) || fabs(t0) > 1e9) return LR_ERR_T0;
    const lr_ub_shape p = lr_ub_plan(n_bins);
    const size_t ws_bytes = 256 + (size_t)p.cols * 8;      // the ticket, then one 64-bit accumulator per column
    if ((int64_t)ws_bytes > workspace_bytes) return LR_ERR_WORKSPACE;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (p.lds_bytes > 64 * 1024) {
        // (per call: the attribute belongs to the function on the CURRENT device, and a process may drive several)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_bin_unit_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    // one block of 16 waves per CU (the histograms fill most of its LDS); short inputs: >= 8 lineages per thread
    const long long unit = 2 * LR_UB_THREADS;
    // (as many blocks as fit a CU's LDS side by side: two when the histograms take less than half of it)
    static const int bpc_env = getenv("LR_UB_BLOCKS_PER_CU") ? atoi(getenv("LR_UB_BLOCKS_PER_CU")) : 0;
    const int per_cu = bpc_env > 0 ? bpc_env : (p.lds_bytes <= 76 * 1024 ? 2 : 1);
    long long blocks = min((long long)min(n_cu * per_cu, LR_UB_MAX_BLOCKS), (n + 4 * unit - 1) / (4 * unit));
    if (blocks < 1) blocks = 1;
    const long long chunk = lr_align_up64((n + blocks - 1) / blocks, unit);
    blocks = (n + chunk - 1) / chunk;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    // ticket and accumulators start at zero: the workspace is the caller's, uninitialised memory
    hipError_t e = hipMemsetAsync(ws, 0, ws_bytes, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lr_bin_unit_kernel, dim3((unsigned)blocks), dim3(LR_UB_THREADS), p.lds_bytes, stream, ts, te,
                       (long long)n, t0, p, chunk, (long long*)(ws + 256), (unsigned int*)ws, (long long*)sp_events,
                       (long long*)ex_events, br_length);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// bench / profile hook: the yardstick for the two kernels above and for lr_scan_*_kernel - a kernel that only READS two
// n-double arrays and computes nothing, in the access shape of lr_bin_unit_kernel (the fastest of the streaming kernels: one
// 1024-thread block per CU over a contiguous chunk, two 16-byte loads per array and thread in flight).  What it reaches on
// a box is what "the HBM roofline" means for a pass over ts / te on that box (bench.py abi.stream2_GBs;
// scratch/ubench/stream2.hip tried the other shapes: 2048 blocks of 256 threads in grid stride read 5-10 % slower at 1e8).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lr_debug_stream2_kernel(const double2* __restrict__ a, const double2* __restrict__ b,
                                                                long long n2, long long chunk2, double* __restrict__ out) {
    const long long start = (long long)blockIdx.x * chunk2, end = min(start + chunk2, n2);
    long long i = start + threadIdx.x;
    double acc = 0.0;
    double2 x0 = make_double2(0.0, 0.0), y0 = x0, x1 = x0, y1 = x0;
    if (i < end) x0 = a[i], y0 = b[i];
    if (i + 1024 < end) x1 = a[i + 1024], y1 = b[i + 1024];
    while (i < end) {
        const double2 xc = x0, yc = y0;
        x0 = x1, y0 = y1;
        if (i + 2048 < end) x1 = a[i + 2048], y1 = b[i + 2048];
        acc += (xc.x + xc.y) + (yc.x + yc.y);
        i += 1024;
    }
    if (acc == 123.456) out[0] = acc;       // (never true for the bench's inputs: the loads stay, nothing is written)
}

extern "C" int lr_debug_stream2(const double* a, const double* b, int64_t n, double* out, void* stream_) {
    if (!a || !b || !out) return LR_ERR_NULL;
    if (n < 2 || ((((uintptr_t)a) | ((uintptr_t)b)) & 15)) return LR_ERR_SIZE;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long n2 = n / 2;
    const long long chunk2 = lr_align_up64((n2 + n_cu - 1) / n_cu, 1024);
    const long long blocks = (n2 + chunk2 - 1) / chunk2;
    hipLaunchKernelGGL(lr_debug_stream2_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream_, (const double2*)a,
                       (const double2*)b, n2, chunk2, out);
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
