// lr_scan.h - the fast lineage-scan block body, shared by the stand-alone scan kernel
// (lr_loglik.hip) and the fused scan+chain-step kernel of the engine (lr_mcmc.hip).
#pragma once
#include <climits>
#include "lr_device.h"
#include "lr_internal.h"

// diagnostic build only (-DLR_DIAG): per-block phase stamps (100 MHz wall clock) for timeline analysis
#ifdef LR_DIAG
static __device__ unsigned long long lr_diag_buf[8192 * 8];
#define LR_STAMP(blk, k)                                                                          \
    if (threadIdx.x == 0 && (blk) < 8192) lr_diag_buf[(blk) * 8 + (k)] = wall_clock64()
#else
#define LR_STAMP(blk, k)
#endif

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

// The same for a PAIR of lineages with the LDS gathers issued in batches: the entries of B chains x 2 lineages x {S, E}
// (4 B ds_read_b128) are requested back to back and consumed behind ONE wait, instead of two gathers - wait - four fp64
// instructions per (lineage, chain) as the compiler schedules the plain loop (a wave then spends an LDS round trip per
// chain: 16 chains x 2 lineages x ~100 cycles, which four waves per SIMD do not hide - the wide kernel ran at 0.39 of the
// HBM peak, 64 % of what its LDS traffic allows).  Same operations in the same order per chain: the sums are unchanged.
template <int CB, int H, int B>
__device__ __forceinline__ void lr_score_pair_batched(double2 s2, double2 e2, double t0, int n_bins, const char* __restrict__ lds,
                                                      double (&acc)[CB]) {
    static_assert(CB % B == 0, "whole batches");
    const double fl0 = floor(s2.x), fl1 = floor(s2.y), ce0 = ceil(e2.x), ce1 = ceil(e2.y);
    const int a0 = min(max(__double2int_rz(fl0 - t0), -1), n_bins), a1 = min(max(__double2int_rz(fl1 - t0), -1), n_bins);
    const int b0 = min(max(__double2int_rz(ce0 - t0), 0), n_bins + 1), b1 = min(max(__double2int_rz(ce1 - t0), 0), n_bins + 1);
    const double fs0 = s2.x - fl0, fs1 = s2.y - fl1;
    const double fe0 = (e2.x - ce0) + 1.0, fe1 = (e2.y - ce1) + 1.0;
    const char* pS0 = lds + ((a0 + 1) << 4);
    const char* pS1 = lds + ((a1 + 1) << 4);
    const char* pE0 = lds + (b0 << 4) + H * 16;
    const char* pE1 = lds + (b1 << 4) + H * 16;
#pragma unroll
    for (int c0 = 0; c0 < CB; c0 += B) {
        double2 S0[B], E0[B], S1[B], E1[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            S0[j] = *reinterpret_cast<const double2*>(pS0 + (c0 + j) * (2 * H * 16));
            E0[j] = *reinterpret_cast<const double2*>(pE0 + (c0 + j) * (2 * H * 16));
            S1[j] = *reinterpret_cast<const double2*>(pS1 + (c0 + j) * (2 * H * 16));
            E1[j] = *reinterpret_cast<const double2*>(pE1 + (c0 + j) * (2 * H * 16));
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            double t = S0[j].x + E0[j].x;
            t = fma(fs0, S0[j].y, t);
            t = fma(fe0, E0[j].y, t);
            acc[c0 + j] += t;
            double u = S1[j].x + E1[j].x;
            u = fma(fs1, S1[j].y, u);
            u = fma(fe1, E1[j].y, u);
            acc[c0 + j] += u;
        }
    }
}

// Lineages sorted by birth time (how the reference's input files are written): the 128 lineages a wave scores per trip
// nearly always share their birth bin, so their birth-side entries (S, slope) of the CB chains are the same 16 bytes for
// every lane.  The wave then keeps them in SCALAR registers (read once per bin through lane 0) and gathers only the
// death-side entries: half the ds_read_b128 per lineage - the wide kernel, bound by exactly those gathers (32 per lineage
// at 16 chains = 2 cycles of the CU's LDS path against 1.2 for the HBM stream), ran at 0.40 of the HBM peak.  The same
// values enter the same operations in the same order: the sums are bit for bit those of the gather path, which every
// trip whose lanes disagree takes as before (the test is wave-uniform: no divergence).
#ifndef LR_SCAN_BIRTH_CACHE
#define LR_SCAN_BIRTH_CACHE 1
#endif
#ifndef LR_UNIT_BIRTH_CACHE
#define LR_UNIT_BIRTH_CACHE 1    /* the same in the unit-resolution scan */
#endif
#ifndef LR_SCAN_CACHED_BATCH
#define LR_SCAN_CACHED_BATCH 4   /* chains whose death-side entries (of two lineages) are gathered behind one wait */
#endif
template <int CB>
struct lr_birth_cache {
    int bin;            // birth bin the entries below belong to; INT_MIN: none yet
    double2 S[CB];      // wave-uniform (the compiler keeps them in SGPRs: they come out of v_readfirstlane)
};

__device__ __forceinline__ double lr_uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int CB, int H>
__device__ __forceinline__ void lr_birth_cache_fill(lr_birth_cache<CB>& bc, int bin, const char* __restrict__ lds) {
    bc.bin = bin;
    const char* pS = lds + ((bin + 1) << 4);
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(pS + c * (2 * H * 16));
        bc.S[c].x = lr_uniform_f64(v.x), bc.S[c].y = lr_uniform_f64(v.y);
    }
}

// a pair of lineages whose birth bin is the cached one in every lane of the wave: death-side gathers only, B chains per batch
template <int CB, int H, int B>
__device__ __forceinline__ void lr_score_pair_cached(double2 s2, double2 e2, double t0, int n_bins, const char* __restrict__ lds,
                                                     const lr_birth_cache<CB>& bc, double (&acc)[CB]) {
    static_assert(CB % B == 0, "whole batches");
    const double fl0 = floor(s2.x), fl1 = floor(s2.y), ce0 = ceil(e2.x), ce1 = ceil(e2.y);
    const int b0 = min(max(__double2int_rz(ce0 - t0), 0), n_bins + 1), b1 = min(max(__double2int_rz(ce1 - t0), 0), n_bins + 1);
    const double fs0 = s2.x - fl0, fs1 = s2.y - fl1;
    const double fe0 = (e2.x - ce0) + 1.0, fe1 = (e2.y - ce1) + 1.0;
    const char* pE0 = lds + (b0 << 4) + H * 16;
    const char* pE1 = lds + (b1 << 4) + H * 16;
#pragma unroll
    for (int c0 = 0; c0 < CB; c0 += B) {
        double2 E0[B], E1[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            E0[j] = *reinterpret_cast<const double2*>(pE0 + (c0 + j) * (2 * H * 16));
            E1[j] = *reinterpret_cast<const double2*>(pE1 + (c0 + j) * (2 * H * 16));
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            double t = bc.S[c0 + j].x + E0[j].x;
            t = fma(fs0, bc.S[c0 + j].y, t);
            t = fma(fe0, E0[j].y, t);
            acc[c0 + j] += t;
            double u = bc.S[c0 + j].x + E1[j].x;
            u = fma(fs1, bc.S[c0 + j].y, u);
            u = fma(fe1, E1[j].y, u);
            acc[c0 + j] += u;
        }
    }
}

// -> true if both lineages of every lane of the wave are born in one and the same bin (then *bin is it)
__device__ __forceinline__ bool lr_pair_birth_uniform(double2 s2, double t0, int n_bins, int* bin) {
    const int a0 = min(max(__double2int_rz(floor(s2.x) - t0), -1), n_bins), a1 = min(max(__double2int_rz(floor(s2.y) - t0), -1), n_bins);
    const int ua = __builtin_amdgcn_readfirstlane(a0);
    *bin = ua;
    return __all(a0 == ua && a1 == ua) != 0;
}

// XCD-aware block -> (chain group, tile) map.  Blocks are dealt round-robin over the 8 XCDs (block b and
// b + 8 share an XCD and its L2, MI355X_MICROARCH.md "Workgroup dispatch"), and L2 does not survive a kernel
// boundary, so every XCD re-fetches what its blocks stage.  Keeping ALL tiles of a chain group on ONE XCD
// means the group's 35 KB of tables leave the Infinity Cache once per launch instead of once per tile.
// Placement only changes speed, never results.
__device__ __forceinline__ void lr_xcd_remap(int sb, int tiles, int groups, int* group, int* tile) {
    if ((groups & 7) == 0) {
        const int x = sb & 7, idx = sb >> 3;
        *group = x + 8 * (idx / tiles);
        *tile = idx % tiles;
    } else {
        *tile = sb % tiles;
        *group = sb / tiles;
    }
}

// what a scan body does between issuing its first loads of ts / te and staging the tables: nothing in the launch-based
// kernels; the resident streaming kernel (lr_stream_kernel) waits there for the tables of its iteration to be published
// ... and how it reads a table entry and stores a partial sum (plain accesses; that kernel's go to the agent's point of
// coherence, see lr_stream.hip)
struct lr_no_wait {
    __device__ __forceinline__ bool operator()() const { return false; }   // true: the tables already stand in LDS
    static __device__ __forceinline__ double2 load16(const double2* p) { return *p; }
    static __device__ __forceinline__ void store_partial(double* p, double v) { *p = v; }
};

// Block reduction of the per-thread accumulators through LDS (no cross-lane shuffles, which run on the LDS
// crossbar and cost ~3.5 us per block as 17 dependent steps): every thread stores its CB sums ([chain][thread],
// conflict-free), THREADS/CB threads per chain each add CB of them in a fixed order, then one thread per chain
// adds those.  Needs CB*THREADS + THREADS doubles of LDS (the staged tables are dead by then).
template <int CB, int T = LR_SCAN_THREADS, class IO = lr_no_wait>
__device__ __forceinline__ void lr_block_reduce_chains(const double (&acc)[CB], double* red, int tid, int nvalid,
                                                       double* __restrict__ out /* chain0's row of partials + tile */,
                                                       size_t chain_stride) {
    constexpr int TPC = T / CB;            // threads per chain in stage 1
#pragma unroll
    for (int c = 0; c < CB; ++c) red[c * T + tid] = acc[c];
    __syncthreads();
    const int c1 = tid / TPC, j = tid % TPC;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < CB; ++k) s += red[c1 * T + j + TPC * k];
    red[CB * T + tid] = s;                 // = [c1][j]
    __syncthreads();
    if (tid < nvalid) {
        double t = 0.0;
        for (int k = 0; k < TPC; ++k) t += red[CB * T + tid * TPC + k];
        IO::store_partial(out + (size_t)tid * chain_stride, t);
    }
}

// One block: tile `tile` of the lineages x chains [chain0, chain0+CB) of the `n_chains` whose tables start at
// `tables`; partial sums go to partials[chain * partial_stride + tile].  T threads; DEPTH pairs of lineages (32 B each) per
// thread in flight: the wide form (CB = 16: 70 KB of tables at H = 136, two 512-thread blocks per CU) keeps two, so that a
// CU has 64 KB on its way although only 16 waves fit.
// stage the CB general tables of chains [chain0, chain0 + nvalid): all 16-byte global loads are issued back to back (one
// latency), then written; chain slots past nvalid are zeroed
template <int CB, int H, int T, class IO>
__device__ __forceinline__ void lr_stage_fast_tables(double2* lds, const double2* __restrict__ tables, int chain0, int nvalid, int tid) {
    constexpr int STRIDE = 2 * H;
    const double2* src = tables + (size_t)chain0 * STRIDE;
    const int n_valid_entries = nvalid * STRIDE;
    constexpr int NI = (CB * STRIDE + T - 1) / T;
    double2 buf[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = tid + k * T;
        buf[k] = IO::load16(src + min(i, n_valid_entries - 1));
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = tid + k * T;
        if (i < CB * STRIDE) lds[i] = (i < n_valid_entries) ? buf[k] : make_double2(0.0, 0.0);
    }
}

template <int CB, int H, int T = LR_SCAN_THREADS, int DEPTH = 1, int BATCH = 0, class WAIT = lr_no_wait>
__device__ __forceinline__ void lr_scan_fast_body(double2* lds, int tile, int chain0, const double* __restrict__ ts,
                                                  const double* __restrict__ te, long long n, double t0, int n_bins,
                                                  const double2* __restrict__ tables, int n_chains, long long chunk,
                                                  double* __restrict__ partials, int partial_stride, WAIT wait_tables = WAIT()) {
    static_assert(DEPTH == 1 || DEPTH == 2, "one or two pairs in flight");
    constexpr int STRIDE = 2 * H;
    const int tid = threadIdx.x;
    const int nvalid = min(CB, n_chains - chain0);
    const long long start = (long long)tile * chunk;
    const long long end = min(start + chunk, n);
    const bool aligned = ((((uintptr_t)ts) | ((uintptr_t)te)) & 15) == 0;
    long long i = start + 2 * tid;
    // first pair(s) of lineages: in flight while the tables are staged.  (One pair - 32 B - per thread in flight is enough
    // at T = 256: with four to eight blocks per CU a second pair measured the same at 1e7 - 3e7 lineages, round 4.)
    double2 s2 = make_double2(0.0, 0.0), e2 = s2, s3 = s2, e3 = s2;
    if (aligned && i + 1 < end) {
        s2 = *reinterpret_cast<const double2*>(ts + i);
        e2 = *reinterpret_cast<const double2*>(te + i);
        if (DEPTH == 2 && i + 2 * T + 1 < end) {
            s3 = *reinterpret_cast<const double2*>(ts + i + 2 * T);
            e3 = *reinterpret_cast<const double2*>(te + i + 2 * T);
        }
    }
    if (!wait_tables()) lr_stage_fast_tables<CB, H, T, WAIT>(lds, tables, chain0, nvalid, tid);
    __syncthreads();

    double acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.0;
    const char* lbase = reinterpret_cast<const char*>(lds);
    lr_birth_cache<CB> bc;
    bc.bin = INT_MIN;
#pragma unroll
    for (int c = 0; c < CB; ++c) bc.S[c] = make_double2(0.0, 0.0);
    if (aligned) {
        while (i + 1 < end) {
            const double2 sc = s2, ec = e2;
            if (DEPTH == 2) {
                s2 = s3, e2 = e3;
                const long long nx2 = i + 4 * T;
                if (nx2 + 1 < end) {  // prefetch the pair after the next
                    s3 = *reinterpret_cast<const double2*>(ts + nx2);
                    e3 = *reinterpret_cast<const double2*>(te + nx2);
                }
            } else {
                const long long nx = i + 2 * T;
                if (nx + 1 < end) {  // prefetch the next pair
                    s2 = *reinterpret_cast<const double2*>(ts + nx);
                    e2 = *reinterpret_cast<const double2*>(te + nx);
                }
            }
            int ubin = 0;
            if (LR_SCAN_BIRTH_CACHE && CB >= 4 && lr_pair_birth_uniform(sc, t0, n_bins, &ubin)) {
                if (ubin != bc.bin) lr_birth_cache_fill<CB, H>(bc, ubin, lbase);
                lr_score_pair_cached<CB, H, (CB >= LR_SCAN_CACHED_BATCH ? LR_SCAN_CACHED_BATCH : CB)>(sc, ec, t0, n_bins, lbase, bc, acc);
            } else if (BATCH > 0) {
                lr_score_pair_batched<CB, H, (BATCH > 0 ? BATCH : 1)>(sc, ec, t0, n_bins, lbase, acc);
            } else {
                lr_score_lineage_fast<CB, H>(sc.x, ec.x, t0, n_bins, lbase, acc);
                lr_score_lineage_fast<CB, H>(sc.y, ec.y, t0, n_bins, lbase, acc);
            }
            i += 2 * T;
        }
        if (i < end) lr_score_lineage_fast<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
    } else {
        for (; i < end; i += 2 * T) {
            lr_score_lineage_fast<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
            if (i + 1 < end) lr_score_lineage_fast<CB, H>(ts[i + 1], te[i + 1], t0, n_bins, lbase, acc);
        }
    }

    __syncthreads();
    lr_block_reduce_chains<CB, T, WAIT>(acc, reinterpret_cast<double*>(lds), tid, nvalid,
                                  partials + (size_t)chain0 * partial_stride + tile, (size_t)partial_stride);
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
    int group, tile;
    lr_xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.y, &group, &tile);
    lr_scan_fast_body<CB, H>(lds, tile, group * CB, ts, te, n, t0, n_bins, tables, n_chains, chunk, partials,
                             partial_stride);
}

// The wide form of lr_bd_loglik_batch: SIXTEEN chains' general tables (4.3 KB each at H = 136) per pass over ts / te -
// half the passes of the CB = 8 kernel for C > 8 chains.  Per lineage 32 ds_read_b128 = 2 cycles of the CU's LDS data
// path: at 16 chains the LDS gathers (256 CUs x 2.4 GHz / 2 = 3.1e11 lineages/s = 4.9 TB/s of ts / te) and HBM are in
// balance, so 32 chains per pass would take as long as two passes of 16.
#define LR_SCAN_WIDE_THREADS 512
template <int H>
#ifndef LR_SCAN_WIDE_BATCH
#define LR_SCAN_WIDE_BATCH 4
#endif
__global__ __launch_bounds__(LR_SCAN_WIDE_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void lr_scan_wide_kernel(const double* __restrict__ ts,
                                                                            const double* __restrict__ te, long long n,
                                                                            double t0, int n_bins,
                                                                            const double2* __restrict__ tables,
                                                                            int n_chains, long long chunk,
                                                                            double* __restrict__ partials,
                                                                            int partial_stride) {
    extern __shared__ double2 lds[];
    int group, tile;
    lr_xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.y, &group, &tile);
    lr_scan_fast_body<16, H, LR_SCAN_WIDE_THREADS, 2, LR_SCAN_WIDE_BATCH>(lds, tile, group * 16, ts, te, n, t0, n_bins, tables, n_chains,
                                                                          chunk, partials, partial_stride);
}

// ------------------------------------------------------------------------------------------
// unit-resolution path: every lineage has the same in-bin fractions (fs0, fe0), which the table
// builder has folded into 8-byte entries  S'[b+1] = logB_b + cum_b + fs0*R_b,
// E'[b+1] = logD_b - cum_b - fe0*R_b.  Two chains share one 16-byte entry, so per lineage and PAIR of chains:
// two ds_read_b128 gathers + 4 fp64 adds (b128 reads reach the LDS rate with few waves per SIMD, b64 do not).
// Layout per group of CB chains: [CB/2 pairs][2 (S', E')][H] double2 = (chain 2p, chain 2p+1).
// ------------------------------------------------------------------------------------------
// ... and with the wave's shared birth-bin entries in scalar registers (lr_birth_cache above; CB / 2 pair entries)
template <int CB, int H>
__device__ __forceinline__ void lr_unit_cache_fill(lr_birth_cache<(CB + 1) / 2>& bc, int bin, const char* __restrict__ lds) {
    bc.bin = bin;
    const char* pS = lds + ((bin + 1) << 4);
#pragma unroll
    for (int p = 0; p < CB / 2; ++p) {
        const double2 v = *reinterpret_cast<const double2*>(pS + p * (2 * H * 16));
        bc.S[p].x = lr_uniform_f64(v.x), bc.S[p].y = lr_uniform_f64(v.y);
    }
}
template <int CB, int H>
__device__ __forceinline__ void lr_score_lineage_unit_cached(double e, double t0, int n_bins, const char* __restrict__ lds,
                                                             const lr_birth_cache<(CB + 1) / 2>& bc, double (&acc)[CB]) {
    const int b = min(max(__double2int_rz(ceil(e) - t0), 0), n_bins + 1);
    const char* pE = lds + (b << 4) + H * 16;
#pragma unroll
    for (int p = 0; p < CB / 2; ++p) {
        const double2 E = *reinterpret_cast<const double2*>(pE + p * (2 * H * 16));
        acc[2 * p] += bc.S[p].x + E.x;
        acc[2 * p + 1] += bc.S[p].y + E.y;
    }
}

template <int CB, int H>
__device__ __forceinline__ void lr_score_lineage_unit(double s, double e, double t0, int n_bins,
                                                      const char* __restrict__ lds, double (&acc)[CB]) {
    const int a = min(max(__double2int_rz(floor(s) - t0), -1), n_bins);
    const int b = min(max(__double2int_rz(ceil(e) - t0), 0), n_bins + 1);
    const char* pS = lds + ((a + 1) << 4);
    const char* pE = lds + (b << 4) + H * 16;
    if (CB == 1) {
        acc[0] += reinterpret_cast<const double2*>(pS)->x + reinterpret_cast<const double2*>(pE)->x;
        return;
    }
#pragma unroll
    for (int p = 0; p < CB / 2; ++p) {
        const double2 S = *reinterpret_cast<const double2*>(pS + p * (2 * H * 16));
        const double2 E = *reinterpret_cast<const double2*>(pE + p * (2 * H * 16));
        acc[2 * p] += S.x + E.x;
        acc[2 * p + 1] += S.y + E.y;
    }
}

#ifndef LR_UNIT_DEPTH
#define LR_UNIT_DEPTH 1   /* 32-byte (ts, te) pairs a thread of the unit-resolution scan keeps in flight (2: measured, no gain - the kernel
                             runs at 0.97 of the read-only yardstick at 1e8 lineages with one) */
#endif

// stage the pair tables of the group of CB chains from chain0 on: the whole group region - pair tables of chains past
// n_chains hold zeros (the workspace is zeroed)
template <int CB, int H, class IO>
__device__ __forceinline__ void lr_stage_unit_tables(double2* lds, const double2* __restrict__ tables, int chain0, int tid) {
    constexpr int GROUP_ENTRIES = (CB < 2 ? 2 : CB) * H;
    const double2* src = tables + (size_t)chain0 * H;
    constexpr int NI = (GROUP_ENTRIES + LR_SCAN_THREADS - 1) / LR_SCAN_THREADS;
    double2 buf[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j = tid + k * LR_SCAN_THREADS;
        buf[k] = IO::load16(src + min(j, GROUP_ENTRIES - 1));
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j = tid + k * LR_SCAN_THREADS;
        if (j < GROUP_ENTRIES) lds[j] = buf[k];
    }
}

template <int CB, int H, int DEPTH = LR_UNIT_DEPTH, class WAIT = lr_no_wait>
__device__ __forceinline__ void lr_scan_unit_body(double2* lds, int tile, int chain0, const double* __restrict__ ts,
                                                  const double* __restrict__ te, long long n, double t0, int n_bins,
                                                  const double2* __restrict__ tables, int n_chains, long long chunk,
                                                  double* __restrict__ partials, int partial_stride, WAIT wait_tables = WAIT()) {
    constexpr int STRIDE = H;  // double2 entries per chain; a group of CB chains owns max(CB,2)*H entries
    constexpr int GROUP_ENTRIES = (CB < 2 ? 2 : CB) * STRIDE;
    const int tid = threadIdx.x;
    const int diag_blk = blockIdx.x + blockIdx.y * gridDim.x;
    (void)diag_blk;
    LR_STAMP(diag_blk, 0);
    const int nvalid = min(CB, n_chains - chain0);
    const long long start = (long long)tile * chunk;
    const long long end = min(start + chunk, n);
    const bool aligned = ((((uintptr_t)ts) | ((uintptr_t)te)) & 15) == 0;
    long long i = start + 2 * tid;
    double2 s2[DEPTH], e2[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        s2[d] = make_double2(0.0, 0.0), e2[d] = make_double2(0.0, 0.0);
        const long long j = i + (long long)d * 2 * LR_SCAN_THREADS;
        if (aligned && j + 1 < end) {
            s2[d] = *reinterpret_cast<const double2*>(ts + j);
            e2[d] = *reinterpret_cast<const double2*>(te + j);
        }
    }
    if (!wait_tables()) lr_stage_unit_tables<CB, H, WAIT>(lds, tables, chain0, tid);
    __syncthreads();
    LR_STAMP(diag_blk, 1);

    double acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = 0.0;
    const char* lbase = reinterpret_cast<const char*>(lds);
    lr_birth_cache<(CB + 1) / 2> bc;
    bc.bin = INT_MIN;
#pragma unroll
    for (int p = 0; p < (CB + 1) / 2; ++p) bc.S[p] = make_double2(0.0, 0.0);
    if (aligned) {
        // DEPTH pairs in flight per thread, in a ring of registers: slot d holds the pair of trip (DEPTH m + d)
        bool more = i + 1 < end;
        while (more) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (more) {
                    const double2 sc = s2[d], ec = e2[d];
                    const long long nx = i + (long long)DEPTH * 2 * LR_SCAN_THREADS;
                    if (nx + 1 < end) {
                        s2[d] = *reinterpret_cast<const double2*>(ts + nx);
                        e2[d] = *reinterpret_cast<const double2*>(te + nx);
                    }
                    int ubin = 0;
                    if (LR_UNIT_BIRTH_CACHE && CB >= 4 && lr_pair_birth_uniform(sc, t0, n_bins, &ubin)) {
                        if (ubin != bc.bin) lr_unit_cache_fill<CB, H>(bc, ubin, lbase);
                        lr_score_lineage_unit_cached<CB, H>(ec.x, t0, n_bins, lbase, bc, acc);
                        lr_score_lineage_unit_cached<CB, H>(ec.y, t0, n_bins, lbase, bc, acc);
                    } else {
                        lr_score_lineage_unit<CB, H>(sc.x, ec.x, t0, n_bins, lbase, acc);
                        lr_score_lineage_unit<CB, H>(sc.y, ec.y, t0, n_bins, lbase, acc);
                    }
                    i += 2 * LR_SCAN_THREADS;
                    more = i + 1 < end;
                }
            }
        }
        if (i < end) lr_score_lineage_unit<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
    } else {
        for (; i < end; i += 2 * LR_SCAN_THREADS) {
            lr_score_lineage_unit<CB, H>(ts[i], te[i], t0, n_bins, lbase, acc);
            if (i + 1 < end) lr_score_lineage_unit<CB, H>(ts[i + 1], te[i + 1], t0, n_bins, lbase, acc);
        }
    }

    LR_STAMP(diag_blk, 2);
    __syncthreads();
    LR_STAMP(diag_blk, 3);
    lr_block_reduce_chains<CB, LR_SCAN_THREADS, WAIT>(acc, reinterpret_cast<double*>(lds), tid, nvalid,
                               partials + (size_t)chain0 * partial_stride + tile, (size_t)partial_stride);
    LR_STAMP(diag_blk, 4);
#ifdef LR_DIAG
    if (threadIdx.x == 0 && diag_blk < 8192) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        lr_diag_buf[diag_blk * 8 + 5] = hw;
        lr_diag_buf[diag_blk * 8 + 6] = xcc;
        lr_diag_buf[diag_blk * 8 + 7] = ((unsigned long long)tile << 32) | (unsigned)chain0;
    }
#endif
}

template <int CB, int H>
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_scan_unit_kernel(const double* __restrict__ ts,
                                                                       const double* __restrict__ te, long long n,
                                                                       double t0, int n_bins,
                                                                       const double2* __restrict__ tables,
                                                                       int n_chains, long long chunk,
                                                                       double* __restrict__ partials,
                                                                       int partial_stride) {
    extern __shared__ double2 lds[];
    int group, tile;
    lr_xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.y, &group, &tile);
    lr_scan_unit_body<CB, H>(lds, tile, group * CB, ts, te, n, t0, n_bins, tables, n_chains, chunk, partials,
                             partial_stride);
}

// ------------------------------------------------------------------------------------------
// persistent engines: the lineages as packed groups (lr_pack.hip), pair tables resident in LDS
// ------------------------------------------------------------------------------------------
#define LR_GRP 14      /* lineages per 16-byte group: byte 0 birth index, byte 1 count, bytes 2..15 death indices */

// byte k (0..15) of a packed group as an LDS byte offset into a table of (1 << SH)-byte entries
template <int SH>
__device__ __forceinline__ unsigned int lr_grp_off(const uint4& w, int k) {
    const unsigned int v = (k < 4) ? w.x : (k < 8 ? w.y : (k < 12 ? w.z : w.w));
    if ((k & 3) == 0) {
        // byte 0 of a word: the compiler would shift and mask (two instructions); the byte-select form of the shift does
        // it in one, as it does by itself for bytes 1..3 (the scan loop is bound by vector instruction issue)
        unsigned int r;
        const unsigned int sh = SH;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(v));
        return r;
    }
    const int sh = 8 * (k & 3) - SH;                                   // (v >> 8j) & 0xff, then << SH
    const unsigned int m = 0xffu << SH;
    return (sh >= 0 ? (v >> sh) : (v << -sh)) & m;
}

// ---- unit resolution: pair slots ------------------------------------------------------------------------------------
// Group format (lr_pack.hip, unit-resolution data): uint4 = a 16-bit header (birth index a << 4 | number of lineages:
// masked with 0xfff0 it IS the byte offset of the birth entry), then LR_SLOTS = 7 sixteen-bit entry BYTE OFFSETS (entry index << 4) into the
// block's pair table.  A slot holds ONE lineage (entry H + j, its death
// entry E[j]) or TWO consecutive lineages of the run whose death entries are j and j + d, 0 <= d <= 3 (entry
// (2 + d) H + j, the pair sum E[j] + E[j + d]); padding slots point at E[0] = 0.  The pair table in LDS therefore has six
// planes of H entries: S, E and the four pair-sum planes, which every block derives from its E plane (lr_pair_planes_*);
// only S and E exist in global memory.  Sorted lineages of one birth bin die in nearly sorted order, so almost every slot
// is a pair (cfg4: 49,953 pairs and 94 singles for 100,000 lineages): 8 gathers and 17 fp64 operations score 14
// lineages x 2 chains, every lineage through its own (birth, death) entry.
#define LR_SLOTS 7
#define LR_PAIR_DMAX 3
#define LR_UNIT_PLANES 6                /* S, E, E2[0..3] */

// byte offset of the table entry a slot points at = 16-bit field `hi` of a word: lr_pack.hip stores the slots as BYTE
// offsets (entry index << 4; six planes of at most 520 entries stay below 2^16), so a field is used as it is.  What the
// extraction costs (scratch/ubench/issue_rate.hip, profiles/r04_ubench.txt; cycles per wave64 instruction per SIMD at
// >= 2 waves per SIMD): v_and_b32 with a literal 2.4, v_lshrrev_b32 2.3 - against 4.2 for any SDWA form (the word-select
// shift that used to turn an entry INDEX into an offset), for v_lshlrev_b32, v_bfe_u32, v_and_or_b32, v_perm_b32 and for
// every fp64 instruction.  The scan loop is bound by vector instruction issue.
__device__ __forceinline__ unsigned int lr_word_off16(unsigned int v, int hi) { return hi ? (v >> 16) : (v & 0xffffu); }

// A 16-byte global load the compiler does not see (base: uniform pointer, off: 32-bit byte offset of the lane), and the
// wait that makes its result - and every load issued before it - usable.  The scan loops issue the NEXT group's load
// right after decoding the current one, into the same registers, and wait at the top of the next trip; written as
// plain loads the compiler folds the two into one load at the loop top and waits for it on the spot.
typedef unsigned int lr_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lr_gload16_async(lr_u32x4& w, const char* base, unsigned int off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w) : "v"(off), "s"(base));
}
// (all outstanding loads but the newest `LEFT` have landed; `w` ties the uses of the loaded value behind the wait)
// a pointer every lane of the wave holds alike, moved to scalar registers (the base operand of lr_gload16_async)
__device__ __forceinline__ const char* lr_uniform_ptr(const void* p) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
template <int LEFT>
__device__ __forceinline__ void lr_gload_wait(lr_u32x4& w) {
    if (LEFT == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w));
    else asm volatile("s_waitcnt vmcnt(1)" : "+v"(w));
    static_assert(LEFT == 0 || LEFT == 1, "");
}
template <int LEFT>
__device__ __forceinline__ void lr_gload_wait(lr_u32x4& a, lr_u32x4& b, lr_u32x4& c) {
    if (LEFT == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
    else asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b), "+v"(c));
    static_assert(LEFT == 0 || LEFT == 1, "");
}

// the loads a scan leaves in flight when it returns (the idle prefetch of its last trip), and their wait
struct lr_scan_tail {
    lr_u32x4 w, f0, f1, f2;
};
#ifndef LR_SCAN_DEPTH2
#define LR_SCAN_DEPTH2 0     /* unit-resolution ZERO_TAIL scans (lr_persist_scan_pair), experiments of round 5: 1 = two groups per
                                lane in flight (cfg4 6.28-6.30 us per iteration against 6.27-6.33: the scan does not wait for its
                                group loads), 2 = two groups per TRIP (6.70: a wasted half trip at 8.5 trips, more registers) */
#endif
__device__ __forceinline__ void lr_scan_drain(lr_scan_tail& t, bool with_fractions = false) {
    lr_gload_wait<0>(t.w);
    if (with_fractions) lr_gload_wait<0>(t.f0, t.f1, t.f2);
#if LR_SCAN_DEPTH2
    else lr_gload_wait<0>(t.f0);       // (unit resolution: the second idle load of a two-deep scan; a no-op wait otherwise)
#endif
}

// Scan of `n8` groups against ONE pair table (unit resolution, six planes of 16-byte entries = the two chains' values)
// by `n_scan` threads, this thread being number `sid`: the inner loop of the persistent engines.  Per group one 16-byte
// load, ONE gather of the birth entry - it enters `count` times - and one gather per slot.  The loop costs what it
// issues, so its control is pared down: a group's fields are decoded first and the NEXT group is then loaded into the
// same registers (in flight while this one is scored, no copy), addressed by a 32-bit byte offset from the uniform base.
// ASYNC: this loop; otherwise lr_persist_scan_pair_slice (plain loads), the better one for scans of one to three trips.
// ZERO_TAIL: the caller scans to the end of the packed lineages, behind which the groups read as zeros (lr_groups_alloc:
// a zero group gathers entry 0 = 0.0 with count 0): the trip count is then the wave's, kept in scalar registers, and
// no lane is ever masked - a lane past the end scores zeros.  Otherwise (a tile or a team member's slice) every lane
// tests its own index.
// A lane's first group of every scan is the same group: a persistent kernel may keep it (and, on general times, its
// fractions) in registers across iterations instead of waiting for the load at the top of each scan.
struct lr_first_group {
    uint4 w;
    uint4 fw[4];
};

// the same scan with plain loads and a test per lane: for a tile or a team member's slice (short scans, often a single
// trip that starts from the first group kept in registers: an idle prefetch behind it would cost a memory round trip)
// ONE: the table holds ONE chain (the speculative kernel with a team per chain: entries (chain, unused)) - the gathers
// read the 8-byte half they need (ds_read_b64: half the LDS bytes of a ds_read_b128) and the second chain's eight fp64
// instructions per group are not issued; acc0 is formed by the same operations in the same order as in the pair form.
template <int H, int UNROLL = 1, bool ONE = false>
__device__ __forceinline__ void lr_persist_scan_pair_slice(const char* __restrict__ lbase, const uint4* __restrict__ idx8,
                                                     long long n8, long long sid, int n_scan, double* acc0_,
                                                     double* acc1_, const lr_first_group* first = nullptr) {
    if (ONE) {
        double acc0 = *acc0_;
        const int n = (int)n8;
        int i = (int)sid;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (first) w = first->w;
        else if (i < n) w = idx8[i];
#pragma unroll UNROLL
        while (i < n) {
            const uint4 cur = w;
            const int nx = i + n_scan;
            if (nx < n) w = idx8[nx];
            const double S = *reinterpret_cast<const double*>(lbase + (cur.x & 0xfff0u));
            const double cnt = (double)(cur.x & 0xfu);
            const double E0 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.x, 1));
            const double E1 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.y, 0));
            const double E2 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.y, 1));
            const double E3 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.z, 0));
            const double E4 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.z, 1));
            const double E5 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.w, 0));
            const double E6 = *reinterpret_cast<const double*>(lbase + lr_word_off16(cur.w, 1));
            const double u0 = ((E0 + E1) + (E2 + E3)) + ((E4 + E5) + E6);
            acc0 += fma(cnt, S, u0);
            i = nx;
        }
        *acc0_ = acc0;
        return;
    }
    double acc0 = *acc0_, acc1 = *acc1_;
    // 32-bit loop arithmetic (fewer than 2^31 groups): a 64-bit compare and add per trip are two instructions each
    const int n = (int)n8;
    int i = (int)sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (first) w = first->w;
    else if (i < n) w = idx8[i];
#pragma unroll UNROLL
    while (i < n) {
        const uint4 cur = w;
        const int nx = i + n_scan;
        if (nx < n) w = idx8[nx];
        const double2 S = *reinterpret_cast<const double2*>(lbase + (cur.x & 0xfff0u));
        const double cnt = (double)(cur.x & 0xfu);
        double2 E[LR_SLOTS];
        E[0] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.x, 1));
        E[1] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.y, 0));
        E[2] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.y, 1));
        E[3] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.z, 0));
        E[4] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.z, 1));
        E[5] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.w, 0));
        E[6] = *reinterpret_cast<const double2*>(lbase + lr_word_off16(cur.w, 1));
        // fixed pairwise tree over the slots, then the birth entry `count` times
        const double u0 = ((E[0].x + E[1].x) + (E[2].x + E[3].x)) + ((E[4].x + E[5].x) + E[6].x);
        const double u1 = ((E[0].y + E[1].y) + (E[2].y + E[3].y)) + ((E[4].y + E[5].y) + E[6].y);
        acc0 += fma(cnt, S.x, u0);
        acc1 += fma(cnt, S.y, u1);
        i = nx;
    }
    *acc0_ = acc0, *acc1_ = acc1;
}

template <int H, int UNROLL = 1, bool ZERO_TAIL = false, bool ASYNC = ZERO_TAIL, bool ONE = false>
__device__ __forceinline__ void lr_persist_scan_pair(const char* __restrict__ lbase, const uint4* __restrict__ idx8,
                                                     long long n8, long long sid, int n_scan, double* acc0_,
                                                     double* acc1_, const lr_first_group* first = nullptr,
                                                     lr_scan_tail* tail = nullptr) {
    static_assert(ASYNC || !ZERO_TAIL, "the wave-uniform trip count comes with the hand-placed loads");
    static_assert(!(ONE && ASYNC), "the one-chain form exists for the plain-load slices only");
    if (!ASYNC) {
        lr_persist_scan_pair_slice<H, UNROLL, ONE>(lbase, idx8, n8, sid, n_scan, acc0_, acc1_, first);
        if (tail) tail->w = lr_u32x4{0u, 0u, 0u, 0u}, tail->f0 = lr_u32x4{0u, 0u, 0u, 0u};
        return;
    }
    double acc0 = *acc0_, acc1 = *acc1_;
#if LR_SCAN_DEPTH2
    if (ZERO_TAIL && !first && tail) {
        // (experiment, off by default) TWO groups in flight per lane: registers A and B take the groups of alternate trips, a
        // wait leaves the younger load in flight (vmcnt(1): vector-memory loads return in order).  The four-chain kernel with
        // its chain steps compiled out (-DLR_P4_NOSTEP=2) takes 2.97 us per phase for 9 trips = 0.33 us a trip, twice what the
        // trip's LDS gathers (35 cycles of the CU's LDS path x 12 waves) and vector instructions need - but not because the scan
        // waited for its group loads: with two in flight nothing changes.
        const int n = __builtin_amdgcn_readfirstlane((int)n8);
        const char* gnext = lr_uniform_ptr(idx8);
        const unsigned int stride_b = (unsigned int)n_scan * 16u;
        const unsigned int off = (unsigned int)sid * 16u;
        int i0 = __builtin_amdgcn_readfirstlane((int)sid);
        bool has = i0 < n;
        lr_u32x4 wA = {0u, 0u, 0u, 0u}, wB = {0u, 0u, 0u, 0u};
        lr_gload16_async(wA, gnext, off);
        gnext += (i0 + n_scan < n) ? stride_b : 0u;       // (past the end: the same group again - a line the wave has just had)
        lr_gload16_async(wB, gnext, off);
#define LR_SCAN_TRIP(W)                                                                                                        \
        {                                                                                                                      \
            lr_gload_wait<1>(W);                                                                                               \
            const unsigned int oS = W.x & 0xfff0u;                                                                             \
            const double cnt = (double)(W.x & 0xfu);                                                                           \
            const unsigned int o0 = lr_word_off16(W.x, 1), o1 = lr_word_off16(W.y, 0), o2 = lr_word_off16(W.y, 1),             \
                               o3 = lr_word_off16(W.z, 0), o4 = lr_word_off16(W.z, 1), o5 = lr_word_off16(W.w, 0),             \
                               o6 = lr_word_off16(W.w, 1);                                                                     \
            i0 += n_scan, has = i0 < n;                                                                                        \
            gnext += (i0 + n_scan < n) ? stride_b : 0u;                                                                        \
            lr_gload16_async(W, gnext, off);               /* the group two trips on, into the registers just decoded */      \
            const double2 S = *reinterpret_cast<const double2*>(lbase + oS);                                                   \
            const double2 E0 = *reinterpret_cast<const double2*>(lbase + o0), E1 = *reinterpret_cast<const double2*>(lbase + o1); \
            const double2 E2 = *reinterpret_cast<const double2*>(lbase + o2), E3 = *reinterpret_cast<const double2*>(lbase + o3); \
            const double2 E4 = *reinterpret_cast<const double2*>(lbase + o4), E5 = *reinterpret_cast<const double2*>(lbase + o5); \
            const double2 E6 = *reinterpret_cast<const double2*>(lbase + o6);                                                  \
            const double u0 = ((E0.x + E1.x) + (E2.x + E3.x)) + ((E4.x + E5.x) + E6.x);                                        \
            const double u1 = ((E0.y + E1.y) + (E2.y + E3.y)) + ((E4.y + E5.y) + E6.y);                                        \
            acc0 += fma(cnt, S.x, u0);                                                                                         \
            acc1 += fma(cnt, S.y, u1);                                                                                         \
        }
#if LR_SCAN_DEPTH2 == 2
        // TWO groups per trip: both decoded, both next loads issued, sixteen gathers in flight, then the arithmetic of both -
        // half as many waits for the LDS per group
        while (has) {
            lr_gload_wait<0>(wA);
            lr_gload_wait<0>(wB);
            const unsigned int aS = wA.x & 0xfff0u, bS = wB.x & 0xfff0u;
            const double acnt = (double)(wA.x & 0xfu);
            const unsigned int a0 = lr_word_off16(wA.x, 1), a1 = lr_word_off16(wA.y, 0), a2 = lr_word_off16(wA.y, 1),
                               a3 = lr_word_off16(wA.z, 0), a4 = lr_word_off16(wA.z, 1), a5 = lr_word_off16(wA.w, 0),
                               a6 = lr_word_off16(wA.w, 1);
            i0 += n_scan;
            const bool hasB = i0 < n;                      // the second group of this trip exists
            // (a B past the end: the re-read of A's line - its count and offsets are masked to entry 0 = 0.0 below)
            const unsigned int mB = hasB ? 0xffffffffu : 0u;
            const double bcnt = (double)(wB.x & 0xfu & mB);
            const unsigned int bSo = bS & mB;
            const unsigned int b0 = lr_word_off16(wB.x, 1) & mB, b1 = lr_word_off16(wB.y, 0) & mB, b2 = lr_word_off16(wB.y, 1) & mB,
                               b3 = lr_word_off16(wB.z, 0) & mB, b4 = lr_word_off16(wB.z, 1) & mB, b5 = lr_word_off16(wB.w, 0) & mB,
                               b6 = lr_word_off16(wB.w, 1) & mB;
            i0 += n_scan, has = i0 < n;
            gnext += (i0 < n) ? stride_b : 0u;
            lr_gload16_async(wA, gnext, off);
            gnext += (i0 + n_scan < n) ? stride_b : 0u;
            lr_gload16_async(wB, gnext, off);
            const double2 S = *reinterpret_cast<const double2*>(lbase + aS), T = *reinterpret_cast<const double2*>(lbase + bSo);
            const double2 E0 = *reinterpret_cast<const double2*>(lbase + a0), E1 = *reinterpret_cast<const double2*>(lbase + a1);
            const double2 E2 = *reinterpret_cast<const double2*>(lbase + a2), E3 = *reinterpret_cast<const double2*>(lbase + a3);
            const double2 E4 = *reinterpret_cast<const double2*>(lbase + a4), E5 = *reinterpret_cast<const double2*>(lbase + a5);
            const double2 E6 = *reinterpret_cast<const double2*>(lbase + a6);
            const double2 F0 = *reinterpret_cast<const double2*>(lbase + b0), F1 = *reinterpret_cast<const double2*>(lbase + b1);
            const double2 F2 = *reinterpret_cast<const double2*>(lbase + b2), F3 = *reinterpret_cast<const double2*>(lbase + b3);
            const double2 F4 = *reinterpret_cast<const double2*>(lbase + b4), F5 = *reinterpret_cast<const double2*>(lbase + b5);
            const double2 F6 = *reinterpret_cast<const double2*>(lbase + b6);
            const double u0 = ((E0.x + E1.x) + (E2.x + E3.x)) + ((E4.x + E5.x) + E6.x);
            const double u1 = ((E0.y + E1.y) + (E2.y + E3.y)) + ((E4.y + E5.y) + E6.y);
            acc0 += fma(acnt, S.x, u0);
            acc1 += fma(acnt, S.y, u1);
            const double v0 = ((F0.x + F1.x) + (F2.x + F3.x)) + ((F4.x + F5.x) + F6.x);
            const double v1 = ((F0.y + F1.y) + (F2.y + F3.y)) + ((F4.y + F5.y) + F6.y);
            acc0 += fma(bcnt, T.x, v0);
            acc1 += fma(bcnt, T.y, v1);
        }
#else
        while (has) {
            LR_SCAN_TRIP(wA)
            if (!has) break;
            LR_SCAN_TRIP(wB)
        }
#endif
#undef LR_SCAN_TRIP
        // both idle loads are still in flight: the caller drains them (lr_scan_drain)
        tail->w = wA, tail->f0 = wB;
        *acc0_ = acc0, *acc1_ = acc1;
        return;
    }
#endif
    // 32-bit loop arithmetic (fewer than 2^27 groups)
    const int n = ZERO_TAIL ? __builtin_amdgcn_readfirstlane((int)n8) : (int)n8;
    // The group address of a trip = a wave-uniform base (scalar registers, advanced by scalar instructions) + the lane's
    // constant byte offset: ZERO_TAIL keeps the whole loop control off the vector ALU (a v_add_u32 with a scalar operand
    // costs 4.2 cycles of the SIMD's issue, as much as an fp64 add).
    const char* gbase = lr_uniform_ptr(idx8);
    const unsigned int stride_b = (unsigned int)n_scan * 16u;
    const unsigned int end_b = (unsigned int)n * 16u;
    unsigned int off = (unsigned int)sid * 16u;
    int i0 = __builtin_amdgcn_readfirstlane((int)sid);
    bool has = ZERO_TAIL ? (i0 < n) : (off < end_b);
    lr_u32x4 w = {0u, 0u, 0u, 0u};
    if (first) w = lr_u32x4{first->w.x, first->w.y, first->w.z, first->w.w};
    else lr_gload16_async(w, gbase, off);
#pragma unroll UNROLL
    while (has) {
        lr_gload_wait<0>(w);
        const unsigned int oS = w.x & 0xfff0u;
        const double cnt = (double)(w.x & 0xfu);
        const unsigned int o0 = lr_word_off16(w.x, 1), o1 = lr_word_off16(w.y, 0), o2 = lr_word_off16(w.y, 1),
                           o3 = lr_word_off16(w.z, 0), o4 = lr_word_off16(w.z, 1), o5 = lr_word_off16(w.w, 0),
                           o6 = lr_word_off16(w.w, 1);
        if (ZERO_TAIL) i0 += n_scan, has = i0 < n, gbase += has ? stride_b : 0u;
        else off += stride_b, has = off < end_b;
        // Unconditionally: a lane's last trip loads a group it will not score (behind a tile or slice the next one's,
        // behind the data zeros: lr_groups_alloc keeps more spare groups than any stride).  Under a condition `w` becomes
        // a merge of two values, which the compiler copies while the load is in flight (scratch/check_async_loads.py
        // checks the compiled code for such reads).  ZERO_TAIL: the wave's last trip re-reads its current group - a
        // line it has just had - so that the wait for this idle load is short.
        lr_gload16_async(w, gbase, off);
        const double2 S = *reinterpret_cast<const double2*>(lbase + oS);
        const double2 E0 = *reinterpret_cast<const double2*>(lbase + o0), E1 = *reinterpret_cast<const double2*>(lbase + o1);
        const double2 E2 = *reinterpret_cast<const double2*>(lbase + o2), E3 = *reinterpret_cast<const double2*>(lbase + o3);
        const double2 E4 = *reinterpret_cast<const double2*>(lbase + o4), E5 = *reinterpret_cast<const double2*>(lbase + o5);
        const double2 E6 = *reinterpret_cast<const double2*>(lbase + o6);
        // fixed pairwise tree over the slots, then the birth entry `count` times
        const double u0 = ((E0.x + E1.x) + (E2.x + E3.x)) + ((E4.x + E5.x) + E6.x);
        const double u1 = ((E0.y + E1.y) + (E2.y + E3.y)) + ((E4.y + E5.y) + E6.y);
        acc0 += fma(cnt, S.x, u0);
        acc1 += fma(cnt, S.y, u1);
    }
    // the idle load of the last trip is still in flight: the caller drains it (lr_scan_drain) once it has done what does
    // not need the registers - or it is waited for here
    if (tail) tail->w = w, tail->f0 = lr_u32x4{0u, 0u, 0u, 0u};
    else lr_gload_wait<0>(w);
    *acc0_ = acc0, *acc1_ = acc1;
}

// The pair-sum planes of a pair table from its E plane.  `tab` = the table's doubles (entry e of chain c at 2 e + c);
// plane 2 + d, entry j = E[j] + E[j + d] for the death entries j, j + d <= n_bins + 1 that lineages can be paired on (the
// extant block of model 3 behind them is gathered through single slots only).
// (a) one chain's column, by the wave that just built its S and E planes; `dup` != 0: also the copy `dup` doubles on
__device__ __forceinline__ void lr_pair_planes_wave(double* tab, int H, int n_bins, int lane, int dup) {
    const double* E = tab + 2 * H;
    for (int j = lane; j <= n_bins + 1; j += LR_WAVE) {
        const double e0 = E[2 * j];
#pragma unroll
        for (int d = 0; d <= LR_PAIR_DMAX; ++d) {
            if (j + d <= n_bins + 1) {
                const double v = e0 + E[2 * (j + d)];
                tab[2 * ((2 + d) * H + j)] = v;
                if (dup) tab[2 * ((2 + d) * H + j) + dup] = v;
            }
        }
    }
}
// (b) both chains at once, by a whole block that just copied S and E from global memory
__device__ __forceinline__ void lr_pair_planes_block(double2* tab, int H, int n_bins, int tid, int n_threads) {
    // one death entry per lane (four independent reads, four writes: one LDS round trip, no index arithmetic)
    const double2* E = tab + H;
    for (int j = tid; j <= n_bins + 1; j += n_threads) {
        double2 v[LR_PAIR_DMAX + 1];
#pragma unroll
        for (int d = 0; d <= LR_PAIR_DMAX; ++d) v[d] = E[min(j + d, n_bins + 1)];
#pragma unroll
        for (int d = 0; d <= LR_PAIR_DMAX; ++d)
            if (j + d <= n_bins + 1) tab[(2 + d) * H + j] = make_double2(v[0].x + v[d].x, v[0].y + v[d].y);
    }
}

// The same scan on GENERAL lineage times.  Pair tables in LDS (LR_TAB_PAIRGEN as the persistent kernels lay it out): six
// planes of H 16-byte entries,  S | E | E2 | slopes of S | slopes of E | slopes of E2,  slopes scaled by 2^-32 and
// E2[j] = 2 E[j] (value and slope alike; derived in LDS by lr_pair_planes_*_general, global memory holds S, E and their
// slopes only).  Groups as at unit resolution - header (birth index << 4 | number of lineages), LR_SLOTS sixteen-bit
// value-entry indices - where a slot holds one lineage (entry H + j) or two consecutive lineages of the run that die in
// the SAME bin j (entry 2 H + j); the slope entry of a slot is its value entry + 3 H.  Beside the groups LR_FRAC_ARRAYS
// arrays of uint4, `fstride` entries apart (every load a fully coalesced 16-byte load): arrays 0 and 1 hold the slots'
// in-bin fractions fe' = ceil te - te as 32-bit fixed point (a pair: the mean of its two, which the doubled slope of the
// E2 plane turns back into their sum), array 2 the SUM of the group's birth fractions fs as one exact double - the
// lineages of a group share the birth bin, so the birth side needs nothing else:
//     sum_i (S.v + fs_i S.s + E_i.v + fe'_i E_i.s)  =  cnt S.v + (sum_i fs_i) S.s + sum_slots fma(fe'_slot, E*_slot.s, E*_slot.v)
// Per group: 2 + 14 ds_read_b128; per slot and chain pair one conversion and four fp64 operations.
#define LR_FRAC_ARRAYS 3
// (plain loads, a test per lane: see lr_persist_scan_pair_slice)
template <int H, int UNROLL = 1, bool PREFETCH = false, bool ONE = false>
__device__ __forceinline__ void lr_persist_scan_pair_general_slice(const char* __restrict__ lbase, const uint4* __restrict__ idx8,
                                                             const uint4* __restrict__ frac, long long fstride,
                                                             long long n8, long long sid, int n_scan, double* acc0_,
                                                             double* acc1_, const lr_first_group* first = nullptr) {
    double acc0 = *acc0_, acc1 = *acc1_;
    const int n = (int)n8;
    int i = (int)sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    uint4 fw[LR_FRAC_ARRAYS];
#pragma unroll
    for (int j = 0; j < LR_FRAC_ARRAYS; ++j) fw[j] = make_uint4(0u, 0u, 0u, 0u);
    if (first && PREFETCH) {
        w = first->w;
#pragma unroll
        for (int j = 0; j < LR_FRAC_ARRAYS; ++j) fw[j] = first->fw[j];
    } else if (i < n) {
        w = idx8[i];
        if (PREFETCH) {
#pragma unroll
            for (int j = 0; j < LR_FRAC_ARRAYS; ++j) fw[j] = frac[i + j * fstride];
        }
    }
    constexpr int SLOPES = 3 * H * 16;        // bytes from a value entry to its slope entry
#pragma unroll UNROLL
    while (i < n) {
        const uint4 cur = w;
        uint4 fr[LR_FRAC_ARRAYS];
#pragma unroll
        for (int j = 0; j < LR_FRAC_ARRAYS; ++j) fr[j] = PREFETCH ? fw[j] : frac[i + j * fstride];
        const int nx = i + n_scan;
        if (nx < n) {
            // the next group in flight while this one is scored - with its fractions where the register budget allows
            w = idx8[nx];
            if (PREFETCH) {
#pragma unroll
                for (int j = 0; j < LR_FRAC_ARRAYS; ++j) fw[j] = frac[nx + j * fstride];
            }
        }
        const char* pS = lbase + (cur.x & 0xfff0u);
        const double cnt = (double)(cur.x & 0xfu);
        const double sfs = __hiloint2double((int)fr[2].y, (int)fr[2].x);
        const unsigned int off[LR_SLOTS] = {lr_word_off16(cur.x, 1), lr_word_off16(cur.y, 0), lr_word_off16(cur.y, 1), lr_word_off16(cur.z, 0),
                                            lr_word_off16(cur.z, 1), lr_word_off16(cur.w, 0), lr_word_off16(cur.w, 1)};
        const unsigned int fq[LR_SLOTS] = {fr[0].x, fr[0].y, fr[0].z, fr[0].w, fr[1].x, fr[1].y, fr[1].z};
        if (ONE) {
            // one chain per table: 8-byte reads of the halves in use, the same operations on them in the same order
            const double Sv1 = *reinterpret_cast<const double*>(pS), Ss1 = *reinterpret_cast<const double*>(pS + SLOPES);
            double q0[LR_SLOTS];
#pragma unroll
            for (int k = 0; k < LR_SLOTS; ++k) {
                const double Ev = *reinterpret_cast<const double*>(lbase + off[k]);
                const double Es = *reinterpret_cast<const double*>(lbase + off[k] + SLOPES);
                q0[k] = fma((double)fq[k], Es, Ev);
            }
            const double u0 = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q0[4] + q0[5]) + q0[6]);
            acc0 += fma(sfs, Ss1, fma(cnt, Sv1, u0));
            i = nx;
            continue;
        }
        const double2 Sv = *reinterpret_cast<const double2*>(pS);
        const double2 Ss = *reinterpret_cast<const double2*>(pS + SLOPES);
        double p0[LR_SLOTS], p1[LR_SLOTS];
#pragma unroll
        for (int k = 0; k < LR_SLOTS; ++k) {
            const double2 Ev = *reinterpret_cast<const double2*>(lbase + off[k]);
            const double2 Es = *reinterpret_cast<const double2*>(lbase + off[k] + SLOPES);
            const double fe = (double)fq[k];
            p0[k] = fma(fe, Es.x, Ev.x);
            p1[k] = fma(fe, Es.y, Ev.y);
        }
        // the same fixed tree over the slots as the unit-resolution scan, then the birth side of the whole group
        const double u0 = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p0[4] + p0[5]) + p0[6]);
        const double u1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + p1[6]);
        acc0 += fma(sfs, Ss.x, fma(cnt, Sv.x, u0));
        acc1 += fma(sfs, Ss.y, fma(cnt, Sv.y, u1));
        i = nx;
    }
    *acc0_ = acc0, *acc1_ = acc1;
}

template <int H, int UNROLL = 1, bool PREFETCH = false, bool ZERO_TAIL = false, bool ASYNC = ZERO_TAIL, bool ONE = false>
__device__ __forceinline__ void lr_persist_scan_pair_general(const char* __restrict__ lbase, const uint4* __restrict__ idx8,
                                                             const uint4* __restrict__ frac, long long fstride,
                                                             long long n8, long long sid, int n_scan, double* acc0_,
                                                             double* acc1_, const lr_first_group* first = nullptr,
                                                             lr_scan_tail* tail = nullptr) {
    static_assert(ASYNC || !ZERO_TAIL, "the wave-uniform trip count comes with the hand-placed loads");
    static_assert(!(ONE && ASYNC), "the one-chain form exists for the plain-load slices only");
    if (!ASYNC) {
        lr_persist_scan_pair_general_slice<H, UNROLL, PREFETCH, ONE>(lbase, idx8, frac, fstride, n8, sid, n_scan, acc0_, acc1_, first);
        if (tail) tail->w = tail->f0 = tail->f1 = tail->f2 = lr_u32x4{0u, 0u, 0u, 0u};
        return;
    }
    double acc0 = *acc0_, acc1 = *acc1_;
    const int n = ZERO_TAIL ? __builtin_amdgcn_readfirstlane((int)n8) : (int)n8;
    const char* gbase = lr_uniform_ptr(idx8);
    const char* fb0 = lr_uniform_ptr(frac);
    const char* fb1 = lr_uniform_ptr(frac + fstride);
    const char* fb2 = lr_uniform_ptr(frac + 2 * fstride);
    static_assert(LR_FRAC_ARRAYS == 3, "three fraction arrays");
    const unsigned int stride_b = (unsigned int)n_scan * 16u;
    const unsigned int end_b = (unsigned int)n * 16u;
    unsigned int off = (unsigned int)sid * 16u;
    int i0 = __builtin_amdgcn_readfirstlane((int)sid);
    bool has = ZERO_TAIL ? (i0 < n) : (off < end_b);
    lr_u32x4 w = {0u, 0u, 0u, 0u}, f0 = {0u, 0u, 0u, 0u}, f1 = {0u, 0u, 0u, 0u}, f2 = {0u, 0u, 0u, 0u};
    if (first && PREFETCH) {
        w = lr_u32x4{first->w.x, first->w.y, first->w.z, first->w.w};
        f0 = lr_u32x4{first->fw[0].x, first->fw[0].y, first->fw[0].z, first->fw[0].w};
        f1 = lr_u32x4{first->fw[1].x, first->fw[1].y, first->fw[1].z, first->fw[1].w};
        f2 = lr_u32x4{first->fw[2].x, first->fw[2].y, first->fw[2].z, first->fw[2].w};
    } else {
        lr_gload16_async(w, gbase, off);
        if (PREFETCH) lr_gload16_async(f0, fb0, off), lr_gload16_async(f1, fb1, off), lr_gload16_async(f2, fb2, off);
    }
    constexpr int SLOPES = 3 * H * 16;        // bytes from a value entry to its slope entry
#pragma unroll UNROLL
    while (has) {
        // decode the group, then refill `w` with the next one (in flight while this one is scored)
        if (PREFETCH) lr_gload_wait<0>(w), lr_gload_wait<0>(f0, f1, f2);
        else lr_gload_wait<0>(w);
        const unsigned int oS = w.x & 0xfff0u;
        const double cnt = (double)(w.x & 0xfu);
        const unsigned int o[LR_SLOTS] = {lr_word_off16(w.x, 1), lr_word_off16(w.y, 0), lr_word_off16(w.y, 1),
                                          lr_word_off16(w.z, 0), lr_word_off16(w.z, 1), lr_word_off16(w.w, 0),
                                          lr_word_off16(w.w, 1)};
        double fe[LR_SLOTS], sfs = 0.0;
        if (PREFETCH) {
            // the fractions came with the group: turn them into doubles before their registers are refilled
            const unsigned int fq[LR_SLOTS] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z};
#pragma unroll
            for (int k = 0; k < LR_SLOTS; ++k) fe[k] = (double)fq[k];
            // (a move the compiler cannot postpone: read as plain registers, the sum would be copied out of f2 whenever
            // convenient - after the refill below has been issued, for one)
            unsigned int s_lo, s_hi;
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(s_lo), "=&v"(s_hi) : "v"(f2.x), "v"(f2.y));
            sfs = __hiloint2double((int)s_hi, (int)s_lo);
        } else {
            lr_gload16_async(f0, fb0, off), lr_gload16_async(f1, fb1, off), lr_gload16_async(f2, fb2, off);
        }
        if (ZERO_TAIL) i0 += n_scan, has = i0 < n, off += has ? stride_b : 0u;
        else off += stride_b, has = off < end_b;
        // unconditionally (see lr_persist_scan_pair): a lane's last trip loads a group it will not score
        lr_gload16_async(w, gbase, off);
        if (PREFETCH) lr_gload16_async(f0, fb0, off), lr_gload16_async(f1, fb1, off), lr_gload16_async(f2, fb2, off);
        const char* pS = lbase + oS;
        const double2 Sv = *reinterpret_cast<const double2*>(pS);
        const double2 Ss = *reinterpret_cast<const double2*>(pS + SLOPES);
        double2 Ev[LR_SLOTS], Es[LR_SLOTS];
#pragma unroll
        for (int k = 0; k < LR_SLOTS; ++k) {
            Ev[k] = *reinterpret_cast<const double2*>(lbase + o[k]);
            Es[k] = *reinterpret_cast<const double2*>(lbase + o[k] + SLOPES);
        }
        if (!PREFETCH) {
            // this group's fractions: everything but the next group's load has landed
            lr_gload_wait<1>(f0, f1, f2);
            const unsigned int fq[LR_SLOTS] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z};
#pragma unroll
            for (int k = 0; k < LR_SLOTS; ++k) fe[k] = (double)fq[k];
            sfs = __hiloint2double((int)f2.y, (int)f2.x);
        }
        double p0[LR_SLOTS], p1[LR_SLOTS];
#pragma unroll
        for (int k = 0; k < LR_SLOTS; ++k) {
            p0[k] = fma(fe[k], Es[k].x, Ev[k].x);
            p1[k] = fma(fe[k], Es[k].y, Ev[k].y);
        }
        // the same fixed tree over the slots as the unit-resolution scan, then the birth side of the whole group
        const double u0 = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p0[4] + p0[5]) + p0[6]);
        const double u1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + p1[6]);
        acc0 += fma(sfs, Ss.x, fma(cnt, Sv.x, u0));
        acc1 += fma(sfs, Ss.y, fma(cnt, Sv.y, u1));
    }
    if (tail) {
        tail->w = w;
        if (PREFETCH) tail->f0 = f0, tail->f1 = f1, tail->f2 = f2;
    } else {
        lr_gload_wait<0>(w);
        if (PREFETCH) lr_gload_wait<0>(f0, f1, f2);
    }
    *acc0_ = acc0, *acc1_ = acc1;
}

// the doubled planes of a pair-general table in LDS: value plane E2 (entries [2H, 3H)) and its slopes ([5H, 6H)) from E
// ([H, 2H)) and its slopes ([4H, 5H)); `tab` in doubles, entry e of chain c at 2 e + c
__device__ __forceinline__ void lr_pair_planes_wave_general(double* tab, int H, int n_bins, int lane, int dup) {
    for (int j = lane; j <= n_bins + 1; j += LR_WAVE) {
        const double v = 2.0 * tab[2 * (H + j)], sl = 2.0 * tab[2 * (4 * H + j)];
        tab[2 * (2 * H + j)] = v, tab[2 * (5 * H + j)] = sl;
        if (dup) tab[2 * (2 * H + j) + dup] = v, tab[2 * (5 * H + j) + dup] = sl;
    }
}
__device__ __forceinline__ void lr_pair_planes_block_general(double2* tab, int H, int n_bins, int tid, int n_threads) {
    for (int j = tid; j <= n_bins + 1; j += n_threads) {
        const double2 v = tab[H + j], sl = tab[4 * H + j];
        tab[2 * H + j] = make_double2(2.0 * v.x, 2.0 * v.y);
        tab[5 * H + j] = make_double2(2.0 * sl.x, 2.0 * sl.y);
    }
}
// The six-plane scan table of a block from the COLUMNS of its two chains (`c0`, `c1`: one chain's [S | E] planes - general
// times: + their slopes - entries 1 double apart; c1 = nullptr: no second chain, its half of every entry is zero): copies
// the planes side by side and derives the pair planes from the columns themselves - no lane depends on another's writes.
template <bool GENERAL>
__device__ __forceinline__ void lr_build_scan_table(double2* scan, const double* c0, const double* c1, int H, int n_bins, int tid,
                                                    int n_threads) {
    auto at = [&](int i) { return make_double2(c0[i], c1 ? c1[i] : 0.0); };
    if (GENERAL) {
        for (int i = tid; i < 4 * H; i += n_threads) scan[i < 2 * H ? i : i + H] = at(i);
        for (int j = tid; j <= n_bins + 1; j += n_threads) {
            const double2 v = at(H + j), sl = at(3 * H + j);
            scan[2 * H + j] = make_double2(2.0 * v.x, 2.0 * v.y);
            scan[5 * H + j] = make_double2(2.0 * sl.x, 2.0 * sl.y);
        }
    } else {
        for (int i = tid; i < 2 * H; i += n_threads) scan[i] = at(i);
        for (int j = tid; j <= n_bins + 1; j += n_threads) {
            double2 v[LR_PAIR_DMAX + 1];
#pragma unroll
            for (int d = 0; d <= LR_PAIR_DMAX; ++d) v[d] = at(H + min(j + d, n_bins + 1));
#pragma unroll
            for (int d = 0; d <= LR_PAIR_DMAX; ++d)
                if (j + d <= n_bins + 1) scan[(2 + d) * H + j] = make_double2(v[0].x + v[d].x, v[0].y + v[d].y);
        }
    }
}

// global memory keeps a pair-general table as [S | E | slopes of S | slopes of E] (4 H entries); its place in the six
// planes of the LDS image
__host__ __device__ __forceinline__ int lr_pairgen_lds_entry(int i, int H) { return i < 2 * H ? i : i + H; }

// what a persistent engine scans: packed groups and, on general times, the fractions behind them
struct lr_packed_lineages {
    const uint4* idx8;
    const uint4* frac;      // nullptr for unit-resolution data
    long long fstride;
};

template <int H, bool GENERAL, int UNROLL = 1, bool PREFETCH = false, bool ZERO_TAIL = false, bool ASYNC = ZERO_TAIL, bool ONE = false>
__device__ __forceinline__ void lr_persist_scan(const char* __restrict__ lbase, const lr_packed_lineages& pk, long long g0,
                                                long long n8, long long sid, int n_scan, double* acc0, double* acc1,
                                                const lr_first_group* first = nullptr, lr_scan_tail* tail = nullptr) {
    if (GENERAL) lr_persist_scan_pair_general<H, UNROLL, PREFETCH, ZERO_TAIL, ASYNC, ONE>(lbase, pk.idx8 + g0, pk.frac + g0, pk.fstride, n8, sid, n_scan, acc0, acc1, first, tail);
    else lr_persist_scan_pair<H, UNROLL, ZERO_TAIL, ASYNC, ONE>(lbase, pk.idx8 + g0, n8, sid, n_scan, acc0, acc1, first, tail);
}

// the first group of lane `sid` (zeros when the lane has none)
template <bool GENERAL>
__device__ __forceinline__ void lr_load_first_group(const lr_packed_lineages& pk, long long g0, long long n8, long long sid,
                                                    lr_first_group* f) {
    f->w = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < 4; ++j) f->fw[j] = make_uint4(0u, 0u, 0u, 0u);
    if (sid < n8) {
        f->w = pk.idx8[g0 + sid];
        if (GENERAL) {
#pragma unroll
            for (int j = 0; j < LR_FRAC_ARRAYS; ++j) f->fw[j] = pk.frac[g0 + sid + j * pk.fstride];
        }
    }
}
