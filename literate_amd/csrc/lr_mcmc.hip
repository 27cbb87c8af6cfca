// lr_mcmc.hip - proposal scorers with explicit draws (A7/A8/A10) and the fused multi-chain
// RJMCMC engine (A11: runMCMC, LiteRateForward.py:216-373).
//
// Engine iteration = [lr_scan_kernel over all lineages x all chains] -> [lr_chain_step_kernel:
// one wave per chain: reduce the tile partials in order, Metropolis-Hastings accept, write the
// trace row, draw the next proposal from the chain's Philox stream, build its lookup tables].
// Iterations are captured in a hipGraph so the host only replays it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>

#include "lr_chain.h"
#include "lr_dd.h"
#include "lr_internal.h"
#include "lr_scan.h"

// ------------------------------------------------------------------------------------------
// explicit-draw scorers (parity with reference-generated vectors)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LR_WAVE) void lr_rj_propose_score_kernel(
    const double* __restrict__ rates, const double* __restrict__ times, const int* __restrict__ K, int kmax,
    const int* __restrict__ move, const int* __restrict__ index, const double* __restrict__ draws, double mult_d,
    double* __restrict__ out_rates, double* __restrict__ out_times, int* __restrict__ out_K,
    double* __restrict__ out_score) {
    const int c = blockIdx.x, lane = threadIdx.x;
    int k = K[c];
    double R = (lane < k) ? rates[(size_t)c * kmax + lane] : 0.0;
    double T = (lane <= k) ? times[(size_t)c * (kmax + 1) + lane] : 0.0;
    const double* dr = draws + (size_t)c * 2 * kmax;
    const int mv = move[c];
    double score = 0.0;
    if (mv == 0) {
        const bool ff = (lane < k) ? (dr[lane] != 0.0) : false;
        const double u = (lane < k) ? dr[kmax + lane] : 0.5;
        score = lr_wave_multiplier(R, k, ff, u, 2.0 * log(mult_d), lane);
    } else if (mv == 1) {
        score = lr_wave_add_shift(R, T, k, index[c], dr[0], dr[1], lane);
    } else if (mv == 2) {
        score = lr_wave_remove_shift(R, T, k, index[c], lane);
    }
    if (lane < kmax) out_rates[(size_t)c * kmax + lane] = (lane < k) ? R : 0.0;
    if (lane <= kmax) out_times[(size_t)c * (kmax + 1) + lane] = (lane <= k) ? T : 0.0;
    if (lane == 0) out_K[c] = k, out_score[c] = score;
}

extern "C" int lr_rj_propose_score(const double* rates, const double* times, const int32_t* K, int32_t kmax,
                                   int32_t n_chains, const int32_t* move, const int32_t* index, const double* draws,
                                   double mult_d, double* out_rates, double* out_times, int32_t* out_K,
                                   double* out_score, void* stream_) {
    if (!rates || !times || !K || !move || !index || !draws || !out_rates || !out_times || !out_K || !out_score)
        return LR_ERR_NULL;
    // an add needs room for K+1 rates: kmax must leave it (caller pads), and a wave holds 64 lanes
    if (kmax < 2 || kmax > LR_WAVE - 1 || n_chains < 1) return LR_ERR_SIZE;
    hipLaunchKernelGGL(lr_rj_propose_score_kernel, dim3(n_chains), dim3(LR_WAVE), 0, (hipStream_t)stream_, rates, times,
                       K, kmax, move, index, draws, mult_d, out_rates, out_times, out_K, out_score);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(LR_WAVE) void lr_log_priors_kernel(const double* __restrict__ rates,
                                                                const int* __restrict__ K, int kmax, double shape,
                                                                const double* __restrict__ gamma_rate,
                                                                const double* __restrict__ poi_rate,
                                                                double* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int k = K[c];
    const double R = (lane < k) ? rates[(size_t)c * kmax + lane] : 1.0;
    double p = lr_wave_prior_gamma(R, k, shape, gamma_rate[c], lane);
    if (poi_rate) p += lr_wave_poisson_prior(k, poi_rate[c], lane);
    if (lane == 0) out[c] = p;
}

extern "C" int lr_log_priors(const double* rates, const int32_t* K, int32_t kmax, int32_t n_chains, double shape,
                             const double* gamma_rate, const double* poi_rate, double* out, void* stream_) {
    if (!rates || !K || !gamma_rate || !out) return LR_ERR_NULL;
    if (kmax < 1 || kmax > LR_WAVE - 1 || n_chains < 1) return LR_ERR_SIZE;
    hipLaunchKernelGGL(lr_log_priors_kernel, dim3(n_chains), dim3(LR_WAVE), 0, (hipStream_t)stream_, rates, K, kmax,
                       shape, gamma_rate, poi_rate, out);
    return (int)hipGetLastError();
}

// debug/parity hook for the RNG: out[i] = draw kind[i] at (it[i], purpose[i], idx[i]) of chain stream
//   kind 0: u_a   1: u_b   2: normal   3: gamma(shape[i])
__global__ void lr_debug_draws_kernel(uint32_t seed, uint32_t chain, const long long* it, const int* purpose,
                                      const int* idx, const int* kind, const double* shape, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lr_stream s{seed, chain};
    const uint64_t t = (uint64_t)it[i];
    double v;
    switch (kind[i]) {
        case 0: v = lr_pair(s, t, purpose[i], idx[i]).a; break;
        case 1: v = lr_pair(s, t, purpose[i], idx[i]).b; break;
        case 2: v = lr_normal(s, t, purpose[i], idx[i]); break;
        default: v = lr_gamma(s, t, purpose[i], idx[i], shape[i]); break;
    }
    out[i] = v;
}

extern "C" int lr_debug_draws(uint64_t seed, int64_t chain, const int64_t* it, const int32_t* purpose,
                              const int32_t* idx, const int32_t* kind, const double* shape, int32_t n, double* out,
                              void* stream_) {
    if (!it || !purpose || !idx || !kind || !shape || !out) return LR_ERR_NULL;
    hipLaunchKernelGGL(lr_debug_draws_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream_, (uint32_t)seed,
                       (uint32_t)chain, (const long long*)it, purpose, idx, kind, shape, n, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// the engine
// ------------------------------------------------------------------------------------------
#define LR_MAX_PARTS 4
// a partition = a contiguous, independent block of chains with its own stream and captured graph; inside it
// the chains are software-pipelined in two halves (A = [base, base+hA), B = the rest)
struct lr_part {
    int base, count, hA;
    bool pipelined;
    hipStream_t stream;
    hipEvent_t done;
    hipGraphExec_t graph_exec;
    int graph_units;
};

// Four-chain kernel: delta[s] = trips scanner slot s (wave s + 2) scores beyond (+) or short of (-) the equal share
// k_tot; the deltas sum to zero.  The trips given up are stored, in slot / trip order, behind the takers' own shares.
struct lr_p4_shares {
    int delta[16];
    int n_slots;          // scanner waves striding over the groups: 14 (four-chain kernel) or 8 (two-chain kernel)
};

struct lr_engine {
    lr_mcmc_config cfg;
    lr_mcmc_layout lay;
    lr_scan_plan plan;
    const double* ts;
    const double* te;
    const double* br_length;
    char* ws;
    bool initialised;
    int n_parts;
    lr_part part[LR_MAX_PARTS];
    bool persistent;          // use lr_persist_kernel in lr_mcmc_steps
    long long n8;             // 16-byte groups of packed lineage indices
    long long n8_alloc;       // ... allocated (zero-filled behind the data)
    lr_p4_shares p4;          // per scanner wave: trips more (+) or fewer (-) than the equal share (four-chain kernel)
    hipEvent_t fork;
};

struct lr_step_args {
    lr_mcmc_config cfg;
    double* state_f64;
    int* state_i32;
    const double* log_br;   // [n_bins] log(br_length) (models 0/1)
    double2* tables;
    const double* partials;
    double* trace;
    const double* br_length;
    double log_T;           // log(end_time - start_time)
    double mult_l;          // 2 log d of the multiplier proposal (LRF:169)
    const double* dd_consts; // DD sampler: {max(DT), log max(DT)} = PRIOR_K0_L (DD:45) and its log
    int tab_stride, n_cls, tiles, H, unit, cb;
};

// where chain c's lookup table starts.  General layout: chain-major, tab_stride double2 per chain.
// Unit-resolution layout: groups of cb chains, inside a group [pair][2H] double2 = (even chain, odd chain);
// the returned pointer addresses this chain's component, consecutive entries are 2 doubles apart.
__device__ __forceinline__ double2* lr_chain_table(const lr_step_args& a, int c) {
    if (!a.unit) return a.tables + (size_t)c * a.tab_stride;
    const int l = c % a.cb;
    double* base = reinterpret_cast<double*>(a.tables + (size_t)(c - l) * a.tab_stride);
    return reinterpret_cast<double2*>(base + (size_t)(l >> 1) * (4 * a.H) + (l & 1));
}

// per-wave LDS scratch: segment rates, their logs and integer edges of both processes
struct lr_seg_scratch {
    double rate[2][LR_KMAX];
    double lograte[2][LR_KMAX];
    int edge[2][LR_KMAX + 1];
    int marks[4 * LR_WAVE + 2];   // per unit bin: number of birth-rate shifts (low half) / death-rate shifts (high half)
};

// Lookup tables of one chain straight from its segments (get_rate_index + L[indL] + the table
// builder of lr_device.h in one go, no per-bin transcendental): bin b of process p takes segment
// j with edge[p][j] <= b < edge[p][j+1].  Model conventions as lr_bin_terms, with
// log(k*lam) taken as log k + log lam (log_br = log k is a data constant).
__device__ inline double lr_build_tables_segments_wave(const lr_seg_scratch* sc, int KL, int KM,
                                                       const double* __restrict__ br_length,
                                                       const double* __restrict__ log_br, int model, int n_bins,
                                                       int n_cls, int H, double2* __restrict__ tab, int lane,
                                                       bool unit = false, double fs0 = 0.0, double fe0 = 0.0,
                                                       int es = 2) {
    // unit-resolution layout: tab points at this chain's component of its pair table, entries 2 doubles apart
    double* tabd = reinterpret_cast<double*>(tab);
    const int P = (n_bins + LR_WAVE - 1) / LR_WAVE;
    const int b0 = min(lane * P, n_bins), b1 = min(b0 + P, n_bins);
    int sl0 = 0, sm0 = 0;
    while (sl0 + 1 < KL && sc->edge[0][sl0 + 1] <= b0) ++sl0;
    while (sm0 + 1 < KM && sc->edge[1][sm0 + 1] <= b0) ++sm0;
    double sumR = 0.0, sumRl = 0.0, csum = 0.0;
    int sl = sl0, sm = sm0;
    for (int b = b0; b < b1; ++b) {
        while (sl + 1 < KL && sc->edge[0][sl + 1] <= b) ++sl;
        while (sm + 1 < KM && sc->edge[1][sm + 1] <= b) ++sm;
        const double lam = sc->rate[0][sl], mu = sc->rate[1][sm];
        const bool live = (model >= 2) || (br_length[b] > 0.0);
        double R = 0.0;
        if (live) R = (model == 1) ? mu : lam + mu;
        sumR += R;
        if (model >= 2) sumRl += lam;
        if (model == 1 && live) csum -= lam;
    }
    double totR, totRl = 0.0;
    double cum = lr_wave_exclusive_scan(sumR, lane, &totR);
    double cuml = 0.0;
    if (n_cls == 2) cuml = lr_wave_exclusive_scan(sumRl, lane, &totRl);
    sl = sl0, sm = sm0;
    for (int b = b0; b < b1; ++b) {
        while (sl + 1 < KL && sc->edge[0][sl + 1] <= b) ++sl;
        while (sm + 1 < KM && sc->edge[1][sm + 1] <= b) ++sm;
        const double lam = sc->rate[0][sl], mu = sc->rate[1][sm];
        const double llam = sc->lograte[0][sl], lmu = sc->lograte[1][sm];
        double logB = 0.0, logD = 0.0, R = 0.0;
        if (model >= 2) {
            logB = llam, logD = lmu, R = lam + mu;
        } else if (br_length[b] > 0.0) {
            const double lk = log_br[b];
            logB = (model == 0) ? lk + llam : llam;
            logD = lmu + lk;
            R = (model == 0) ? lam + mu : mu;
        }
        if (unit) {
            tabd[es * (b + 1)] = (logB + cum) + fs0 * R;
            tabd[es * (H + b + 1)] = (logD - cum) - fe0 * R;
        } else {
            tab[b + 1] = make_double2(logB + cum, R);
            tab[H + b + 1] = make_double2(logD - cum, -R);
        }
        cum += R;
        if (n_cls == 2) {
            tab[2 * H + b + 1] = make_double2(logB + cuml, lam);
            tab[3 * H + b + 1] = make_double2(-cuml, -lam);
            cuml += lam;
        }
    }
    if (lane == 0) {
        if (unit) {
            tabd[0] = 0.0, tabd[es * H] = 0.0;
            tabd[es * (n_bins + 1)] = totR, tabd[es * (H + n_bins + 1)] = -totR;
        } else {
            tab[0] = make_double2(0.0, 0.0);
            tab[H] = make_double2(0.0, 0.0);
            tab[n_bins + 1] = make_double2(totR, 0.0);
            tab[H + n_bins + 1] = make_double2(-totR, 0.0);
        }
        if (n_cls == 2) {
            tab[2 * H] = make_double2(0.0, 0.0);
            tab[3 * H] = make_double2(0.0, 0.0);
            tab[2 * H + n_bins + 1] = make_double2(totRl, 0.0);
            tab[3 * H + n_bins + 1] = make_double2(-totRl, 0.0);
        }
    }
    return lr_wave_sum(csum);
}

// Same tables for the common shape (one table class, at most 4 bins per lane, i.e. n_bins <= 256) in ONE pass
// with everything in registers: the segment of a bin is the number of interior edges <= bin, counted by
// broadcasting the K-1 edges with v_readlane (edges live in lanes: lane j holds edge j); rates and their logs
// come from the LDS scratch with independent reads; one DPP scan gives the cumulative exposure.
template <int P>
__device__ __forceinline__ double lr_build_tables_segments_fast(const lr_seg_scratch* sc, int eL, int eM, int KL,
                                                                int KM, const double* __restrict__ br_length,
                                                                const double* __restrict__ log_br, int model,
                                                                int n_bins, int H, double2* __restrict__ tab,
                                                                int lane, bool unit, double fs0, double fe0, int es) {
    double* tabd = reinterpret_cast<double*>(tab);
    const int b0 = lane * P;
    int segL[P], segM[P];
    double k_b[P], lk_b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        segL[p] = 0, segM[p] = 0;
        const int b = min(b0 + p, n_bins - 1);
        k_b[p] = (model < 2) ? br_length[b] : 1.0;
        lk_b[p] = (model < 2) ? log_br[b] : 0.0;
    }
    // Segment of every bin = number of shifts at or before it.  The K - 1 shift lanes drop a count on their bin in
    // LDS (birth shifts in the low half-word, death shifts in the high one), every lane reads the counts of its own
    // bins and one integer wave scan turns them into ranks: a fixed ~35 instructions instead of a dependent
    // (K_l + K_m) x P compare-and-add chain (the largest single item of the chain step before).
    {
        int* marks = const_cast<int*>(sc->marks);
        for (int b = lane; b <= n_bins; b += LR_WAVE) marks[b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane >= 1 && lane < KL) atomicAdd(&marks[eL], 1);
        if (lane >= 1 && lane < KM) atomicAdd(&marks[eM], 0x10000);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int cnt[P], tot = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            cnt[p] = (b0 + p <= n_bins) ? marks[b0 + p] : 0;
            tot += cnt[p];
        }
        int run = lr_wave_exclusive_scan_i32(tot);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            run += cnt[p];
            segL[p] = run & 0xffff, segM[p] = run >> 16;
        }
    }
    double logB[P], logD[P], R[P];
    double sumR = 0.0, csum = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const double lam = sc->rate[0][segL[p]], mu = sc->rate[1][segM[p]];
        const double llam = sc->lograte[0][segL[p]], lmu = sc->lograte[1][segM[p]];
        logB[p] = 0.0, logD[p] = 0.0, R[p] = 0.0;
        if (b0 + p < n_bins) {
            if (model >= 2) {
                logB[p] = llam, logD[p] = lmu, R[p] = lam + mu;
            } else if (k_b[p] > 0.0) {
                logB[p] = (model == 0) ? lk_b[p] + llam : llam;
                logD[p] = lmu + lk_b[p];
                R[p] = (model == 0) ? lam + mu : mu;
                if (model == 1) csum -= lam;
            }
        }
        sumR += R[p];
    }
    double totR;
    double cum = lr_wave_exclusive_scan(sumR, lane, &totR);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int b = b0 + p;
        if (b < n_bins) {
            if (unit) {
                tabd[es * (b + 1)] = (logB[p] + cum) + fs0 * R[p];
                tabd[es * (H + b + 1)] = (logD[p] - cum) - fe0 * R[p];
            } else {
                tab[b + 1] = make_double2(logB[p] + cum, R[p]);
                tab[H + b + 1] = make_double2(logD[p] - cum, -R[p]);
            }
        }
        cum += R[p];
    }
    if (lane == 0) {
        if (unit) {
            tabd[0] = 0.0, tabd[es * H] = 0.0;
            tabd[es * (n_bins + 1)] = totR, tabd[es * (H + n_bins + 1)] = -totR;
        } else {
            tab[0] = make_double2(0.0, 0.0);
            tab[H] = make_double2(0.0, 0.0);
            tab[n_bins + 1] = make_double2(totR, 0.0);
            tab[H + n_bins + 1] = make_double2(-totR, 0.0);
        }
    }
    return (model == 1) ? lr_wave_sum(csum) : 0.0;
}

// dispatcher: fast one-pass builder when the shape allows, general two-pass builder otherwise
__device__ __forceinline__ double lr_build_tables_segments(const lr_seg_scratch* sc, int eL, int eM, int KL, int KM,
                                                           const double* __restrict__ br_length,
                                                           const double* __restrict__ log_br, int model, int n_bins,
                                                           int n_cls, int H, double2* __restrict__ tab, int lane,
                                                           bool unit, double fs0, double fe0, int es = 2) {
    if (n_cls == 1 && n_bins <= 2 * LR_WAVE)
        return lr_build_tables_segments_fast<2>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab, lane, unit,
                                                fs0, fe0, es);
    if (n_cls == 1 && n_bins <= 4 * LR_WAVE)
        return lr_build_tables_segments_fast<4>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab, lane, unit,
                                                fs0, fe0, es);
    return lr_build_tables_segments_wave(sc, KL, KM, br_length, log_br, model, n_bins, n_cls, H, tab, lane, unit, fs0,
                                         fe0, es);
}

// stage one chain's segments in the wave's LDS scratch; log of all rates in ONE call
// (lanes 0..31 carry the birth rates, lanes 32..63 the death rates)
__device__ __forceinline__ void lr_stage_segments(lr_seg_scratch* sc, double L, double M, int eL, int eM, int KL,
                                                  int KM, int lane, double* logL, double* logM, double extra = 1.0,
                                                  double* log_extra = nullptr) {
    const double Mhi = __shfl(M, lane & 31, LR_WAVE);
    const bool hi = lane >= 32;
    const int j = lane & 31;
    const bool valid = hi ? (j < KM) : (j < KL);
    // lane 63 is free unless the death process holds LR_KMAX rates: it takes one more logarithm along (`extra`)
    const bool free63 = KM < LR_KMAX;
    double x = valid ? (hi ? Mhi : L) : 1.0;
    if (lane == LR_WAVE - 1 && free63) x = extra;
    const double lx = log(x);
    if (log_extra) *log_extra = free63 ? lr_bcast(lx, LR_WAVE - 1) : log(extra);
    sc->rate[hi][j] = x;
    sc->lograte[hi][j] = lx;
    if (lane <= LR_KMAX) sc->edge[0][lane] = eL, sc->edge[1][lane] = eM;
    *logL = lx;                                        // valid on lanes < 32
    *logM = __shfl(lx, 32 + (lane & 31), LR_WAVE);     // lane j gets log M[j]
    // the scratch is private to this wave; LDS operations of one wave execute in order, the fence keeps
    // the compiler from moving the reads of lr_build_tables_segments_wave above these writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifdef LR_DIAG
static __device__ unsigned long long lr_diag_step[4096 * 12];
#define LR_SSTAMP(k) if (lane == 0 && c < 4096) lr_diag_step[c * 12 + (k)] = wall_clock64()
#else
#define LR_SSTAMP(k)
#endif

// mode: 0 = regular step (accept pending proposal, then propose), 1 = finish init (adopt the
// evaluated initial state as accepted, then propose iteration 0)
// chain state as it lives in the registers of the chain's wave (lane j holds element j of every row)
struct lr_chain_regs {
    double L, M, tL, tM;        // accepted rates / shift times
    double pL, pM, ptL, ptM;    // pending proposal
    double sc;                  // LR_ROW_SCALARS (lane s holds scalar s)
    int eL, eM, peL, peM;       // integer bin edges, accepted / proposed
    int isc;                    // LR_IROW_SCALARS
};

__device__ __forceinline__ void lr_chain_load(lr_chain_regs& r, const double* S, const int* I, int lane) {
    r.L = S[LR_ROW_L * LR_ROW + lane], r.M = S[LR_ROW_M * LR_ROW + lane];
    r.tL = S[LR_ROW_TL * LR_ROW + lane], r.tM = S[LR_ROW_TM * LR_ROW + lane];
    r.pL = S[LR_ROW_PL * LR_ROW + lane], r.pM = S[LR_ROW_PM * LR_ROW + lane];
    r.ptL = S[LR_ROW_PTL * LR_ROW + lane], r.ptM = S[LR_ROW_PTM * LR_ROW + lane];
    r.sc = S[LR_ROW_SCALARS * LR_ROW + lane];
    r.eL = I[LR_IROW_EL * LR_ROW + lane], r.eM = I[LR_IROW_EM * LR_ROW + lane];
    r.peL = I[LR_IROW_PEL * LR_ROW + lane], r.peM = I[LR_IROW_PEM * LR_ROW + lane];
    r.isc = I[LR_IROW_SCALARS * LR_ROW + lane];
}

__device__ __forceinline__ void lr_chain_store(const lr_chain_regs& r, double* S, int* I, int lane) {
    S[LR_ROW_L * LR_ROW + lane] = r.L, S[LR_ROW_M * LR_ROW + lane] = r.M;
    S[LR_ROW_TL * LR_ROW + lane] = r.tL, S[LR_ROW_TM * LR_ROW + lane] = r.tM;
    S[LR_ROW_PL * LR_ROW + lane] = r.pL, S[LR_ROW_PM * LR_ROW + lane] = r.pM;
    S[LR_ROW_PTL * LR_ROW + lane] = r.ptL, S[LR_ROW_PTM * LR_ROW + lane] = r.ptM;
    S[LR_ROW_SCALARS * LR_ROW + lane] = r.sc;
    I[LR_IROW_EL * LR_ROW + lane] = r.eL, I[LR_IROW_EM * LR_ROW + lane] = r.eM;
    I[LR_IROW_PEL * LR_ROW + lane] = r.peL, I[LR_IROW_PEM * LR_ROW + lane] = r.peM;
    I[LR_IROW_SCALARS * LR_ROW + lane] = r.isc;
}

// One chain step on register-resident state: accept the pending proposal given its log-likelihood sum
// (lik_sum, without the model constant), write the trace row, draw the next proposal and build its lookup
// tables at `table` (global memory or LDS; see lr_chain_table for the addressing).
// mode: 0 = regular step, 1 = finish init (adopt the evaluated initial state, then propose iteration 0)
__device__ __forceinline__ void lr_chain_step_core(lr_chain_regs& st, const lr_step_args& a, int mode, int c, int lane,
                                                   lr_seg_scratch* scratch_p, double lik_sum, double2* table,
                                                   int table_es = 2) {
    lr_seg_scratch& scratch = *scratch_p;
    const lr_mcmc_config& cfg = a.cfg;
    const int C = cfg.n_chains, n_bins = cfg.n_bins;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    double L = st.L, M = st.M, tL = st.tL, tM = st.tM;
    const double pL0 = st.pL, pM0 = st.pM, ptL0 = st.ptL, ptM0 = st.ptM;
    const double sc = st.sc;
    int eL = st.eL, eM = st.eM;
    const int peL0 = st.peL, peM0 = st.peM;
    const int isc = st.isc;
    double lik_p = lr_bcast(sc, LR_S_LIK_P);

    double likA = lr_bcast(sc, LR_S_LIKA), priorA = lr_bcast(sc, LR_S_PRIORA);
    double priorPoiA = lr_bcast(sc, LR_S_PRIORPOIA);
    double g0 = lr_bcast(sc, LR_S_GRATE_L), g1 = lr_bcast(sc, LR_S_GRATE_M), poi = lr_bcast(sc, LR_S_POI);
    double lg0 = lr_bcast(sc, LR_S_LOG_G0), lg1 = lr_bcast(sc, LR_S_LOG_G1), lpoi = lr_bcast(sc, LR_S_LOG_POI);
    double constA = lr_bcast(sc, LR_S_CONST_A);
    int KL = lr_bcast_i(isc, LR_I_KL), KM = lr_bcast_i(isc, LR_I_KM);
    uint64_t it = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_HI) << 32);
    int n_acc = lr_bcast_i(isc, LR_I_ACCEPTED);
    uint64_t next_sample = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_HI) << 32);
    int trace_slot = lr_bcast_i(isc, LR_I_SLOT);
    LR_SSTAMP(1);

    // The wave-uniform draws of the iteration about to be proposed in ONE Philox call (a block costs 40 quarter-rate
    // 32-bit multiplies whether one lane needs it or all 64): lane 0 its acceptance uniform, lane 1 the move selector,
    // lanes 2..3 the two RJ pairs.  Same (iteration, purpose, index) addresses as separate calls would use, so the
    // stream is unchanged.  The logarithm of the acceptance uniform rides along in the packed log of the rates below
    // and waits in LR_S_LOG_U until the proposal is decided, one step later.
    lr_u2 ud;
    {
        const uint64_t it_prop = (mode == 1) ? it : it + 1;
        const uint32_t purpose = (lane == 0) ? LR_P_ACCEPT : (lane == 1 ? LR_P_MOVE : LR_P_RJ);
        ud = lr_pair(rng, it_prop, purpose, lane == 3 ? 1u : 0u);
    }
    const double u_next = lr_bcast(ud.a, 0);

    if (mode == 1) {
        // LRF:224-230.  The initial prior uses prior_gamma's default rate b=2 (LRF:201, 227).
        likA = lik_sum + constA;
        priorA = lr_wave_prior_gamma(L, KL, LR_GAMMA_SHAPE, 2.0, lane) + lr_wave_prior_gamma(M, KM, LR_GAMMA_SHAPE, 2.0, lane);
        priorA += -a.log_T * (KL - 1 + KM - 1);
        priorPoiA = lr_poisson_prior(KL, poi, lpoi) + lr_poisson_prior(KM, poi, lpoi);
        priorA += priorPoiA;
    } else {
        // ---- Metropolis-Hastings accept of iteration `it` (LRF:305-319) ----
        const int gibbs = lr_bcast_i(isc, LR_I_GIBBS), invalid = lr_bcast_i(isc, LR_I_INVALID);
        const double hasting = lr_bcast(sc, LR_S_HASTING), priorP = lr_bcast(sc, LR_S_PRIOR_P);
        const double priorPoiP = lr_bcast(sc, LR_S_PRIORPOI_P), constP = lr_bcast(sc, LR_S_CONST_P);
        const double lik = gibbs ? likA : lik_sum + constP;
        const bool ok = gibbs || (!invalid && (lik - likA + priorP - priorA + hasting >= lr_bcast(sc, LR_S_LOG_U)));
        lik_p = invalid ? -INFINITY : lik;
        if (ok) {
            L = pL0, M = pM0, tL = ptL0, tM = ptM0, eL = peL0, eM = peM0;
            KL = lr_bcast_i(isc, LR_I_PKL), KM = lr_bcast_i(isc, LR_I_PKM);
            likA = lik, priorA = priorP, priorPoiA = priorPoiP, constA = constP;
            n_acc += 1;
        }
        // ---- trace row (LRF:321-359) ----
        // `it % s_freq == 0` kept as a running (next sample, slot) pair: no 64-bit division on the device
        if (it == next_sample) {
            const int slot = trace_slot;
            trace_slot += 1;
            next_sample += (uint64_t)cfg.s_freq;
            if (slot < cfg.n_trace_slots) {
                double* row = a.trace + ((size_t)slot * C + c) * LR_TRACE_W;
                const double meanL = lr_wave_sum(lane < KL ? L : 0.0) / KL;
                const double meanM = lr_wave_sum(lane < KM ? M : 0.0) / KM;
                double h = 0.0;
                switch (lane) {
                    case 0: h = (double)it; break;
                    case 1: h = likA + priorA; break;
                    case 2: h = likA; break;
                    case 3: h = priorA; break;
                    case 4: h = meanL; break;
                    case 5: h = meanM; break;
                    case 6: h = KL; break;
                    case 7: h = KM; break;
                    case 8: h = cfg.start_time; break;
                    case 9: h = cfg.end_time; break;
                    case 10: h = g0; break;
                    case 11: h = g1; break;
                    case 12: h = poi; break;
                }
                if (lane < LR_TRACE_HEAD) row[lane] = h;
                const double nan = __longlong_as_double(0x7ff8000000000000LL);
                double* rl = row + LR_TRACE_HEAD;
                double* rm = rl + (2 * LR_KMAX - 1);
                if (lane < LR_KMAX) rl[lane] = lane < KL ? L : nan, rm[lane] = lane < KM ? M : nan;
                if (lane >= 1 && lane < LR_KMAX) {
                    rl[LR_KMAX + lane - 1] = lane < KL ? tL : nan;
                    rm[LR_KMAX + lane - 1] = lane < KM ? tM : nan;
                }
            }
        }
        it += 1;
    }

    LR_SSTAMP(2);
    // ---- propose iteration `it` (LRF:234-287) ----
    double pL = L, pM = M, ptL = tL, ptM = tM;
    int peL = eL, peM = eM, PKL = KL, PKM = KM;
    double hasting = 0.0, priorPoi = 0.0;
    int gibbs = 0, invalid = 0, move_kind;
    const double sample_shift_mu = cfg.const_death_rate ? 0.0 : 0.5;
    const double b_freq = cfg.const_death_rate ? 0.7 : 0.4, d_freq = 0.8;
    const double fL = cfg.update_fraction, fM = cfg.const_death_rate ? 1.0 : cfg.update_fraction;
    const lr_u2 r{lr_bcast(ud.a, 1), lr_bcast(ud.b, 1)};
    if (r.a < b_freq) {
        if (r.b < .5 || KL == 1) {
            const lr_u2 d = lr_pair(rng, it, LR_P_MULT, lane);
            hasting = lr_wave_multiplier(pL, KL, d.a < fL, d.b, a.mult_l, lane);
            move_kind = 0;
        } else {
            peL = lr_wave_edges(tL, 0);  // update_times leaves the times unchanged (LRF:178-195)
            move_kind = 1;
        }
    } else if (r.a < d_freq) {
        if (r.b < .5 || KM == 1) {
            const lr_u2 d = lr_pair(rng, it, LR_P_MULT, lane);
            hasting = lr_wave_multiplier(pM, KM, d.a < fM, d.b, a.mult_l, lane);
            move_kind = 2;
        } else {
            peM = lr_wave_edges(tM, 0);
            move_kind = 3;
        }
    } else if (r.a < 0.999 && cfg.const_rates == 0) {
        // RJMCMC (LRF:71-97)
        move_kind = 4;
        const lr_u2 q{lr_bcast(ud.a, 2), lr_bcast(ud.b, 2)};
        const bool sideL = q.a > sample_shift_mu;
        double R = sideL ? L : M, T = sideL ? tL : tM;
        int K = sideL ? KL : KM;
        double score = 0.0;
        const lr_u2 q2{lr_bcast(ud.a, 3), lr_bcast(ud.b, 3)};
        if (q.b > 0.5) {
            if (K >= LR_KMAX) {
                invalid = 1;  // device cap on the number of rates; the reference has none
            } else {
                const int ind = min((int)(q2.a * K), K - 1);
                const double delta = q2.b * (lr_bcast(T, ind + 1) - lr_bcast(T, ind));
                double ga, gb;
                lr_wave_gamma2(rng, it, LR_P_BETA_A, LR_SHAPE_BETA_RJ, LR_P_BETA_B, LR_SHAPE_BETA_RJ, lane, &ga, &gb);
                score = lr_wave_add_shift(R, T, K, ind, delta, ga / (ga + gb), lane);
            }
        } else if (K > 1) {
            const int idx = 1 + min((int)(q2.a * (K - 1)), K - 2);
            score = lr_wave_remove_shift(R, T, K, idx, lane);
        }
        hasting = score;
        const int E = lr_wave_edges(T, 0);
        if (sideL) pL = R, ptL = T, PKL = K, peL = E;
        else pM = R, ptM = T, PKM = K, peM = E;
        priorPoi = lr_poisson_prior(PKL, poi, lpoi) + lr_poisson_prior(PKM, poi, lpoi);
    } else {
        // Gibbs draws of the hyper-parameters (LRF:283-287, 99-108, 210-213)
        move_kind = 5;
        double gl = 0.0, gm = 0.0, gp = 0.0, dummy;
        if (cfg.use_rate_HP)
            lr_wave_gamma2(rng, it, LR_P_GIBBS_L, LR_HP_GAMMA_SHAPE + LR_GAMMA_SHAPE * KL, LR_P_GIBBS_M,
                           LR_HP_GAMMA_SHAPE + LR_GAMMA_SHAPE * KM, lane, &gl, &gm);
        if (cfg.poisson_HP == 0.0)
            lr_wave_gamma2(rng, it, LR_P_GIBBS_POI, LR_RJHP_SHAPE + KL + KM, LR_P_GIBBS_POI, LR_RJHP_SHAPE + KL + KM, lane,
                           &gp, &dummy);
        if (cfg.poisson_HP == 0.0) poi = gp * (1. / (LR_RJHP_RATE + 2));
        if (cfg.use_rate_HP) {
            const double sL = lr_wave_sum(lane < KL ? L : 0.0), sM = lr_wave_sum(lane < KM ? M : 0.0);
            g0 = gl * (1. / (LR_HP_GAMMA_RATE + sL));
            g1 = gm * (1. / (LR_HP_GAMMA_RATE + sM));
        }
        // one packed log for the three cached logarithms
        const double lx = log(lane == 0 ? g0 : (lane == 1 ? g1 : (lane == 2 ? poi : 1.0)));
        lg0 = lr_bcast(lx, 0), lg1 = lr_bcast(lx, 1), lpoi = lr_bcast(lx, 2);
        gibbs = 1;
    }

    LR_SSTAMP(3);
    // segments of the proposal -> LDS scratch (+ log of every rate in one call)
    double logpL, logpM, log_u_next;
    lr_stage_segments(&scratch, pL, pM, peL, peM, PKL, PKM, lane, &logpL, &logpM, u_next, &log_u_next);

    LR_SSTAMP(4);
    // guard against tiny time frames (LRF:290-292) and the prior of the proposal (LRF:296-304)
    double priorP = -INFINITY;
    {
        // one reduction for both processes: min over all segment lengths
        const double nL = lr_dpp_zero<0x130 /* wave_shl:1 */, 0xf, 0xf>(ptL);   // lane l <- element l+1
        const double nM = lr_dpp_zero<0x130, 0xf, 0xf>(ptM);
        const double dmin = fmin(lane < PKL ? fabs(nL - ptL) : 1e300, lane < PKM ? fabs(nM - ptM) : 1e300);
        if (__ballot(dmin <= LR_MIN_ALLOWED_T)) invalid = 1;       // min <= 1  <=>  any <= 1: one compare, no reduction
    }
    if (!invalid) {
        // Gamma(2, g) log-densities of all rates of both processes in one reduction (LRF:296)
        const double vL = (lane < PKL) ? (logpL + lg0) - pL * g0 + lg0 : 0.0;
        const double vM = (lane < PKM) ? (logpM + lg1) - pM * g1 + lg1 : 0.0;
        priorP = lr_wave_sum(vL + vM);
        priorP += -a.log_T * (PKL - 1 + PKM - 1);
        if (priorPoi != 0.0) priorP += priorPoi;
        else priorP += priorPoiA, priorPoi = priorPoiA;
    }

    LR_SSTAMP(5);
    // ---- lookup tables of the proposal ----
    const double constP = lr_build_tables_segments(&scratch, peL, peM, PKL, PKM, a.br_length, a.log_br, cfg.model,
                                                   n_bins, a.n_cls, a.H, table, lane,
                                                   a.unit != 0, cfg.frac_birth, cfg.frac_death, table_es);

    LR_SSTAMP(6);
    // ---- back into the state registers ----
    st.L = L, st.M = M, st.tL = tL, st.tM = tM;
    st.pL = pL, st.pM = pM, st.ptL = ptL, st.ptM = ptM;
    st.eL = eL, st.eM = eM, st.peL = peL, st.peM = peM;
    {
        // scalar slots: branch-free select chains (a switch over the lane id runs every case under its own exec mask)
        double so = 0.0;
        so = (lane == LR_S_LIKA) ? likA : so;
        so = (lane == LR_S_PRIORA) ? priorA : so;
        so = (lane == LR_S_PRIORPOIA) ? priorPoiA : so;
        so = (lane == LR_S_GRATE_L) ? g0 : so;
        so = (lane == LR_S_GRATE_M) ? g1 : so;
        so = (lane == LR_S_POI) ? poi : so;
        so = (lane == LR_S_HASTING) ? hasting : so;
        so = (lane == LR_S_PRIOR_P) ? priorP : so;
        so = (lane == LR_S_PRIORPOI_P) ? priorPoi : so;
        so = (lane == LR_S_CONST_P) ? constP : so;
        so = (lane == LR_S_CONST_A) ? constA : so;
        so = (lane == LR_S_LIK_P) ? lik_p : so;
        so = (lane == LR_S_LOG_G0) ? lg0 : so;
        so = (lane == LR_S_LOG_G1) ? lg1 : so;
        so = (lane == LR_S_LOG_POI) ? lpoi : so;
        so = (lane == LR_S_LOG_U) ? log_u_next : so;
        st.sc = so;
        int io = 0;
        io = (lane == LR_I_KL) ? KL : io;
        io = (lane == LR_I_KM) ? KM : io;
        io = (lane == LR_I_PKL) ? PKL : io;
        io = (lane == LR_I_PKM) ? PKM : io;
        io = (lane == LR_I_GIBBS) ? gibbs : io;
        io = (lane == LR_I_INVALID) ? invalid : io;
        io = (lane == LR_I_IT_LO) ? (int)(uint32_t)it : io;
        io = (lane == LR_I_IT_HI) ? (int)(uint32_t)(it >> 32) : io;
        io = (lane == LR_I_ACCEPTED) ? n_acc : io;
        io = (lane == LR_I_MOVE) ? move_kind : io;
        io = (lane == LR_I_NEXT_LO) ? (int)(uint32_t)next_sample : io;
        io = (lane == LR_I_NEXT_HI) ? (int)(uint32_t)(next_sample >> 32) : io;
        io = (lane == LR_I_SLOT) ? trace_slot : io;
        st.isc = io;
    }
    LR_SSTAMP(7);
#ifdef LR_DIAG
    if (lane == 0 && c < 4096) lr_diag_step[c * 12 + 8] = move_kind;
#endif
}

// the step of chain c with its state in global memory: load, sum the tile partials in tile order, step, store

// ---- parametric samplers: one iteration of DDRate.py's loop (DD:194-239; sampler 1) or of trend_rate.py's
// (trend_rate.py:160-195; sampler 2) for one chain -----------------------------------------------------------
// Same pipeline position as lr_chain_step_core: decide the pending proposal with the scanned likelihood, write the
// trace row, propose the next parameter vector, evaluate its prior and build its lookup tables.
// State: lane j < NPAR of st.L = accepted parameter j, of st.pL = proposed parameter j; scalars in st.sc / st.isc.
// `aux` = the per-bin array the rate map needs: DT (DDRate) or the normalised covariate TREND (trend_rate).
__device__ __forceinline__ void lr_dd_step_core(lr_chain_regs& st, const lr_step_args& a, int mode, int c, int lane,
                                                double lik_sum, double2* table, int table_es = 2) {
    const lr_mcmc_config& cfg = a.cfg;
    const int C = cfg.n_chains;
    const bool trend = cfg.sampler == 2;
    const int npar = trend ? LR_TR_NPAR : LR_DD_NPAR;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const double origin = cfg.t0, present = cfg.dd_present;
    const double k0 = a.dd_consts[0], log_k0 = a.dd_consts[1];
    const double* aux = a.br_length;
    double A = st.L;
    const double P0 = st.pL;
    const double sc = st.sc;
    const int isc = st.isc;
    double likA = lr_bcast(sc, LR_S_LIKA), priorA = lr_bcast(sc, LR_S_PRIORA), lik_p = lr_bcast(sc, LR_S_LIK_P);
    uint64_t it = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_HI) << 32);
    int n_acc = lr_bcast_i(isc, LR_I_ACCEPTED);
    uint64_t next_sample = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_HI) << 32);
    int trace_slot = lr_bcast_i(isc, LR_I_SLOT);
    // wave-uniform draws in one Philox call (see lr_chain_step_core): lane 0 acceptance uniform of `it`, lane 1 the move
    // selector and lane 2 the sliding-window uniform of the iteration about to be proposed
    lr_u2 ud;
    {
        const uint64_t it_prop = (mode == 1) ? it : it + 1;
        const uint32_t purpose = trend ? (lane == 0 ? LR_P_TR_ACCEPT : LR_P_TR_MOVE)
                                       : (lane == 0 ? LR_P_DD_ACCEPT : (lane == 1 ? LR_P_DD_MOVE : LR_P_DD_SLIDE));
        ud = lr_pair(rng, lane == 0 ? it : it_prop, purpose, 0u);
    }
    if (mode == 1) {
        likA = lik_sum;                                                        // DD:184-186, trend_rate.py:150-151
        priorA = trend ? lr_trend_prior(A, lane) : lr_dd_prior(A, origin, present, k0, log_k0, lane);
    } else {
        const double hasting = lr_bcast(sc, LR_S_HASTING), priorP = lr_bcast(sc, LR_S_PRIOR_P);
        const double u = lr_bcast(ud.a, 0);
        const double lik = lik_sum;
        const bool ok = ((lik - likA) + (priorP - priorA) + hasting > log(u)) || it == 0;   // DD:211, trend_rate.py:176
        lik_p = lik;
        if (ok) A = P0, likA = lik, priorA = priorP, n_acc += 1;
        if (it == next_sample) {                                               // DD:221
            const int slot = trace_slot;
            trace_slot += 1;
            next_sample += (uint64_t)cfg.s_freq;
            if (slot < cfg.n_trace_slots) {
                double* row = a.trace + ((size_t)slot * C + c) * LR_TRACE_W;
                double h = __longlong_as_double(0x7ff8000000000000LL);
                if (lane == 0) h = (double)it;
                if (lane == 1) h = likA + priorA;
                if (lane == 2) h = likA;
                if (lane == 3) h = priorA;
                const double Aj = __shfl(A, (lane - 4) & (LR_WAVE - 1));        // rare path (sampling only)
                if (lane >= 4 && lane < 4 + npar) h = Aj;
                for (int j = lane; j < LR_TRACE_W; j += LR_WAVE) row[j] = (j == lane) ? h : __longlong_as_double(0x7ff8000000000000LL);
            }
        }
        it += 1;
    }
    // ---- propose iteration `it` ----
    double P = A, hasting = 0.0;
    int move_kind;
    if (trend) {
        // trend_rate.py:165-169: 33 % additive normal step on the slopes, else the vector multiplier
        const double rr = lr_bcast(ud.a, 1);
        double f_mult, f_norm;
        lr_trend_update_freq(cfg.m_birth, cfg.m_death, lane, &f_mult, &f_norm);
        const lr_u2 d = lr_pair(rng, it, LR_P_TR_MULT, lane);
        if (rr < .33) {
            const double z = lr_normal(rng, it, LR_P_TR_NORM, lane);               // update_normal_nobound_vec (lib:140-146)
            if (lane < npar && d.a < f_norm) P = A + z * .001;
            move_kind = 1;
        } else {
            hasting = lr_wave_multiplier(P, npar, d.a < f_mult, d.b, a.mult_l, lane);   // lib:156-165
            move_kind = 0;
        }
    } else {
        // DD:195-207
        const lr_u2 rr{lr_bcast(ud.a, 1), lr_bcast(ud.b, 1)};
        if (rr.b < 0.1 && (cfg.m_birth == 2 || cfg.m_death == 2)) {
            // update_sliding_win(x0, m=0, M=PRESENT, d=1.5) (lib:124-128)
            double ii = lr_bcast(A, 2) + (lr_bcast(ud.a, 2) - .5) * 1.5;
            if (ii > present) ii = present - (ii - present);
            ii = fabs(ii);
            if (lane == 2) P = ii;
            if (cfg.m_death == -1) {
                const double z = lr_normal(rng, it, LR_P_DD_SLIDE, 1);              // update_normal_nobound(k, d=0.2) (lib:136-138)
                if (lane == 1) P = A + z * 0.2;
            }
            move_kind = 1;
        } else {
            const double f = lr_dd_update_freq(cfg.m_birth, cfg.m_death, lane);
            const lr_u2 d = lr_pair(rng, it, LR_P_DD_MULT, lane);
            hasting = lr_wave_multiplier(P, npar, d.a < f, d.b, a.mult_l, lane);   // lib:156-165
            move_kind = 0;
        }
    }
    double priorP;
    if (trend) {
        priorP = lr_trend_prior(P, lane);
        const lr_trend_params tp = lr_trend_unpack(P);
        lr_rates_build_tables_wave([&](int b, double* br, double* dr) { lr_trend_bin_rates(tp, aux[b], cfg.m_birth, cfg.m_death, br, dr); },
                                   cfg.n_bins, a.H, table, lane, a.unit != 0, cfg.frac_birth, cfg.frac_death, table_es);
    } else {
        priorP = lr_dd_prior(P, origin, present, k0, log_k0, lane);
        const lr_dd_params pp = lr_dd_unpack(P);
        lr_dd_build_tables_wave(pp, aux, cfg.m_birth, cfg.m_death, cfg.n_bins, a.H, table, lane, a.unit != 0,
                                cfg.frac_birth, cfg.frac_death, table_es);
    }
    st.L = A, st.pL = P;
    {
        double so = 0.0;
        so = (lane == LR_S_LIKA) ? likA : so;
        so = (lane == LR_S_PRIORA) ? priorA : so;
        so = (lane == LR_S_HASTING) ? hasting : so;
        so = (lane == LR_S_PRIOR_P) ? priorP : so;
        so = (lane == LR_S_LIK_P) ? lik_p : so;
        st.sc = so;
        int io = 0;
        io = (lane == LR_I_KL || lane == LR_I_KM || lane == LR_I_PKL || lane == LR_I_PKM) ? npar : io;
        io = (lane == LR_I_IT_LO) ? (int)(uint32_t)it : io;
        io = (lane == LR_I_IT_HI) ? (int)(uint32_t)(it >> 32) : io;
        io = (lane == LR_I_ACCEPTED) ? n_acc : io;
        io = (lane == LR_I_MOVE) ? move_kind : io;
        io = (lane == LR_I_NEXT_LO) ? (int)(uint32_t)next_sample : io;
        io = (lane == LR_I_NEXT_HI) ? (int)(uint32_t)(next_sample >> 32) : io;
        io = (lane == LR_I_SLOT) ? trace_slot : io;
        st.isc = io;
    }
}

__device__ __forceinline__ void lr_chain_step_body(const lr_step_args& a, int mode, int c, int lane,
                                                   lr_seg_scratch* scratch_p) {
    LR_SSTAMP(0);
    lr_chain_regs st;
    double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
    int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
    lr_chain_load(st, S, I, lane);
    double part = 0.0;
    for (int t = lane; t < a.tiles; t += LR_WAVE) part += a.partials[(size_t)t * a.cfg.n_chains + c];
    const double lik_sum = lr_wave_sum(part);
    if (a.cfg.sampler != 0) lr_dd_step_core(st, a, mode, c, lane, lik_sum, lr_chain_table(a, c));
    else lr_chain_step_core(st, a, mode, c, lane, scratch_p, lik_sum, lr_chain_table(a, c));
    lr_chain_store(st, S, I, lane);
}

#define LR_STEP_WAVES (LR_SCAN_THREADS / LR_WAVE)

// chains [chain_base, chain_base + n_sub): one wave per chain, LR_STEP_WAVES chains per block
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_chain_step_kernel(lr_step_args a, int mode, int chain_base,
                                                                        int n_sub) {
    __shared__ lr_seg_scratch scratch[LR_STEP_WAVES];
    const int wave = threadIdx.x / LR_WAVE, lane = threadIdx.x & (LR_WAVE - 1);
    const int j = blockIdx.x * LR_STEP_WAVES + wave;
    if (j < n_sub) lr_chain_step_body(a, mode, chain_base + j, lane, &scratch[wave]);
}

// Fused launch of the software-pipelined engine: the first `step_blocks` blocks run the chain step of the
// half whose scan finished in the PREVIOUS launch, all other blocks scan the lineages for the other half.
// The step is latency bound (a serial stream per wave) and the scan is throughput bound, so they overlap.
struct lr_fused_args {
    const double* ts;
    const double* te;
    long long n, chunk;
    double t0;
    int n_bins, tiles;
    int scan_base, scan_n;     // chains scanned by this launch
    int step_base, step_n;     // chains stepped by this launch
    int step_blocks;
};

template <int CB, int H, bool UNIT>
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_fused_iter_kernel(lr_step_args a, lr_fused_args f) {
    extern __shared__ double2 lds[];
    // the step blocks do not stage tables: their per-wave scratch lives in the same dynamic LDS (a static array on top
    // of the tables would cost the scan blocks one resident block per CU)
    lr_seg_scratch* scratch = reinterpret_cast<lr_seg_scratch*>(lds);
    const int bid = blockIdx.x;
    if (bid < f.step_blocks) {
        const int wave = threadIdx.x / LR_WAVE, lane = threadIdx.x & (LR_WAVE - 1);
        const int j = bid * LR_STEP_WAVES + wave;
        if (j < f.step_n) lr_chain_step_body(a, 0, f.step_base + j, lane, &scratch[wave]);
        return;
    }
    int group, tile;
    if ((f.step_blocks & 7) == 0) {
        lr_xcd_remap(bid - f.step_blocks, f.tiles, (f.scan_n + CB - 1) / CB, &group, &tile);
    } else {
        const int sb = bid - f.step_blocks;
        tile = sb % f.tiles, group = sb / f.tiles;
    }
    const double2* tables = a.tables + (size_t)f.scan_base * a.tab_stride;
    double* partials = const_cast<double*>(a.partials) + f.scan_base;
    if (UNIT)
        lr_scan_unit_body<CB, H>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables, f.scan_n, f.chunk,
                                 partials, a.cfg.n_chains);
    else
        lr_scan_fast_body<CB, H>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables, f.scan_n, f.chunk,
                                 partials, a.cfg.n_chains);
}

// ------------------------------------------------------------------------------------------
// Persistent engine for unit-resolution data and many chains (>= one pair of chains per block slot).
//
// One 512-thread block owns TWO chains for the whole call and iterates inside the kernel: all 8 waves
// scan every lineage against the pair's lookup table in LDS, the sums are reduced in the block, then waves
// 0 and 1 run the chain step of "their" chain on register-resident state and rebuild the table in LDS while
// the others wait at the barrier.  No launch per iteration, no inter-block communication at all (a block
// never needs another block's data), state and tables touch global memory only at entry and exit.
// The lineages are read as one packed uint16 per lineage (table indices of its birth and death bins,
// lr_pack_lineages_kernel), 8 lineages per 16-byte load: 200 KB per pass for 100k lineages, L2 resident.
// ------------------------------------------------------------------------------------------
#ifndef LR_PERSIST_THREADS
#define LR_PERSIST_THREADS 512
#endif
// spare zero-filled 16-byte groups behind the packed lineage indices (index 0 = sentinel table entries, contribution 0):
// room for the trips the four-chain kernel moves between waves and for its prefetch past the end
#define LR_P4_MAX_GIVE 64
// sized for the widest stride (16 scanner waves x 64 lanes): the takers' extra trips reach group (k_tot + give) * stride
#define LR_IDX_SPARE ((LR_P4_MAX_GIVE + 2) * 1024)
#ifndef LR_PERSIST_MINWAVES
#define LR_PERSIST_MINWAVES 4
#endif

// p4 = 1: layout for the four-chain kernel.  Its 14 scanner waves stride over the groups (wave slot = (group % 896) /
// 64, trip = group / 896) and do not all score the same number of trips (lr_p4_shares): the trips a slot gives up,
// taken in slot / trip order, are stored as the extra trips of the taking slots, in slot / trip order, where those
// waves simply keep striding.  The hand-over costs the scan loop nothing.
__global__ void lr_pack_lineages_kernel(const double* __restrict__ ts, const double* __restrict__ te, long long n,
                                        double t0, int n_bins, int p4, int k_tot, lr_p4_shares sh,
                                        unsigned short* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = min(max(__double2int_rz(floor(ts[i]) - t0), -1), n_bins);
    const int b = min(max(__double2int_rz(ceil(te[i]) - t0), 0), n_bins + 1);
    long long g = i >> 3;
    if (p4) {
        const int stride = sh.n_slots * 64;
        const int slot = (int)(g % stride) / 64, trip = (int)(g / stride), lane = (int)(g % 64);
        if (sh.delta[slot] < 0 && trip >= k_tot + sh.delta[slot]) {
            int r = trip - (k_tot + sh.delta[slot]);                     // rank of this trip among all given trips
            for (int q = 0; q < slot; ++q) r += sh.delta[q] < 0 ? -sh.delta[q] : 0;
            int to = 0;
            for (; to < sh.n_slots; ++to) {
                const int extra = sh.delta[to] > 0 ? sh.delta[to] : 0;
                if (r < extra) break;
                r -= extra;
            }
            g = (long long)(k_tot + r) * stride + to * 64 + lane;
        }
    }
    out[g * 8 + (i & 7)] = (unsigned short)((a + 1) | (b << 8));
}

// Scan of all lineages against ONE pair table by `n_scan` threads (this thread is number `sid`): the inner loop
// of the persistent engines.  8 lineages per 16-byte load, next load in flight while the current one is scored.
template <int H, int UNROLL = 1>
__device__ __forceinline__ void lr_persist_scan_pair(const char* __restrict__ lbase, const uint4* __restrict__ idx8,
                                                     long long n8, long long sid, int n_scan, double* acc0_,
                                                     double* acc1_) {
    double acc0 = *acc0_, acc1 = *acc1_;
    // 32-bit loop arithmetic (n8 = N / 8 < 2^31): a 64-bit compare and add per trip are two instructions each
    const int n = (int)n8;
    int i = (int)sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) w = idx8[i];
    // UNROLL = 2 saves the register rotation of the prefetched index word (+1 % on long scans) at the price of a dozen
    // spills around the chain-step call, which the short-scan configurations feel: the four-chain kernel uses it
#pragma unroll UNROLL
    while (i < n) {
        const uint4 cur = w;
        const int nx = i + n_scan;
        if (nx < n) w = idx8[nx];
        const unsigned int words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int v = words[k];
            const double2 s0 = *reinterpret_cast<const double2*>(lbase + ((v << 4) & 0xff0u));
            const double2 e0 = *reinterpret_cast<const double2*>(lbase + ((v >> 4) & 0xff0u) + H * 16);
            const double2 s1 = *reinterpret_cast<const double2*>(lbase + ((v >> 12) & 0xff0u));
            const double2 e1 = *reinterpret_cast<const double2*>(lbase + ((v >> 20) & 0xff0u) + H * 16);
            acc0 += (s0.x + e0.x) + (s1.x + e1.x);
            acc1 += (s0.y + e0.y) + (s1.y + e1.y);
        }
        i = nx;
    }
    *acc0_ = acc0, *acc1_ = acc1;
}

// the chain step of the persistent kernel as a real call: its ~120 live registers then do not add to the scan
// loop's, and both fit the 128-VGPR budget of 4 waves per SIMD without spilling
__device__ __attribute__((noinline)) void lr_persist_step(const lr_step_args* a, int c, int lane,
                                                          lr_seg_scratch* scratch, double* st_f64, int* st_i32,
                                                          double lik, double2* table, int table_es) {
    lr_chain_regs st;
    lr_chain_load(st, st_f64, st_i32, lane);
    if (a->cfg.sampler != 0) lr_dd_step_core(st, *a, 0, c, lane, lik, table, table_es);
    else lr_chain_step_core(st, *a, 0, c, lane, scratch, lik, table, table_es);
    lr_chain_store(st, st_f64, st_i32, lane);
}

// T = threads per block: 512 (two blocks share a CU) or, when there are no more blocks than CUs anyway (at most 512
// chains), 1024 - all 16 waves of the CU scan the one pair.
template <int H, int T>
__global__ __launch_bounds__(T, LR_PERSIST_MINWAVES) void lr_persist_kernel(
    const lr_step_args* __restrict__ ap /* in global memory: taking the address of a by-value kernel argument would
                                           copy it to scratch */,
    const uint4* __restrict__ idx8, long long n8, lr_p4_shares sh, long long n_iters, int prio_shift) {
    const lr_step_args& a = *ap;
    __shared__ double2 tab[2 * H];  // the pair table: S' entries [0,H), E' entries [H,2H); (.x, .y) = (chain 0, chain 1)
    __shared__ double red[T / LR_WAVE][2];
    __shared__ lr_seg_scratch scratch[2];
    // the two chains' state rows live in LDS between iterations (registers are needed by the step itself)
    __shared__ double st_f64[2][LR_STATE_ROWS * LR_ROW];
    __shared__ int st_i32[2][LR_ISTATE_ROWS * LR_ROW];
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int c0 = blockIdx.x * 2;
    const int c = c0 + wave;
    const bool stepper = (wave < 2) && (c < a.cfg.n_chains);
    double2* gpair = lr_chain_table(a, c0);  // the pair (c0, c0+1) shares one table in the unit layout
    if (stepper) {
        const double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        const int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        for (int r = 0; r < LR_STATE_ROWS; ++r) st_f64[wave][r * LR_ROW + lane] = S[r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) st_i32[wave][r * LR_ROW + lane] = I[r * LR_ROW + lane];
    }
    for (int i = tid; i < 2 * H; i += T) tab[i] = gpair[i];
    __syncthreads();
    const char* lbase = reinterpret_cast<const char*>(tab);
    // unequal shares of the waves (see lr_persist4_kernel): older waves 0..3 take trips from the younger 4..7
    const long long n8w = sh.delta[0] != 0 ? ((n8 + T - 1) / T + sh.delta[wave]) * T : n8;
    const int grp = (blockIdx.x >> 8) & 1;
#ifdef LR_DIAG
    unsigned long long d_t0 = 0, d_t1 = 0, d_t2 = 0, d_scan = 0, d_red = 0, d_step = 0;
#endif
    for (long long iter = 0; iter < n_iters; ++iter) {
        // Two blocks share a CU; the one dispatched second is the younger wave on every SIMD and loses issue
        // arbitration to its older neighbour all the time (+30 % per iteration, measured).  Both read the same
        // 100 MHz clock, so slicing it gives them complementary priorities that even the two out
        // (MI355X_MICROARCH.md "Two waves per SIMD" item 4; block id >= 256 = second dispatch: speed only).
        if (prio_shift > 0) {
            if (((wall_clock64() >> prio_shift) + grp) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#ifdef LR_DIAG
        d_t0 = wall_clock64();
#endif
        double acc0 = 0.0, acc1 = 0.0;
        lr_persist_scan_pair<H>(lbase, idx8, n8w, tid, T, &acc0, &acc1);
#ifdef LR_DIAG
        if (lane == 0 && blockIdx.x < 512) atomicAdd(&lr_diag_step[20000 + blockIdx.x * 8 + wave], wall_clock64() - d_t0);
#endif
#ifdef LR_DIAG
        d_t1 = wall_clock64();
#endif
        const double s0 = lr_wave_sum(acc0), s1 = lr_wave_sum(acc1);
        if (lane == 0) red[wave][0] = s0, red[wave][1] = s1;
        __syncthreads();  // every scan is done: sums visible, table free to be rebuilt
#ifdef LR_DIAG
        d_t2 = wall_clock64();
#endif
        if (stepper) {
            double lik = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < T / LR_WAVE; ++w2) lik += red[w2][wave];
            lr_persist_step(ap, c, lane, &scratch[wave], st_f64[wave], st_i32[wave], lik,
                            reinterpret_cast<double2*>(reinterpret_cast<double*>(tab) + wave), 2);
        }
        __syncthreads();  // new tables ready
#ifdef LR_DIAG
        d_scan += d_t1 - d_t0, d_red += d_t2 - d_t1, d_step += wall_clock64() - d_t2;
#endif
    }
#ifdef LR_DIAG
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = lr_diag_step + 4096 * 12 - 4096 + (blockIdx.x * 8 + wave) % 4096;
        (void)o;
    }
    if (tid == 0 && blockIdx.x < 340) {
        lr_diag_step[blockIdx.x * 12 + 9] = d_scan, lr_diag_step[blockIdx.x * 12 + 10] = d_red;
        lr_diag_step[blockIdx.x * 12 + 11] = d_step;
    }
    if (tid == 0 && blockIdx.x < 1024) lr_diag_step[2048 * 12 + blockIdx.x * 2 + 1] = wall_clock64();
#endif
    if (stepper) {
        double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        for (int r = 0; r < LR_STATE_ROWS; ++r) S[r * LR_ROW + lane] = st_f64[wave][r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) I[r * LR_ROW + lane] = st_i32[wave][r * LR_ROW + lane];
    }
    for (int i = tid; i < 2 * H; i += T) gpair[i] = tab[i];  // pending tables back to global
}

// Four chains per 1024-thread block (one block per CU), two pairs in ping-pong: while waves 0 and 1 run the chain
// steps of one pair, waves 2..15 scan the lineages for the OTHER pair, so the latency-bound step always hides
// under a scan of the same block and the LDS pipe never waits for it.  Used when there are enough chains to give
// every CU four (cfg4: 1024 chains = 256 blocks).
#define LR_P4_THREADS 1024
#define LR_P4_SCANNERS ((LR_P4_THREADS / LR_WAVE - 2) * LR_WAVE)
template <int H>
__global__ __launch_bounds__(LR_P4_THREADS, 4) void lr_persist4_kernel(const lr_step_args* __restrict__ ap,
                                                                       const uint4* __restrict__ idx8, long long n8,
                                                                       lr_p4_shares sh, long long n_iters) {
    const lr_step_args& a = *ap;
    constexpr int NW = LR_P4_THREADS / LR_WAVE;          // 16 waves: 2 steppers + 14 scanners
    __shared__ double2 tab[2][2 * H];                     // pair tables
    __shared__ double red[2][NW][2];                      // [pair][scanner wave][chain of the pair]
    __shared__ lr_seg_scratch scratch[2];
    __shared__ double st_f64[4][LR_STATE_ROWS * LR_ROW];
    __shared__ int st_i32[4][LR_ISTATE_ROWS * LR_ROW];
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int c0 = blockIdx.x * 4;
    const int C = a.cfg.n_chains;
    if (wave < 4 && c0 + wave < C) {
        const int c = c0 + wave;
        const double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        const int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        for (int r = 0; r < LR_STATE_ROWS; ++r) st_f64[wave][r * LR_ROW + lane] = S[r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) st_i32[wave][r * LR_ROW + lane] = I[r * LR_ROW + lane];
    }
    double2* g0 = lr_chain_table(a, c0);
    double2* g1 = lr_chain_table(a, c0 + 2);             // tables are allocated for whole groups of cb >= 4 chains
    for (int i = tid; i < 2 * H; i += LR_P4_THREADS) tab[0][i] = g0[i], tab[1][i] = g1[i];
    __syncthreads();
    const bool scanner = wave >= 2;
    const int sid = tid - 2 * LR_WAVE;
    // The SIMD issue arbiter serves its oldest wave first: with equal shares the scanner waves of a SIMD finish one
    // after the other (5.0 / 6.4 / 7.9 / 9.5 us per phase, measured with in-kernel stamps), the youngest runs the tail
    // alone, and SIMDs 0, 1 carry the stepper waves on top.  So the waves get unequal shares (lr_p4_shares) chosen to
    // make them finish together.  lr_pack_lineages_kernel stores the moved groups where the takers keep striding, so
    // this is only a per-wave end of the loop: a fixed partition, the summation order - and with it bitwise
    // reproducibility - stays.
    const int k_tot = (int)((n8 + LR_P4_SCANNERS - 1) / LR_P4_SCANNERS);
    const int k_mine = k_tot + (scanner ? sh.delta[wave - 2] : 0);
    // equal shares (short scans): the true end, so that the ragged last trip costs only the lanes that have a group
    bool any_shift = false;
#pragma unroll
    for (int q = 0; q < 14; ++q) any_shift |= sh.delta[q] != 0;
    const long long n8w = any_shift ? (long long)k_mine * LR_P4_SCANNERS : n8;
    // prologue: pair 0's pending proposal is scanned so that phase A can step it
    if (scanner) {
        double s0 = 0.0, s1 = 0.0;
        lr_persist_scan_pair<H, 2>(reinterpret_cast<const char*>(tab[0]), idx8, n8w, sid, LR_P4_SCANNERS, &s0, &s1);
        s0 = lr_wave_sum(s0), s1 = lr_wave_sum(s1);
        if (lane == 0) red[0][wave][0] = s0, red[0][wave][1] = s1;
    }
    __syncthreads();
    for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // phase ph: steppers advance pair `ph`, scanners score pair `1 - ph`
            if (scanner) {
                double s0 = 0.0, s1 = 0.0;
#ifdef LR_DIAG
                const unsigned long long dq0 = wall_clock64();
#endif
                lr_persist_scan_pair<H, 2>(reinterpret_cast<const char*>(tab[1 - ph]), idx8, n8w, sid, LR_P4_SCANNERS, &s0, &s1);
#ifdef LR_DIAG
                if (lane == 0 && blockIdx.x < 64) atomicAdd(&lr_diag_step[20000 + blockIdx.x * 16 + wave], wall_clock64() - dq0);
#endif
                s0 = lr_wave_sum(s0), s1 = lr_wave_sum(s1);
                if (lane == 0) red[1 - ph][wave][0] = s0, red[1 - ph][wave][1] = s1;
            } else {
                const int c = c0 + 2 * ph + wave;
                if (c < C) {
                    double lik = 0.0;
#pragma unroll
                    for (int w2 = 2; w2 < NW; ++w2) lik += red[ph][w2][wave];
                    lr_persist_step(ap, c, lane, &scratch[wave], st_f64[2 * ph + wave], st_i32[2 * ph + wave], lik,
                                    reinterpret_cast<double2*>(reinterpret_cast<double*>(tab[ph]) + wave), 2);
                }
            }
            __syncthreads();
        }
    }
    if (wave < 4 && c0 + wave < C) {
        const int c = c0 + wave;
        double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        for (int r = 0; r < LR_STATE_ROWS; ++r) S[r * LR_ROW + lane] = st_f64[wave][r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) I[r * LR_ROW + lane] = st_i32[wave][r * LR_ROW + lane];
    }
    for (int i = tid; i < 2 * H; i += LR_P4_THREADS) g0[i] = tab[0][i], g1[i] = tab[1][i];
}

__global__ void lr_store_args_kernel(lr_step_args a, lr_step_args* dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = a;
}

// log(br_length) once per engine (data constant used by models 0/1)
__global__ void lr_log_br_kernel(const double* __restrict__ br, int n_bins, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_bins) out[b] = (br && br[b] > 0.0) ? log(br[b]) : 0.0;
}

// initial state (LRF:580-583 or the caller's runMCMC argument) -> state rows and the tables of
// the initial state (so that the first scan evaluates likA, LRF:224-226)
__global__ __launch_bounds__(LR_WAVE) void lr_chain_init_kernel(lr_step_args a, const double* L0, const double* M0,
                                                                const double* tL0, const double* tM0,
                                                                const int* KL0, const int* KM0, int kmax) {
    __shared__ lr_seg_scratch scratch;
    const lr_mcmc_config& cfg = a.cfg;
    const int c = blockIdx.x, lane = threadIdx.x, n_bins = cfg.n_bins;
    double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
    int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    if (cfg.sampler != 0) {
        // initial parameter vector of the parametric samplers (DD:151-161, trend_rate.py:141-147) or the caller's
        // [C, kmax] rows; its tables, so that the first scan evaluates likA (DD:184-186)
        const bool trend = cfg.sampler == 2;
        const int npar = trend ? LR_TR_NPAR : LR_DD_NPAR;
        double A = 0.0;
        if (L0) {
            if (lane < npar) A = L0[(size_t)c * kmax + lane];
        } else if (trend) {
            const double init[LR_TR_NPAR] = {.1, .1, 0.0, 0.0, 1.0, 1.0};
            if (lane < npar) A = init[lane];
        } else {
            const double x0 = cfg.dd_present - (cfg.t0 + cfg.dd_present) / 2.0;       // PRESENT - np.mean([ORIGIN, PRESENT])
            const double init[LR_DD_NPAR] = {0.5, 1.5, x0, 10.0, 20000.0, cfg.dd_init_death, 1.0, 1.0};
            if (lane < npar) A = init[lane];
        }
        if (trend) {
            const lr_trend_params tp = lr_trend_unpack(A);
            const double* aux = a.br_length;
            lr_rates_build_tables_wave([&](int b, double* br, double* dr) { lr_trend_bin_rates(tp, aux[b], cfg.m_birth, cfg.m_death, br, dr); },
                                       n_bins, a.H, lr_chain_table(a, c), lane, a.unit != 0, cfg.frac_birth, cfg.frac_death, 2);
        } else {
            const lr_dd_params pp = lr_dd_unpack(A);
            lr_dd_build_tables_wave(pp, a.br_length, cfg.m_birth, cfg.m_death, n_bins, a.H, lr_chain_table(a, c), lane,
                                    a.unit != 0, cfg.frac_birth, cfg.frac_death, 2);
        }
        for (int r = 0; r < LR_STATE_ROWS; ++r) S[r * LR_ROW + lane] = 0.0;
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) I[r * LR_ROW + lane] = 0;
        S[LR_ROW_L * LR_ROW + lane] = A, S[LR_ROW_PL * LR_ROW + lane] = A;
        int io = 0;
        if (lane == LR_I_KL || lane == LR_I_PKL || lane == LR_I_KM || lane == LR_I_PKM) io = npar;
        I[LR_IROW_SCALARS * LR_ROW + lane] = io;
        return;
    }
    double L = 0.0, M = 0.0, tL = 0.0, tM = 0.0;
    int KL = 1, KM = 1;
    if (L0) {
        KL = KL0[c], KM = KM0[c];
        if (lane < KL) L = L0[(size_t)c * kmax + lane];
        if (lane < KM) M = M0[(size_t)c * kmax + lane];
        if (lane <= KL) tL = tL0[(size_t)c * (kmax + 1) + lane];
        if (lane <= KM) tM = tM0[(size_t)c * (kmax + 1) + lane];
    } else {
        const double l0 = lr_gamma(rng, 0, LR_P_INIT, 0, 2.0) * 2.0;   // np.random.gamma(2,2,1), LRF:580
        const double m0 = lr_gamma(rng, 0, LR_P_INIT, 64, 2.0) * 2.0;  // LRF:581
        if (lane == 0) L = l0, M = m0, tL = tM = cfg.start_time;
        if (lane == 1) tL = tM = cfg.end_time;
    }
    // the initial index comes from get_rate_index on the RAW times: round, not floor (LRF:224-225, 129)
    const int eL = lr_wave_edges(tL, 1), eM = lr_wave_edges(tM, 1);
    double logL, logM;
    lr_stage_segments(&scratch, L, M, eL, eM, KL, KM, lane, &logL, &logM);
    const double constA = lr_build_tables_segments(&scratch, eL, eM, KL, KM, a.br_length, a.log_br, cfg.model, n_bins,
                                                   a.n_cls, a.H, lr_chain_table(a, c), lane,
                                                   a.unit != 0, cfg.frac_birth, cfg.frac_death);
    S[LR_ROW_L * LR_ROW + lane] = L, S[LR_ROW_M * LR_ROW + lane] = M;
    S[LR_ROW_TL * LR_ROW + lane] = tL, S[LR_ROW_TM * LR_ROW + lane] = tM;
    S[LR_ROW_PL * LR_ROW + lane] = L, S[LR_ROW_PM * LR_ROW + lane] = M;
    S[LR_ROW_PTL * LR_ROW + lane] = tL, S[LR_ROW_PTM * LR_ROW + lane] = tM;
    I[LR_IROW_EL * LR_ROW + lane] = eL, I[LR_IROW_EM * LR_ROW + lane] = eM;
    I[LR_IROW_PEL * LR_ROW + lane] = eL, I[LR_IROW_PEM * LR_ROW + lane] = eM;
    const double poi0 = (cfg.poisson_HP == 0.0) ? 1.0 : cfg.poisson_HP;             // LRF:220-221
    double so = 0.0;
    if (lane == LR_S_GRATE_L || lane == LR_S_GRATE_M) so = 1.0;                       // LRF:222 (log = 0)
    if (lane == LR_S_POI) so = poi0;
    if (lane == LR_S_LOG_POI) so = log(poi0);
    if (lane == LR_S_CONST_A || lane == LR_S_CONST_P) so = constA;
    S[LR_ROW_SCALARS * LR_ROW + lane] = so;
    int io = 0;
    if (lane == LR_I_KL || lane == LR_I_PKL) io = KL;
    if (lane == LR_I_KM || lane == LR_I_PKM) io = KM;
    I[LR_IROW_SCALARS * LR_ROW + lane] = io;
}

// ---- host -------------------------------------------------------------------------------
static int lr_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int lr_pipeline_env() {
    static int env = lr_env_int("LR_PIPELINE", 1);
    return env;
}

// Partition layout of the engine.  LR_PARTS (default 2) independent partitions run on their own streams so
// that the ramp-up / drain of one partition's launches overlaps the other's; each partition with at least
// 2*cb chains is software-pipelined in two halves.  All boundaries are multiples of cb.
static int lr_partition(int n_chains, int cb, bool fused_ok, int base[LR_MAX_PARTS + 1], int hA[LR_MAX_PARTS],
                        bool pipelined[LR_MAX_PARTS]) {
    static const int want_parts = lr_env_int("LR_PARTS", 2);
    int parts = want_parts < 1 ? 1 : (want_parts > LR_MAX_PARTS ? LR_MAX_PARTS : want_parts);
    const int groups = (n_chains + cb - 1) / cb;
    const bool pipe = lr_pipeline_env() && fused_ok;
    while (parts > 1 && groups < parts * (pipe ? 2 : 1)) --parts;
    base[0] = 0;
    for (int p = 0; p < parts; ++p) {
        const int g = groups / parts + (p < groups % parts ? 1 : 0);
        base[p + 1] = base[p] + g * cb < n_chains ? base[p] + g * cb : n_chains;
    }
    base[parts] = n_chains;
    for (int p = 0; p < parts; ++p) {
        const int count = base[p + 1] - base[p];
        pipelined[p] = pipe && count >= 2 * cb;
        hA[p] = pipelined[p] ? ((count / 2 + cb - 1) / cb) * cb : count;
    }
    return parts;
}

// launch shape of the engine: as lr_plan_scan, but when the pipelined schedule applies the lineage tiles are
// sized so that the fused launches that run concurrently (one per partition: step blocks of one half + scan
// blocks of the other) fill the resident block slots of the chip (256 CUs x 4 blocks) exactly once.
static int lr_plan_engine(const lr_mcmc_config* cfg, lr_scan_plan* p) {
    int rc = lr_plan_scan(cfg->n_lineages, cfg->n_chains, cfg->n_bins, cfg->model, cfg->unit_resolution, p);
    if (rc) return rc;
    int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
    bool pipelined[LR_MAX_PARTS];
    const int parts = lr_partition(cfg->n_chains, p->cb, lr_fused_supported(*p), base, hA, pipelined);
    if (!pipelined[0]) return LR_OK;
    const int count = base[1] - base[0];
    const int half = hA[0] > count - hA[0] ? hA[0] : count - hA[0];
    const int groups_half = (half + p->cb - 1) / p->cb;
    const int step_blocks = (half + (LR_SCAN_THREADS / LR_WAVE) - 1) / (LR_SCAN_THREADS / LR_WAVE);
    static const int slots = lr_env_int("LR_SLOTS", 256 * 4);
    long long tiles = (slots / parts - step_blocks) / groups_half;
    const long long unit = 2 * LR_SCAN_THREADS;
    const long long max_tiles = (cfg->n_lineages + 4 * unit - 1) / (4 * unit);
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    long long chunk = lr_align_up64((cfg->n_lineages + tiles - 1) / tiles, unit);
    tiles = (cfg->n_lineages + chunk - 1) / chunk;
    p->tiles = (int)tiles;
    p->chunk = chunk;
    return LR_OK;
}

// Persistent engine (lr_persist_kernel): needs unit-resolution tables with byte-sized indices and an
// instantiated table size.  cfg->engine_mode 1 / 2 / 3 force the launch-based / persistent / four-chain persistent
// engine (2 and 3 still need the prerequisites); auto picks the persistent kernel unless the chains are too few for the lineage count: a block
// scans ALL lineages for its two chains, so with few chains and very long inputs the tiled launch-based scan,
// which spreads one chain group over many CUs, is faster.
static bool lr_persist_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    static const int env = lr_env_int("LR_PERSIST", -1);   // debugging override: 0 off, 1 on
    if (env == 0 || cfg->engine_mode == 1) return false;
    if (!p.unit || cfg->n_bins + 1 > 255 || p.cb < 2) return false;
    if (cfg->n_lineages >= (1ll << 33)) return false;   // the scan loop counts 16-byte index groups in 32 bits
    if (p.H != 40 && p.H != 72 && p.H != 136 && p.H != 264) return false;
    if (env == 1 || cfg->engine_mode == 2 || cfg->engine_mode == 3) return true;
    const double n = (double)cfg->n_lineages, c = (double)cfg->n_chains;
    const int blocks = (cfg->n_chains + 1) / 2;
    const double rounds = (double)((blocks + 511) / 512);
    const double t_persist = rounds * (n * 2.0 * (blocks > 256 ? 2.0 : 1.0) / 2.2e10) + 6e-6;
    const double t_launch = n * c / 5e12 + 14e-6;
    return t_persist <= t_launch;
}

// Which persistent kernel: 1 = two chains per 512-thread block (lr_persist_kernel), 2 = four chains per 1024-thread
// block in ping-pong (lr_persist4_kernel).  The four-chain block hides the chain step under the other pair's scan but
// scans with 14 of its 16 waves, so it wins only while a step is a sizeable part of a scan (measured on cfg4-like
// data: ahead for 25k..1M lineages since its waves got unequal shares, behind below), and it fills the chip in rounds of 1024 chains where the
// two-chain kernel's remainder round is cheaper when at most 512 chains are left (C = 1536: 29.4 vs 33.5 us).
// Model in units of one full round (15.2 us on cfg4): t4 = ceil(C/1024), t2 = 1.08 floor(C/1024) + (0.82 | 1.08 for
// the remainder).
static int lr_persist_variant(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    if (!lr_persist_eligible(cfg, p)) return 0;
    static const int p4_env = lr_env_int("LR_PERSIST4", -1);
    if (p.cb < 4) return 1;                 // tables are laid out per group of cb chains; a quad must not straddle
    if (cfg->engine_mode == 3) return 2;
    if (p4_env >= 0) return p4_env ? 2 : 1;
    const int C = cfg->n_chains, rem = C % 1024;
    const double t4 = (double)((C + 1023) / 1024);
    const double t2 = 1.08 * (C / 1024) + (rem == 0 ? 0.0 : (rem <= 512 ? 0.82 : 1.08));
    return (cfg->n_lineages >= 25000 && cfg->n_lineages <= 1000000 && t4 < t2) ? 2 : 1;
}

static int lr_check_cfg(const lr_mcmc_config* cfg) {
    if (!cfg) return LR_ERR_NULL;
    if (cfg->n_lineages < 1 || cfg->n_chains < 1 || cfg->s_freq < 1 || cfg->n_trace_slots < 0) return LR_ERR_SIZE;
    if (cfg->n_bins < 1 || cfg->n_bins > LR_MAX_BINS) return LR_ERR_SIZE;
    if (cfg->model < 0 || cfg->model > 3) return LR_ERR_MODEL;
    if (cfg->t0 != floor(cfg->t0)) return LR_ERR_T0;
    if (!(cfg->end_time > cfg->start_time)) return LR_ERR_SIZE;
    if (cfg->sampler < 0 || cfg->sampler > 2) return LR_ERR_MODEL;
    if (cfg->sampler != 0) {
        if (cfg->model != LR_MODEL_KEIDING) return LR_ERR_MODEL;
        if (cfg->n_bins > LR_DD_MAXP * LR_WAVE) return LR_ERR_SIZE;
    }
    if (cfg->sampler == 1) {
        if (cfg->m_birth < 0 || cfg->m_birth > 2 || cfg->m_death < -2 || cfg->m_death > 2) return LR_ERR_MODEL;
        if (!(cfg->dd_present > cfg->t0)) return LR_ERR_SIZE;
    }
    return LR_OK;
}

extern "C" int lr_mcmc_query_layout(const lr_mcmc_config* cfg, lr_mcmc_layout* out) {
    int rc = lr_check_cfg(cfg);
    if (rc) return rc;
    if (!out) return LR_ERR_NULL;
    lr_scan_plan p;
    rc = lr_plan_engine(cfg, &p);
    if (rc) return rc;
    const long long C = cfg->n_chains;
    long long o = 0;
    out->state_f64 = o, o += lr_align_up64(C * LR_STATE_ROWS * LR_ROW * 8, 256);
    out->state_i32 = o, o += lr_align_up64(C * LR_ISTATE_ROWS * LR_ROW * 4, 256);
    out->bin_consts = o, o += lr_align_up64((long long)(cfg->n_bins + 2) * 8, 256);   // log(br) + the DD constants
    out->lineage_idx = o, o += lr_align_up64((lr_align_up64(cfg->n_lineages, 8) / 8 + LR_IDX_SPARE) * 16, 256);
    out->args_blob = o, o += 1024;   // lr_step_args of the persistent kernel
    out->tables = o, o += lr_align_up64((long long)lr_align_up64(C, p.cb < 2 ? 2 : p.cb) * p.tab_stride * 16, 256);
    out->partials = o, o += lr_align_up64((long long)p.tiles * C * 8, 256);
    out->trace = o, o += lr_align_up64((long long)cfg->n_trace_slots * C * LR_TRACE_W * 8, 256);
    out->total_bytes = o;
    out->table_stride = p.tab_stride;
    out->tiles = p.tiles;
    out->chains_per_block = p.cb;
    out->trace_width = LR_TRACE_W;
    {
        int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
        bool pipelined[LR_MAX_PARTS];
        out->n_parts = lr_partition(cfg->n_chains, p.cb, lr_fused_supported(p), base, hA, pipelined);
        out->pipelined = pipelined[0] ? 1 : 0;
    }
    out->persistent = lr_persist_variant(cfg, p);
    {
        static const int wide_env = lr_env_int("LR_PERSIST_WIDE", -1);
        const bool wide = wide_env >= 0 ? wide_env != 0 : ((cfg->n_chains + 1) / 2 <= 256 && cfg->n_lineages >= 20000);
        out->reserved1 = out->persistent == 2 ? 1024 : (out->persistent == 1 ? (wide ? 1024 : 512) : 0);   // threads per persistent block
    }
    return LR_OK;
}

extern "C" int lr_mcmc_create(const lr_mcmc_config* cfg, const double* ts, const double* te, const double* br_length,
                              void* workspace, int64_t workspace_bytes, lr_engine** out) {
    if (!ts || !te || !workspace || !out) return LR_ERR_NULL;
    lr_mcmc_layout lay;
    int rc = lr_mcmc_query_layout(cfg, &lay);
    if (rc) return rc;
    if ((cfg->model == LR_MODEL_BD || cfg->model == LR_MODEL_ID || cfg->sampler != 0) && !br_length) return LR_ERR_MODEL;
    if (lay.total_bytes > workspace_bytes) return LR_ERR_WORKSPACE;
    lr_engine* e = new (std::nothrow) lr_engine();
    if (!e) return (int)hipErrorOutOfMemory;
    e->cfg = *cfg;
    e->lay = lay;
    lr_plan_engine(cfg, &e->plan);
    e->ts = ts, e->te = te, e->br_length = br_length;
    e->ws = (char*)workspace;
    e->initialised = false;
    int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
    bool pipelined[LR_MAX_PARTS];
    e->n_parts = lr_partition(cfg->n_chains, e->plan.cb, lr_fused_supported(e->plan), base, hA, pipelined);
    e->persistent = lr_persist_eligible(cfg, e->plan);
    e->n8 = lr_align_up64(cfg->n_lineages, 8) / 8;
    e->n8_alloc = e->n8 + LR_IDX_SPARE;
    {
        // Shares of the 14 scanner waves, tuned on cfg4 (14 trips per wave) with in-kernel stamps until the waves of a
        // phase finish together: per wave pair (2,3) (4,5) ... (14,15) the trips beyond / short of the equal share.
        // The SIMD arbiter serves its oldest wave first and SIMDs 0, 1 also host the stepper waves, hence the shape.
        // Kept as fractions of the trip count for other inputs; short scans (bound by the chain step) stay equal.
        static const char* env = getenv("LR_P4_SHARES");       // "d2,d4,d6,d8,d10,d12,d14" for 14 trips
        static const int env2 = lr_env_int("LR_P2_SHARE", 0);   // two-chain kernel: trips per 24 moved from waves 4..7 to 0..3 (no gain measured: off)
        for (int j = 0; j < 16; ++j) e->p4.delta[j] = 0;
        if (e->lay.persistent == 2) {
            e->p4.n_slots = 14;
            int base[7] = {6, 6, 2, 0, -2, -6, -6};
            if (env) sscanf(env, "%d,%d,%d,%d,%d,%d,%d", &base[0], &base[1], &base[2], &base[3], &base[4], &base[5], &base[6]);
            const int k_tot = (int)((e->n8 + 895) / 896);
            int sum = 0;
            for (int j = 0; j < 7; ++j) {
                int d = (k_tot >= 6) ? (int)lrint((double)base[j] * k_tot / 14.0) : 0;
                if (d < -(k_tot - 1)) d = -(k_tot - 1);
                if (d > LR_P4_MAX_GIVE) d = LR_P4_MAX_GIVE;
                e->p4.delta[2 * j] = e->p4.delta[2 * j + 1] = d;
                sum += d;
            }
            // make the deltas sum to zero exactly: trim the largest takers / givers
            for (int guard = 0; sum != 0 && guard < 64; ++guard) {
                int pick = 0;
                for (int j = 1; j < 7; ++j)
                    if (sum > 0 ? e->p4.delta[2 * j] > e->p4.delta[2 * pick] : e->p4.delta[2 * j] < e->p4.delta[2 * pick]) pick = j;
                const int step = sum > 0 ? -1 : 1;
                e->p4.delta[2 * pick] += step, e->p4.delta[2 * pick + 1] += step;
                sum += step;
            }
            if (sum != 0) for (int j = 0; j < 14; ++j) e->p4.delta[j] = 0;
        } else {
            // two-chain kernel: 8 waves, two per SIMD; the younger four (4..7) trail the older four by ~10 % on long scans,
            // but with two unsynchronised blocks per CU moving trips between them bought nothing (16.3 us either way)
            const bool wide2 = e->lay.reserved1 == 1024;
            e->p4.n_slots = wide2 ? 16 : 8;
            const int k_tot = (int)((e->n8 + e->p4.n_slots * 64 - 1) / (e->p4.n_slots * 64));
            int d = (k_tot >= 12) ? (int)lrint((double)env2 * k_tot / 24.0) : 0;
            if (d > LR_P4_MAX_GIVE) d = LR_P4_MAX_GIVE;
            if (e->p4.n_slots == 8)
                for (int j = 0; j < 4; ++j) e->p4.delta[j] = d, e->p4.delta[4 + j] = -d;
            if (e->p4.n_slots == 16 && k_tot >= 6) {
                // wide variant: ONE block per CU, four scanner waves per SIMD - the oldest-first pattern of the four-chain
                // kernel: waves 0..3 / 4..7 take trips from 12..15 / 8..11 (per 12 trips)
                static const char* envw = getenv("LR_P2W_SHARES");
                int a = 5, b = 2;
                if (envw) sscanf(envw, "%d,%d", &a, &b);
                int da = (int)lrint((double)a * k_tot / 12.0), db = (int)lrint((double)b * k_tot / 12.0);
                if (da > LR_P4_MAX_GIVE) da = LR_P4_MAX_GIVE;
                if (db > LR_P4_MAX_GIVE) db = LR_P4_MAX_GIVE;
                if (da > k_tot - 1) da = k_tot - 1;
                if (db > k_tot - 1) db = k_tot - 1;
                for (int j = 0; j < 4; ++j) e->p4.delta[j] = da, e->p4.delta[4 + j] = db, e->p4.delta[8 + j] = -db, e->p4.delta[12 + j] = -da;
            }
        }
    }
    {
        // the takers' extra trips must stay inside the zero-filled spare behind the packed indices
        const long long stride = (long long)e->p4.n_slots * 64;
        const long long k_tot = (e->n8 + stride - 1) / stride;
        int dmax = 0;
        for (int j = 0; j < 16; ++j) dmax = e->p4.delta[j] > dmax ? e->p4.delta[j] : dmax;
        if ((k_tot + dmax + 1) * stride > e->n8_alloc)
            for (int j = 0; j < 16; ++j) e->p4.delta[j] = 0;
    }
    e->fork = nullptr;
    for (int p = 0; p < e->n_parts; ++p) {
        lr_part& q = e->part[p];
        q.base = base[p], q.count = base[p + 1] - base[p], q.hA = hA[p], q.pipelined = pipelined[p];
        q.stream = nullptr, q.done = nullptr, q.graph_exec = nullptr, q.graph_units = 0;
    }
    if (e->n_parts > 1) {
        hipError_t he = hipEventCreateWithFlags(&e->fork, hipEventDisableTiming);
        for (int p = 0; p < e->n_parts && he == hipSuccess; ++p) {
            he = hipStreamCreateWithFlags(&e->part[p].stream, hipStreamNonBlocking);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&e->part[p].done, hipEventDisableTiming);
        }
        if (he != hipSuccess) {
            lr_mcmc_destroy(e);
            return (int)he;
        }
    }
    *out = e;
    return LR_OK;
}

static lr_step_args lr_make_args(const lr_engine* e) {
    lr_step_args a;
    a.cfg = e->cfg;
    a.state_f64 = (double*)(e->ws + e->lay.state_f64);
    a.state_i32 = (int*)(e->ws + e->lay.state_i32);
    a.log_br = (const double*)(e->ws + e->lay.bin_consts);
    a.dd_consts = a.log_br + e->cfg.n_bins;
    a.log_T = log(e->cfg.end_time - e->cfg.start_time);
    a.mult_l = 2.0 * log(LR_MULT_D);
    a.tables = (double2*)(e->ws + e->lay.tables);
    a.partials = (const double*)(e->ws + e->lay.partials);
    a.trace = (double*)(e->ws + e->lay.trace);
    a.br_length = e->br_length;
    a.tab_stride = e->plan.tab_stride;
    a.n_cls = e->plan.n_cls;
    a.tiles = e->plan.tiles;
    a.H = e->plan.H;
    a.unit = e->plan.unit;
    a.cb = e->plan.cb;
    return a;
}

static int lr_enqueue_scan_range(const lr_engine* e, int base, int count, hipStream_t stream) {
    return lr_launch_scan(e->plan, e->ts, e->te, e->cfg.n_lineages, e->cfg.t0, e->cfg.n_bins, e->cfg.end_time,
                          (const double2*)(e->ws + e->lay.tables) + (size_t)base * e->plan.tab_stride, count,
                          (double*)(e->ws + e->lay.partials) + base, e->cfg.n_chains, stream);
}
static int lr_enqueue_scan(const lr_engine* e, hipStream_t stream) {
    return lr_enqueue_scan_range(e, 0, e->cfg.n_chains, stream);
}
static int lr_enqueue_step_range(const lr_engine* e, const lr_step_args& a, int mode, int base, int count,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(lr_chain_step_kernel, dim3((count + LR_STEP_WAVES - 1) / LR_STEP_WAVES), dim3(LR_SCAN_THREADS),
                       0, stream, a, mode, base, count);
    return (int)hipGetLastError();
}

// DD sampler: PRIOR_K0_L = np.max(DT) (DD:45) and its logarithm, behind the n_bins entries of bin_consts
__global__ void lr_dd_consts_kernel(const double* __restrict__ DT, int n_bins, double* __restrict__ out) {
    const int lane = threadIdx.x;
    double m = -INFINITY;
    for (int b = lane; b < n_bins; b += LR_WAVE) m = fmax(m, DT[b]);
    m = -lr_wave_min(-m);
    if (lane == 0) out[0] = m, out[1] = log(m);
}

// everything in the workspace that holds device addresses or derives from the data alone
static void lr_prepare_constants(const lr_engine* e, const lr_step_args& a, hipStream_t stream) {
    hipLaunchKernelGGL(lr_log_br_kernel, dim3((e->cfg.n_bins + 127) / 128), dim3(128), 0, stream, e->br_length,
                       e->cfg.n_bins, (double*)(e->ws + e->lay.bin_consts));
    if (e->cfg.sampler == 1)
        hipLaunchKernelGGL(lr_dd_consts_kernel, dim3(1), dim3(LR_WAVE), 0, stream, e->br_length, e->cfg.n_bins,
                           (double*)(e->ws + e->lay.bin_consts) + e->cfg.n_bins);
    if (e->persistent) {
        static_assert(sizeof(lr_step_args) <= 1024, "args blob too small");
        hipLaunchKernelGGL(lr_store_args_kernel, dim3(1), dim3(64), 0, stream, a, (lr_step_args*)(e->ws + e->lay.args_blob));
        // zero fill (padding entries (0, 0) = "outside the window" on both sides, contribution 0), then the lineages
        (void)hipMemsetAsync(e->ws + e->lay.lineage_idx, 0, (size_t)e->n8_alloc * 16, stream);
        hipLaunchKernelGGL(lr_pack_lineages_kernel, dim3((unsigned)((e->cfg.n_lineages + 255) / 256)), dim3(256), 0, stream,
                           e->ts, e->te, (long long)e->cfg.n_lineages, e->cfg.t0, e->cfg.n_bins,
                           1, (int)((e->n8 + e->p4.n_slots * 64 - 1) / (e->p4.n_slots * 64)), e->p4,
                           (unsigned short*)(e->ws + e->lay.lineage_idx));
    }
}

extern "C" int lr_mcmc_restore(lr_engine* e, void* stream_) {
    if (!e) return LR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const lr_step_args a = lr_make_args(e);
    lr_prepare_constants(e, a, stream);
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
    e->initialised = true;
    return LR_OK;
}

extern "C" int lr_mcmc_init(lr_engine* e, const double* L, const double* M, const double* tL, const double* tM,
                            const int32_t* KL, const int32_t* KM, int32_t kmax, void* stream_) {
    if (!e) return LR_ERR_NULL;
    if (L && e->cfg.sampler == 0 && (!M || !tL || !tM || !KL || !KM)) return LR_ERR_NULL;
    if (L && (kmax < 1 || kmax > LR_KMAX)) return LR_ERR_SIZE;
    if (L && e->cfg.sampler != 0 && kmax < LR_DD_NPAR) return LR_ERR_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    const lr_step_args a = lr_make_args(e);
    lr_prepare_constants(e, a, stream);
    hipLaunchKernelGGL(lr_chain_init_kernel, dim3(e->cfg.n_chains), dim3(LR_WAVE), 0, stream, a, L, M, tL, tM, KL, KM,
                       kmax);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    rc = lr_enqueue_scan(e, stream);
    if (rc) return rc;
    rc = lr_enqueue_step_range(e, a, 1, 0, e->cfg.n_chains, stream);
    if (rc) return rc;
    e->initialised = true;
    return LR_OK;
}

// iterations captured per hipGraph; LR_GRAPH_ITERS=0 in the environment disables graph replay
static int lr_graph_iters() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("LR_GRAPH_ITERS");
        v = e ? atoi(e) : 32;
        if (v < 0) v = 0;
        if (v > 4096) v = 4096;
    }
    return v;
}

// ---- software-pipelined schedule --------------------------------------------------------------
// Chains are split into halves A = [0, hA) and B = [hA, C).  A call of n iterations enqueues
//     scan(A);  (n-1) x { fused(scan B | step A); fused(scan A | step B) };  fused(scan B | step A);  step(B)
// so that every launch but the first and last overlaps the latency-bound chain step of one half with
// the throughput-bound lineage scan of the other.  LR_PIPELINE=0 falls back to scan; step; ...
template <int CB, int H, bool UNIT>
static int lr_launch_fused(const lr_engine* e, const lr_step_args& a, const lr_fused_args& f, hipStream_t stream) {
    const int groups = (f.scan_n + CB - 1) / CB;
    const int blocks = f.step_blocks + groups * f.tiles;
    const size_t lds_bytes = e->plan.lds_bytes > (int)(LR_STEP_WAVES * sizeof(lr_seg_scratch)) ? (size_t)e->plan.lds_bytes : LR_STEP_WAVES * sizeof(lr_seg_scratch);
    hipLaunchKernelGGL((lr_fused_iter_kernel<CB, H, UNIT>), dim3(blocks), dim3(LR_SCAN_THREADS), lds_bytes,
                       stream, a, f);
    return (int)hipGetLastError();
}

// instantiated shapes: see lr_fused_supported()
template <int H>
static int lr_launch_fused_h(const lr_engine* e, const lr_step_args& a, const lr_fused_args& f, hipStream_t stream) {
    if (e->plan.unit) {
        switch (e->plan.cb) {
            case 16: return lr_launch_fused<16, H, true>(e, a, f, stream);
            case 8: return lr_launch_fused<8, H, true>(e, a, f, stream);
            default: return LR_ERR_SIZE;
        }
    }
    switch (e->plan.cb) {
        case 8: return lr_launch_fused<8, H, false>(e, a, f, stream);
        case 4: return lr_launch_fused<4, H, false>(e, a, f, stream);
        default: return LR_ERR_SIZE;
    }
}

// one fused launch: scan chains [scan_base, +scan_n), step chains [step_base, +step_n)
static int lr_enqueue_fused(const lr_engine* e, const lr_step_args& a, int scan_base, int scan_n, int step_base,
                            int step_n, hipStream_t stream) {
    lr_fused_args f;
    f.ts = e->ts, f.te = e->te, f.n = e->cfg.n_lineages, f.chunk = e->plan.chunk, f.t0 = e->cfg.t0;
    f.n_bins = e->cfg.n_bins, f.tiles = e->plan.tiles;
    f.scan_base = scan_base, f.scan_n = scan_n, f.step_base = step_base, f.step_n = step_n;
    f.step_blocks = (step_n + LR_STEP_WAVES - 1) / LR_STEP_WAVES;
    switch (e->plan.H) {
        case 40: return lr_launch_fused_h<40>(e, a, f, stream);
        case 72: return lr_launch_fused_h<72>(e, a, f, stream);
        case 136: return lr_launch_fused_h<136>(e, a, f, stream);
        case 264: return lr_launch_fused_h<264>(e, a, f, stream);
        default: return LR_ERR_SIZE;
    }
}

// ---- schedule of one partition ------------------------------------------------------------------
// pipelined, n iterations:
//     scan(A);  (n-1) x { fused(scan B | step A); fused(scan A | step B) };  fused(scan B | step A);  step(B)
// every launch but the first and last overlaps the latency-bound chain step of one half with the
// throughput-bound lineage scan of the other.  Not pipelined: n x { scan; step }.
// The repeating unit {...} is captured once per partition in a hipGraph of LR_GRAPH_ITERS units.
static int lr_enqueue_unit(const lr_engine* e, const lr_step_args& a, const lr_part& q, hipStream_t stream) {
    if (!q.pipelined) {
        int rc = lr_enqueue_scan_range(e, q.base, q.count, stream);
        if (rc) return rc;
        return lr_enqueue_step_range(e, a, 0, q.base, q.count, stream);
    }
    int rc = lr_enqueue_fused(e, a, q.base + q.hA, q.count - q.hA, q.base, q.hA, stream);      // scan B | step A
    if (rc) return rc;
    return lr_enqueue_fused(e, a, q.base, q.hA, q.base + q.hA, q.count - q.hA, stream);       // scan A | step B
}

static int lr_run_units(lr_engine* e, const lr_step_args& a, lr_part& q, int64_t units, hipStream_t stream) {
    int64_t done = 0;
    const int G = lr_graph_iters();
    if (G > 0 && units >= G) {
        if (!q.graph_exec) {
            // capture G units once; the kernels read the iteration number from device memory, so the
            // same graph is valid for every replay
            hipStream_t cs = nullptr;
            hipGraph_t graph = nullptr;
            hipError_t he = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
            if (he != hipSuccess) return (int)he;
            int rc = LR_OK;
            he = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            if (he == hipSuccess) {
                for (int i = 0; i < G && rc == LR_OK; ++i) rc = lr_enqueue_unit(e, a, q, cs);
                he = hipStreamEndCapture(cs, &graph);       // always ended, so that the stream leaves capture mode
            }
            if (he == hipSuccess && rc == LR_OK) he = hipGraphInstantiate(&q.graph_exec, graph, nullptr, nullptr, 0);
            // the capture stream and the graph are released on every path
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipStreamDestroy(cs);
            if (rc) return rc;
            if (he != hipSuccess) return (int)he;
            q.graph_units = G;
        }
        while (units - done >= q.graph_units) {
            hipError_t he = hipGraphLaunch(q.graph_exec, stream);
            if (he != hipSuccess) return (int)he;
            done += q.graph_units;
        }
    }
    for (; done < units; ++done) {
        const int rc = lr_enqueue_unit(e, a, q, stream);
        if (rc) return rc;
    }
    return LR_OK;
}

static int lr_run_part(lr_engine* e, const lr_step_args& a, lr_part& q, int64_t n_iters, hipStream_t stream) {
    if (!q.pipelined) return lr_run_units(e, a, q, n_iters, stream);
    // prologue: scan A alone (the fused kernel with no step blocks, so that the stand-alone scan kernel is
    // launched only at full size - by lr_mcmc_init and the lr_mcmc_time_scan measurement hook)
    int rc = lr_enqueue_fused(e, a, q.base, q.hA, q.base, 0, stream);                          // scan A
    if (rc) return rc;
    rc = lr_run_units(e, a, q, n_iters - 1, stream);
    if (rc) return rc;
    rc = lr_enqueue_fused(e, a, q.base + q.hA, q.count - q.hA, q.base, q.hA, stream);          // scan B | step A
    if (rc) return rc;
    return lr_enqueue_step_range(e, a, 0, q.base + q.hA, q.count - q.hA, stream);              // step B
}

extern "C" int lr_mcmc_steps(lr_engine* e, int64_t n_iters, void* stream_) {
    if (!e) return LR_ERR_NULL;
    if (!e->initialised) return LR_ERR_STATE;
    if (n_iters < 0) return LR_ERR_SIZE;
    if (n_iters == 0) return LR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const lr_step_args a = lr_make_args(e);
    if (e->persistent) {
        const uint4* idx8 = (const uint4*)(e->ws + e->lay.lineage_idx);
        const lr_step_args* ap = (const lr_step_args*)(e->ws + e->lay.args_blob);
        const int blocks = (e->cfg.n_chains + 1) / 2;
        const bool p4 = e->lay.persistent == 2;
        static const int prio = lr_env_int("LR_PERSIST_PRIO", 12);   // clock bits per priority slice, 0 = off
        // one block per CU at most: give it the whole CU (16 waves on the one pair)
        static const int wide_env = lr_env_int("LR_PERSIST_WIDE", -1);
        const bool wide = e->lay.reserved1 == 1024;   // (short scans keep 512: the 16-wave barrier costs more than it buys)
        (void)wide_env;
        for (int64_t done = 0; done < n_iters;) {
            const int64_t n = (n_iters - done > 4096) ? 4096 : n_iters - done;   // keep single launches short
            switch (e->plan.H) {
                case 40: if (p4) hipLaunchKernelGGL(lr_persist4_kernel<40>, dim3((e->cfg.n_chains + 3) / 4), dim3(LR_P4_THREADS), 0, stream, ap, idx8, e->n8, e->p4, (long long)n); else { if (wide) hipLaunchKernelGGL((lr_persist_kernel<40, 1024>), dim3(blocks), dim3(1024), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, 0); else hipLaunchKernelGGL((lr_persist_kernel<40, 512>), dim3(blocks), dim3(512), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, prio); } break;
                case 72: if (p4) hipLaunchKernelGGL(lr_persist4_kernel<72>, dim3((e->cfg.n_chains + 3) / 4), dim3(LR_P4_THREADS), 0, stream, ap, idx8, e->n8, e->p4, (long long)n); else { if (wide) hipLaunchKernelGGL((lr_persist_kernel<72, 1024>), dim3(blocks), dim3(1024), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, 0); else hipLaunchKernelGGL((lr_persist_kernel<72, 512>), dim3(blocks), dim3(512), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, prio); } break;
                case 136: if (p4) hipLaunchKernelGGL(lr_persist4_kernel<136>, dim3((e->cfg.n_chains + 3) / 4), dim3(LR_P4_THREADS), 0, stream, ap, idx8, e->n8, e->p4, (long long)n); else { if (wide) hipLaunchKernelGGL((lr_persist_kernel<136, 1024>), dim3(blocks), dim3(1024), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, 0); else hipLaunchKernelGGL((lr_persist_kernel<136, 512>), dim3(blocks), dim3(512), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, prio); } break;
                default: if (p4) hipLaunchKernelGGL(lr_persist4_kernel<264>, dim3((e->cfg.n_chains + 3) / 4), dim3(LR_P4_THREADS), 0, stream, ap, idx8, e->n8, e->p4, (long long)n); else { if (wide) hipLaunchKernelGGL((lr_persist_kernel<264, 1024>), dim3(blocks), dim3(1024), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, 0); else hipLaunchKernelGGL((lr_persist_kernel<264, 512>), dim3(blocks), dim3(512), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, prio); } break;
            }
            const int rc = (int)hipGetLastError();
            if (rc) return rc;
            done += n;
        }
        return LR_OK;
    }
    if (e->n_parts == 1) return lr_run_part(e, a, e->part[0], n_iters, stream);
    // fork: every partition's stream waits for the caller's stream, runs its own sequence, and is joined back
    hipError_t he = hipEventRecord(e->fork, stream);
    if (he != hipSuccess) return (int)he;
    for (int p = 0; p < e->n_parts; ++p) {
        lr_part& q = e->part[p];
        he = hipStreamWaitEvent(q.stream, e->fork, 0);
        if (he != hipSuccess) return (int)he;
        const int rc = lr_run_part(e, a, q, n_iters, q.stream);
        if (rc) return rc;
        he = hipEventRecord(q.done, q.stream);
        if (he != hipSuccess) return (int)he;
        he = hipStreamWaitEvent(stream, q.done, 0);
        if (he != hipSuccess) return (int)he;
    }
    return LR_OK;
}

extern "C" int lr_mcmc_time_steps(lr_engine* e, int64_t n_iters, float* total_ms, void* stream_) {
    if (!e || !total_ms) return LR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    hipEvent_t t0, t1;
    hipError_t he = hipEventCreate(&t0);
    if (he != hipSuccess) return (int)he;
    he = hipEventCreate(&t1);
    if (he != hipSuccess) {
        (void)hipEventDestroy(t0);
        return (int)he;
    }
    int rc = (int)hipEventRecord(t0, stream);
    if (rc == LR_OK) rc = lr_mcmc_steps(e, n_iters, stream_);
    if (rc == LR_OK) rc = (int)hipEventRecord(t1, stream);
    if (rc == LR_OK) rc = (int)hipEventSynchronize(t1);
    float ms = 0.f;
    if (rc == LR_OK) rc = (int)hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    *total_ms = ms;
    return rc;
}

extern "C" int lr_mcmc_time_scan(lr_engine* e, int32_t reps, float* avg_ms, void* stream_) {
    if (!e || !avg_ms) return LR_ERR_NULL;
    if (!e->initialised) return LR_ERR_STATE;
    if (reps < 1) return LR_ERR_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    hipEvent_t t0, t1;
    hipError_t he = hipEventCreate(&t0);
    if (he != hipSuccess) return (int)he;
    he = hipEventCreate(&t1);
    if (he != hipSuccess) {
        (void)hipEventDestroy(t0);
        return (int)he;
    }
    int rc = lr_enqueue_scan(e, stream);  // warm
    if (rc == LR_OK) rc = (int)hipEventRecord(t0, stream);
    for (int i = 0; i < reps && rc == LR_OK; ++i) rc = lr_enqueue_scan(e, stream);
    if (rc == LR_OK) rc = (int)hipEventRecord(t1, stream);
    if (rc == LR_OK) rc = (int)hipEventSynchronize(t1);
    float ms = 0.f;
    if (rc == LR_OK) rc = (int)hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    *avg_ms = ms / reps;
    return rc;
}

// name of the kernel lr_mcmc_steps spends its time in, as rocprofv3's kernel trace prints it (without arguments)
extern "C" int lr_mcmc_describe(const lr_engine* e, char* buf, int32_t n) {
    if (!e || !buf) return LR_ERR_NULL;
    if (n < 64) return LR_ERR_SIZE;
    if (e->persistent) {
        if (e->lay.persistent == 2) snprintf(buf, (size_t)n, "lr_persist4_kernel<%d>", e->plan.H);
        else snprintf(buf, (size_t)n, "lr_persist_kernel<%d, %d>", e->plan.H, e->lay.reserved1);
    } else if (e->part[0].pipelined) {
        snprintf(buf, (size_t)n, "lr_fused_iter_kernel<%d, %d, %s>", e->plan.cb, e->plan.H, e->plan.unit ? "true" : "false");
    } else if (e->plan.fast) {
        snprintf(buf, (size_t)n, "%s<%d, %d>", e->plan.unit ? "lr_scan_unit_kernel" : "lr_scan_fast_kernel", e->plan.cb, e->plan.H);
    } else {
        snprintf(buf, (size_t)n, "lr_scan_kernel<%d>", e->plan.cb);
    }
    return LR_OK;
}

extern "C" int lr_mcmc_destroy(lr_engine* e) {
    if (!e) return LR_ERR_NULL;
    for (int p = 0; p < e->n_parts; ++p) {
        lr_part& q = e->part[p];
        if (q.graph_exec) (void)hipGraphExecDestroy(q.graph_exec);
        if (q.done) (void)hipEventDestroy(q.done);
        if (q.stream) (void)hipStreamDestroy(q.stream);
    }
    if (e->fork) (void)hipEventDestroy(e->fork);
    delete e;
    return LR_OK;
}

#ifdef LR_DIAG
extern "C" int lr_diag_dump_step(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_step), (size_t)n_words * 8);
}
#endif
