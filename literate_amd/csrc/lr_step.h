// lr_step.h - the chain step of the RJMCMC engines (runMCMC, LiteRateForward.py:216-373; the DDRate.py and
// trend_rate.py loops) as wave-level device functions: one wave owns one chain, lane j holds element j of every
// state vector.  Split in two halves so that an engine may run them apart:
//   lr_decide_*  - Metropolis-Hastings accept of the pending proposal given its scanned log-likelihood, trace row;
//   lr_propose_* - next proposal from a given state, its prior, its lookup tables.
// lr_chain_step_core / lr_dd_step_core chain the two (launch-based engine, two- and four-chain persistent kernels);
// the speculative engine (lr_spec.h) runs lr_propose_* on BOTH possible outcomes while the scan is still running.
#pragma once
#include "lr_chain.h"
#include "lr_dd.h"
#include "lr_internal.h"
#include "lr_scan.h"

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
    unsigned int* warn;     // engine warning word (LR_WARN_*), beside the status word
};

// where chain c's lookup table starts.  General layout: chain-major, tab_stride double2 per chain.
// Unit-resolution layout: groups of cb chains, inside a group [pair][2H] double2 = (even chain, odd chain);
// the returned pointer addresses this chain's component, consecutive entries are 2 doubles apart.
__device__ __forceinline__ double2* lr_chain_table(const lr_step_args& a, int c) {
    if (!a.unit) return a.tables + (size_t)c * a.tab_stride;
    if (a.unit == LR_TAB_PAIRGEN)   // pair-major: 8H doubles per pair (value plane, slope plane), this chain's at component c & 1
        return reinterpret_cast<double2*>(reinterpret_cast<double*>(a.tables) + (size_t)(c >> 1) * (8 * a.H) + (c & 1));
    const int l = c % a.cb;
    double* base = reinterpret_cast<double*>(a.tables + (size_t)(c - l) * a.tab_stride);
    return reinterpret_cast<double2*>(base + (size_t)(l >> 1) * (4 * a.H) + (l & 1));
}

#ifdef LR_DIAG
static __device__ unsigned long long lr_diag_step[4096 * 12];
// last-iteration stamps of the even waves (s_memtime, shader clock): plain stores, averaged over chains by the reader
static __device__ unsigned long long lr_diag_seg[64 * 16];
#define LR_SSTAMP(k) if (lane == 0 && c < 64 && ((threadIdx.x >> 6) & 1) == 0) lr_diag_seg[c * 16 + (k)] = clock64()
#else
#define LR_SSTAMP(k)
#endif

// Ordering point for LDS traffic INSIDE one wave (lane A writes, lane B of the same wave reads).  The LDS executes a
// wave's instructions in issue order, so no counter has to drain; what is needed is that the compiler keeps the
// program order of the accesses.  (A workgroup-scope fence here costs an s_waitcnt vmcnt(0) lgkmcnt(0): it also waits
// for every global / scratch access the wave has in flight - about a fifth of a microsecond each in the chain step.)
#define LR_WAVE_LDS_ORDER()                 \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_wave_barrier();    \
        asm volatile("" ::: "memory");      \
    } while (0)

// Table layout a builder writes.  The persistent kernels (LDS_CONSTS) only ever run on the packed layouts and know the
// entry stride at compile time, so the layout switches of lr_put_S / lr_put_E fold away in them.
template <bool LDS_CONSTS>
__device__ __forceinline__ int lr_tab_mode(const lr_step_args& a, int table_es) {
    return LDS_CONSTS ? (table_es != 2 ? LR_TAB_PAIRGEN : LR_TAB_UNIT) : a.unit;
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
                                                       int unit = LR_TAB_GENERAL, double fs0 = 0.0, double fe0 = 0.0,
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
        lr_put_S(tabd, unit, es, b + 1, logB + cum, R, fs0);
        lr_put_E(tabd, unit, es, H + b + 1, logD - cum, R, fe0);
        cum += R;
        if (n_cls == 2) {
            tab[2 * H + b + 1] = make_double2(logB + cuml, lam);
            tab[3 * H + b + 1] = make_double2(-cuml, -lam);
            cuml += lam;
        }
    }
    if (lane == 0) {
        lr_put_S(tabd, unit, es, 0, 0.0, 0.0, fs0), lr_put_E(tabd, unit, es, H, 0.0, 0.0, fe0);
        lr_put_S(tabd, unit, es, n_bins + 1, totR, 0.0, fs0), lr_put_E(tabd, unit, es, H + n_bins + 1, -totR, 0.0, fe0);
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
// ranks of a lane's bins (8 bits each) as the last build found them: reusable while the bin edges stay what they were
struct lr_seg_cache {
    int packL, packM;
    bool reuse;
};

// Segment of every bin = number of shifts at or before it, for the P bins of this lane (b0 = lane * P).  The K - 1 shift
// lanes drop a count on their bin in LDS (birth shifts in the low half-word, death shifts in the high one), every lane
// reads the counts of its own bins and one integer wave scan turns them into ranks: a fixed ~35 instructions instead of
// a dependent (K_l + K_m) x P compare-and-add chain (the largest single item of the chain step before).
template <int P>
__device__ __forceinline__ void lr_bin_ranks(const lr_seg_scratch* sc, int eL, int eM, int KL, int KM, int n_bins, int lane,
                                             int (&segL)[P], int (&segM)[P]) {
    const int b0 = lane * P;
    int* marks = const_cast<int*>(sc->marks);
    for (int b = lane; b <= n_bins; b += LR_WAVE) marks[b] = 0;
    LR_WAVE_LDS_ORDER();
    if (lane >= 1 && lane < KL) atomicAdd(&marks[eL], 1);
    if (lane >= 1 && lane < KM) atomicAdd(&marks[eM], 0x10000);
    LR_WAVE_LDS_ORDER();
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
// the ranks of a lane's bins, 8 bits each (at most LR_KMAX = 32 segments, at most 4 bins per lane)
template <int P>
__device__ __forceinline__ void lr_pack_ranks(const int (&segL)[P], const int (&segM)[P], int* packL, int* packM) {
    int pl = 0, pm = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) pl |= segL[p] << (8 * p), pm |= segM[p] << (8 * p);
    *packL = pl, *packM = pm;
}

// CS: doubles between consecutive entries of the chain's column (lr_put_S)
template <int P, int CS = 2>
__device__ __forceinline__ double lr_build_tables_segments_fast(const lr_seg_scratch* sc, int eL, int eM, int KL,
                                                                int KM, const double* __restrict__ br_length,
                                                                const double* __restrict__ log_br, int model,
                                                                int n_bins, int H, double2* __restrict__ tab,
                                                                int lane, int unit, double fs0, double fe0, int es,
                                                                lr_seg_cache* sg = nullptr) {
    double* tabd = reinterpret_cast<double*>(tab);
    auto put_S = [&](int j, double v, double R) { lr_put_S<CS>(tabd, unit, es, j, v, R, fs0); };
    auto put_E = [&](int j, double v, double R) { lr_put_E<CS>(tabd, unit, es, j, v, R, fe0); };
    const int b0 = lane * P;
#ifdef LR_DIAG
    const int c = blockIdx.x * 2 + ((threadIdx.x >> 7) & 1);
#endif
    LR_SSTAMP(9);
    int segL[P], segM[P];
    double k_b[P], lk_b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        segL[p] = 0, segM[p] = 0;
        const int b = min(b0 + p, n_bins - 1);
        k_b[p] = (model < 2) ? br_length[b] : 1.0;
        lk_b[p] = (model < 2) ? log_br[b] : 0.0;
    }
    // segment of every bin (lr_bin_ranks)
    if (sg && sg->reuse) {
        // same bin edges as the state the proposal was made from: its ranks (8 bits per bin of this lane) still hold
#pragma unroll
        for (int p = 0; p < P; ++p) segL[p] = (sg->packL >> (8 * p)) & 0xff, segM[p] = (sg->packM >> (8 * p)) & 0xff;
    } else {
        lr_bin_ranks<P>(sc, eL, eM, KL, KM, n_bins, lane, segL, segM);
        if (sg) lr_pack_ranks<P>(segL, segM, &sg->packL, &sg->packM);
    }
    LR_SSTAMP(10);
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
    LR_SSTAMP(11);
    double totR;
    double cum = lr_wave_exclusive_scan(sumR, lane, &totR);
    LR_SSTAMP(12);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int b = b0 + p;
        if (b < n_bins) {
            put_S(b + 1, logB[p] + cum, R[p]);
            put_E(H + b + 1, logD[p] - cum, R[p]);
        }
        cum += R[p];
    }
    if (lane == 0) {
        put_S(0, 0.0, 0.0), put_E(H, 0.0, 0.0);
        put_S(n_bins + 1, totR, 0.0), put_E(H + n_bins + 1, -totR, 0.0);
    }
    if (model == LR_MODEL_KEIDING_DEAD && unit != LR_TAB_GENERAL) {
        // Model 3 (LRF:141-142, 529-546) in the packed layouts: an EXTANT lineage (te >= end_time) contributes to the
        // birth process only: log lam at its birth + the exposure to lam from there on.  Relative to the birth entry
        // S[a] it already gathers, that is  -tot(lam) - cum(mu)_a - fs mu_a,  a function of its birth bin alone: a second
        // block of n_bins + 2 death-side entries E_ext[a], which the packing points extant lineages at (death byte
        // n_bins + 2 + a; on general times their fe' slot carries fs).  No table class in the scan loop.
        double mu_b[P], sumM = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            mu_b[p] = (b0 + p < n_bins) ? sc->rate[1][segM[p]] : 0.0;
            sumM += mu_b[p];
        }
        double totM;
        double cumM = lr_wave_exclusive_scan(sumM, lane, &totM);
        const double totL = totR - totM;
        const int x0 = H + n_bins + 2;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int b = b0 + p;
            if (b < n_bins) put_S(x0 + b + 1, -totL - cumM, -mu_b[p]);
            cumM += mu_b[p];
        }
        if (lane == 0) put_S(x0, -totL, 0.0), put_S(x0 + n_bins + 1, -totR, 0.0);
    }
    LR_SSTAMP(13);
    return (model == 1) ? lr_wave_sum(csum) : 0.0;
}

// dispatcher: fast one-pass builder when the shape allows, general two-pass builder otherwise; PB > 0: the caller is
// instantiated for a table size whose bins-per-lane count is PB (lr_bins_per_lane) and has one table class
template <int PB = 0, int CS = 2>
__device__ __forceinline__ double lr_build_tables_segments(const lr_seg_scratch* sc, int eL, int eM, int KL, int KM,
                                                           const double* __restrict__ br_length,
                                                           const double* __restrict__ log_br, int model, int n_bins,
                                                           int n_cls, int H, double2* __restrict__ tab, int lane,
                                                           int unit, double fs0, double fe0, int es = 2,
                                                           lr_seg_cache* sg = nullptr) {
    static_assert(CS == 2 || PB > 0, "a column of its own is written by the one-pass builder only");
    if (PB > 0)
        return lr_build_tables_segments_fast<(PB > 0 ? PB : 1), CS>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab,
                                                                    lane, unit, fs0, fe0, es, sg);
    // (run-time dispatch: the bins-per-lane count of the table size, as the kernels instantiated for H use it - every
    // builder of an engine then produces the same doubles for the same state, whichever kernel runs it)
    if (n_cls == 1 && H <= 264) {
        switch (lr_bins_per_lane(H)) {
            case 1: return lr_build_tables_segments_fast<1>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab, lane, unit, fs0, fe0, es, sg);
            case 2: return lr_build_tables_segments_fast<2>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab, lane, unit, fs0, fe0, es, sg);
            default: return lr_build_tables_segments_fast<4>(sc, eL, eM, KL, KM, br_length, log_br, model, n_bins, H, tab, lane, unit, fs0, fe0, es, sg);
        }
    }
    if (sg) sg->reuse = false, sg->packL = sg->packM = -1;
    return lr_build_tables_segments_wave(sc, KL, KM, br_length, log_br, model, n_bins, n_cls, H, tab, lane, unit, fs0,
                                         fe0, es);
}

// stage one chain's segments in the wave's LDS scratch; log of all rates in ONE call
// (lanes 0..31 carry the birth rates, lanes 32..63 the death rates)
__device__ __forceinline__ void lr_stage_segments(lr_seg_scratch* sc, double L, double M, int eL, int eM, int KL,
                                                  int KM, int lane, double* logL, double* logM, double extra = 1.0,
                                                  double* log_extra = nullptr) {
    const double Mhi = lr_half_lo(M);                   // lane l <- M[l & 31]
    const bool hi = lane >= 32;
    const int j = lane & 31;
    const bool valid = hi ? (j < KM) : (j < KL);
    // lane 63 is free unless the death process holds LR_KMAX rates: it takes one more logarithm along (`extra`)
    const bool free63 = KM < LR_KMAX;
    double x = valid ? (hi ? Mhi : L) : 1.0;
    if (lane == LR_WAVE - 1 && free63) x = extra;
    const double lx = lr_log(x);
    if (log_extra) *log_extra = free63 ? lr_bcast(lx, LR_WAVE - 1) : lr_log(extra);
    sc->rate[hi][j] = x;
    sc->lograte[hi][j] = lx;
    if (lane <= LR_KMAX) sc->edge[0][lane] = eL, sc->edge[1][lane] = eM;
    *logL = lx;                                        // valid on lanes < 32
    *logM = lr_half_hi(lx);                            // lane j gets log M[j]
    // the scratch is private to this wave: only the compiler must be kept from moving the reads of the table builders
    // above these writes
    LR_WAVE_LDS_ORDER();
}


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

// the same for a step whose proposal's tables are a helper wave's: that wave writes the model constant of the proposal
// into the state's scalar row itself, at a time of its own - every scalar but that one
__device__ __forceinline__ void lr_chain_store_handed(const lr_chain_regs& r, double* S, int* I, int lane) {
    S[LR_ROW_L * LR_ROW + lane] = r.L, S[LR_ROW_M * LR_ROW + lane] = r.M;
    S[LR_ROW_TL * LR_ROW + lane] = r.tL, S[LR_ROW_TM * LR_ROW + lane] = r.tM;
    S[LR_ROW_PL * LR_ROW + lane] = r.pL, S[LR_ROW_PM * LR_ROW + lane] = r.pM;
    S[LR_ROW_PTL * LR_ROW + lane] = r.ptL, S[LR_ROW_PTM * LR_ROW + lane] = r.ptM;
    if (lane != LR_S_CONST_P) S[LR_ROW_SCALARS * LR_ROW + lane] = r.sc;
    I[LR_IROW_EL * LR_ROW + lane] = r.eL, I[LR_IROW_EM * LR_ROW + lane] = r.eM;
    I[LR_IROW_PEL * LR_ROW + lane] = r.peL, I[LR_IROW_PEM * LR_ROW + lane] = r.peM;
    I[LR_IROW_SCALARS * LR_ROW + lane] = r.isc;
}

// ---- the two halves of a runMCMC step (LRF:216-373) ---------------------------------------------------------
// A state "as accepted" in the registers of one wave: per-lane rows plus wave-uniform scalars.
struct lr_rj_state {
    double L, M, tL, tM;                  // lane j: rate j / shift time j of the birth and death process
    int eL, eM;                           // integer bin edges (relative to bin 0)
    int KL, KM;                           // number of rates per process
    double g0, g1, poi, lg0, lg1, lpoi;   // hyper-parameters Gamma_rate[0..1], Poi_lambda_rjHP and their logs (LRF:220-222)
    double priorPoi;                      // the Poisson prior term cached with this state (LRF:300-304: stale after Gibbs)
    int sgL, sgM, sg_valid;               // speculative engine: the table builder's bin ranks for this state's edges (per lane)
};
// what a proposal carries beside its state
struct lr_rj_prop {
    double hasting, prior, constP;        // Hastings / log q + log J term, log prior, model constant of its likelihood
    double log_u;                         // log of the acceptance uniform of the iteration that decides it
    int gibbs, invalid, move;             // Gibbs step (always accepted); fails the LRF:290 guard / the K cap; LR_I_MOVE kind
    int table_by_helper;                  // the column, the model constant and the rank cache come from a helper wave
};

// The Metropolis-Hastings rule of LRF:305-313 for a pending proposal whose lineage scan returned lik_sum.
// values of a proposal's `invalid` flag (any non-zero value rejects it, LRF:290-292)
#define LR_INVALID_GUARD 1   /* fails the LRF:290 guard                                                 */
#define LR_INVALID_KCAP 3    /* an add-shift from K = LR_KMAX rates: the device cap, not the reference's */
// called where a proposal is decided: the chain really proposed an add-shift at the cap and loses it for that reason
__device__ __forceinline__ void lr_warn_kcap(unsigned int* warn, int invalid, int lane) {
    if (invalid == LR_INVALID_KCAP && lane == 0)
        __hip_atomic_fetch_or(warn, (unsigned int)LR_WARN_KCAP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool lr_mh_accept(int gibbs, int invalid, double lik_sum, double constP, double likA,
                                             double priorP, double priorA, double hasting, double log_u, double* lik) {
    *lik = gibbs ? likA : lik_sum + constP;
    return gibbs || (!invalid && (*lik - likA + priorP - priorA + hasting >= log_u));
}

// One trace row of chain c (LRF:321-359) from its accepted state.
__device__ __forceinline__ void lr_write_trace_row(const lr_step_args& a, int c, int lane, int slot, uint64_t it,
                                                   double likA, double priorA, const lr_rj_state& s) {
    const lr_mcmc_config& cfg = a.cfg;
    double* row = a.trace + ((size_t)slot * cfg.n_chains + c) * LR_TRACE_W;
    const double meanL = lr_wave_sum(lane < s.KL ? s.L : 0.0) / s.KL;
    const double meanM = lr_wave_sum(lane < s.KM ? s.M : 0.0) / s.KM;
    double h = 0.0;
    switch (lane) {
        case 0: h = (double)it; break;
        case 1: h = likA + priorA; break;
        case 2: h = likA; break;
        case 3: h = priorA; break;
        case 4: h = meanL; break;
        case 5: h = meanM; break;
        case 6: h = s.KL; break;
        case 7: h = s.KM; break;
        case 8: h = cfg.start_time; break;
        case 9: h = cfg.end_time; break;
        case 10: h = s.g0; break;
        case 11: h = s.g1; break;
        case 12: h = s.poi; break;
    }
    if (lane < LR_TRACE_HEAD) row[lane] = h;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double* rl = row + LR_TRACE_HEAD;
    double* rm = rl + (2 * LR_KMAX - 1);
    if (lane < LR_KMAX) rl[lane] = lane < s.KL ? s.L : nan, rm[lane] = lane < s.KM ? s.M : nan;
    if (lane >= 1 && lane < LR_KMAX) {
        rl[LR_KMAX + lane - 1] = lane < s.KL ? s.tL : nan;
        rm[LR_KMAX + lane - 1] = lane < s.KM ? s.tM : nan;
    }
}

// The draws of one iteration that do not depend on the chain's state: everything lr_propose_rj takes from the Philox
// stream except the rare Gibbs variates (their shapes depend on the number of rates).  The speculative engine has them
// made one iteration ahead by a wave with time to spare; the values are the ones the inline code below draws.
struct lr_rj_draws {
    double log_u;                            // log of the acceptance uniform of the iteration
    double r_a, r_b, q_a, q_b, q2_a, q2_b;   // move selector, RJ side / kind, RJ position
    double beta;                             // Beta(10, 10) variate of an add-shift move (0 when the iteration is none)
    double x, m, da;                         // per lane: multiplier exponent 2 log d (u - .5), its exp, the mask uniform
};

__device__ __forceinline__ void lr_make_rj_draws(const lr_step_args& a, int c, int lane, uint64_t it, lr_rj_draws& d) {
    const lr_mcmc_config& cfg = a.cfg;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const uint32_t purpose = (lane == 0) ? LR_P_ACCEPT : (lane == 1 ? LR_P_MOVE : LR_P_RJ);
    const lr_u2 ud = lr_pair(rng, it, purpose, lane == 3 ? 1u : 0u);
    const double lu = lr_log(lane == 0 ? ud.a : 1.0);
    d.log_u = lr_bcast(lu, 0);
    d.r_a = lr_bcast(ud.a, 1), d.r_b = lr_bcast(ud.b, 1);
    d.q_a = lr_bcast(ud.a, 2), d.q_b = lr_bcast(ud.b, 2);
    d.q2_a = lr_bcast(ud.a, 3), d.q2_b = lr_bcast(ud.b, 3);
    d.x = 0.0, d.m = 1.0, d.da = 1.0, d.beta = 0.0;
    if (d.r_a < 0.8) {
        // a multiplier move of one of the two processes (or the no-op times move): LRF:165-176
        const lr_u2 u = lr_pair(rng, it, LR_P_MULT, lane);
        d.da = u.a;
        d.x = a.mult_l * (u.b - .5);
        d.m = exp(d.x);
    } else if (d.r_a < 0.999 && cfg.const_rates == 0 && d.q_b > 0.5) {
        double ga, gb;
        lr_wave_gamma2(rng, it, LR_P_BETA_A, LR_SHAPE_BETA_RJ, LR_P_BETA_B, LR_SHAPE_BETA_RJ, lane, &ga, &gb);
        d.beta = ga / (ga + gb);
    }
}

// the state-independent draws of one iteration of one chain (lr_rj_draws), made one iteration ahead
struct lr_draw_slot {
    double sc[8];                 // log_u, r_a, r_b, q_a, q_b, q2_a, q2_b, beta
    double x[LR_ROW], m[LR_ROW], da[LR_ROW];
};

__device__ __forceinline__ void lr_draws_store(lr_draw_slot* q, const lr_rj_draws& d, int lane) {
    double so = d.log_u;
    so = (lane == 1) ? d.r_a : so;
    so = (lane == 2) ? d.r_b : so;
    so = (lane == 3) ? d.q_a : so;
    so = (lane == 4) ? d.q_b : so;
    so = (lane == 5) ? d.q2_a : so;
    so = (lane == 6) ? d.q2_b : so;
    so = (lane == 7) ? d.beta : so;
    if (lane < 8) q->sc[lane] = so;
    q->x[lane] = d.x, q->m[lane] = d.m, q->da[lane] = d.da;
}

__device__ __forceinline__ void lr_draws_load(const lr_draw_slot* q, lr_rj_draws& d, int lane) {
    const double v = q->sc[lane & 7];          // one LDS read for the eight scalars
    d.log_u = lr_bcast(v, 0), d.r_a = lr_bcast(v, 1), d.r_b = lr_bcast(v, 2), d.q_a = lr_bcast(v, 3);
    d.q_b = lr_bcast(v, 4), d.q2_a = lr_bcast(v, 5), d.q2_b = lr_bcast(v, 6), d.beta = lr_bcast(v, 7);
    d.x = q->x[lane], d.m = q->m[lane], d.da = q->da[lane];
}

// draw duty of a scanner wave: the draws of iteration `it` of chain c into `out`
__device__ __forceinline__ void lr_spec_draw(const lr_step_args& a, int c, int lane, unsigned long long it,
                                             lr_draw_slot* out) {
    lr_rj_draws d;
    lr_make_rj_draws(a, c, lane, it, d);
    lr_draws_store(out, d, lane);
}

// The duty split over two waves (the two halves are independent Philox blocks, each a long dependency chain):
// part 0 the wave-uniform draws (acceptance uniform and its log, move selectors, the RJ pairs, the split's beta),
// part 1 the per-rate multiplier draws.  Part 1 runs whether or not the move turns out to be a multiplier move: the
// proposal reads x / m / da in multiplier moves only.
__device__ __forceinline__ void lr_spec_draw_part(const lr_step_args& a, int c, int lane, unsigned long long it,
                                                  lr_draw_slot* out, int part) {
    const lr_mcmc_config& cfg = a.cfg;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    if (part) {
        const lr_u2 u = lr_pair(rng, it, LR_P_MULT, lane);                 // LRF:165-176
        const double x = a.mult_l * (u.b - .5);
        out->x[lane] = x, out->m[lane] = exp(x), out->da[lane] = u.a;
    } else {
        const uint32_t purpose = (lane == 0) ? LR_P_ACCEPT : (lane == 1 ? LR_P_MOVE : LR_P_RJ);
        const lr_u2 ud = lr_pair(rng, it, purpose, lane == 3 ? 1u : 0u);
        const double lu = lr_log(lane == 0 ? ud.a : 1.0);
        const double r_a = lr_bcast(ud.a, 1), q_b = lr_bcast(ud.b, 2);
        double beta = 0.0;
        if (!(r_a < 0.8) && r_a < 0.999 && cfg.const_rates == 0 && q_b > 0.5) {
            double ga, gb;
            lr_wave_gamma2(rng, it, LR_P_BETA_A, LR_SHAPE_BETA_RJ, LR_P_BETA_B, LR_SHAPE_BETA_RJ, lane, &ga, &gb);
            beta = ga / (ga + gb);
        }
        // slots: log_u, r_a, r_b, q_a, q_b, q2_a, q2_b, beta  <-  lanes 0..3 of (ud.a, ud.b)
        double so = lu;
        so = (lane == 1) ? r_a : so;
        so = (lane == 2) ? lr_bcast(ud.b, 1) : so;
        so = (lane == 3) ? lr_bcast(ud.a, 2) : so;
        so = (lane == 4) ? q_b : so;
        so = (lane == 5) ? lr_bcast(ud.a, 3) : so;
        so = (lane == 6) ? lr_bcast(ud.b, 3) : so;
        so = (lane == 7) ? beta : so;
        if (lane < 8) out->sc[lane] = so;
    }
}

#define LR_UD_LANE 32   /* first of the four lanes holding the wave-uniform draws of lr_propose_rj's one Philox call */
static_assert(LR_KMAX <= LR_UD_LANE && LR_UD_LANE + 4 <= LR_WAVE, "one lane per rate below the wave-uniform draws");

// update_multiplier_freq with the exponent and the factor already drawn
__device__ __forceinline__ double lr_wave_multiplier_pre(double& R, int K, bool ff, double x, double m, int lane) {
    const bool active = (lane < K) && ff;
    if (active) R = R * m;
    return lr_wave_sum(active ? x : 0.0);
}

// the same draws by ONE wave in ONE Philox call, addressed as lr_propose_rj addresses them when it draws by itself: lanes
// 0..31 the multiplier pairs of their rates, lanes LR_UD_LANE + 0..3 the wave-uniform ones (the four-chain kernel with
// helper waves: the kernel is bound by its vector instruction count, a Philox block is ~100 instructions)
__device__ __forceinline__ void lr_spec_draw_both(const lr_step_args& a, int c, int lane, unsigned long long it, lr_draw_slot* out) {
    const lr_mcmc_config& cfg = a.cfg;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const int ul = lane - LR_UD_LANE;
    const uint32_t purpose = (ul < 0) ? LR_P_MULT : ((ul == 0) ? LR_P_ACCEPT : (ul == 1 ? LR_P_MOVE : LR_P_RJ));
    const lr_u2 u = lr_pair(rng, it, purpose, ul < 0 ? (uint32_t)lane : (ul == 3 ? 1u : 0u));
    const double x = a.mult_l * (u.b - .5);                                // LRF:165-176
    out->x[lane] = x, out->m[lane] = exp(x), out->da[lane] = u.a;          // (lanes 0..31 are read)
    const double lu = lr_log(ul == 0 ? u.a : 1.0);
    const double r_a = lr_bcast(u.a, LR_UD_LANE + 1), q_b = lr_bcast(u.b, LR_UD_LANE + 2);
    double beta = 0.0;
    if (!(r_a < 0.8) && r_a < 0.999 && cfg.const_rates == 0 && q_b > 0.5) {
        double ga, gb;
        lr_wave_gamma2(rng, it, LR_P_BETA_A, LR_SHAPE_BETA_RJ, LR_P_BETA_B, LR_SHAPE_BETA_RJ, lane, &ga, &gb);
        beta = ga / (ga + gb);
    }
    // slots: log_u, r_a, r_b, q_a, q_b, q2_a, q2_b, beta
    double so = lr_bcast(lu, LR_UD_LANE);
    so = (lane == 1) ? r_a : so;
    so = (lane == 2) ? lr_bcast(u.b, LR_UD_LANE + 1) : so;
    so = (lane == 3) ? lr_bcast(u.a, LR_UD_LANE + 2) : so;
    so = (lane == 4) ? q_b : so;
    so = (lane == 5) ? lr_bcast(u.a, LR_UD_LANE + 3) : so;
    so = (lane == 6) ? lr_bcast(u.b, LR_UD_LANE + 3) : so;
    so = (lane == 7) ? beta : so;
    if (lane < 8) out->sc[lane] = so;
}

// Propose iteration `it` from the state `s` (LRF:234-304): on return `s` IS the proposal (a Gibbs step changes the
// hyper-parameters, every other move the rates / times of one process), `p` its bookkeeping, and its lookup tables
// stand at `table`.  A pure function of (s, it, the chain's Philox stream, the data): the speculative engine calls it
// on both possible outcomes of the pending decision.
// hand-over of a proposal's table build to a helper wave (speculative kernel, a team per chain: lr_spec_help_role)
struct lr_table_hand {
    int epoch;             // launch-local number (iteration + 1) of the proposal whose segments stand in the scratch
    int KL, KM;            // its numbers of rates
    int reuse;             // the base state's bin ranks still hold (lr_seg_cache)
    int noop;              // the candidate wave has copied its base state's column: nothing to build
    int out_idx, base_idx; // sets of the proposal and of the state it was made from
    int pad_;
    double par[8];         // parametric samplers: the proposed parameter vector (what their tables are a function of)
};

// CS: doubles between consecutive entries of the chain's column at `table` (2: inside a pair table, 1: a column of its
// own).  base_col (CS == 1 only): the column of the state `s` comes in as, `col_doubles` long, and base_const its model
// constant: a proposal that changes no rate and no bin edge - the no-op "times" moves (LRF:178-195: 40 % of the
// iterations once both processes hold a shift) and the Gibbs step - has the SAME lookup tables as its base state, so
// its column is copied (a handful of LDS moves) instead of built (the largest single item of a proposal).  The copy
// holds the very doubles a rebuild would produce - the builder is a pure function of rates and edges.
// HAND (with `hand`): the table is built by ANOTHER wave: as soon as the proposal's segments stand in the scratch they are
// handed over (epoch `hand_epoch`), this wave goes on with the guard and the prior and leaves the column, its pair planes,
// the model constant and the rank cache of the proposal to the helper (p.table_by_helper) - this instance then holds no
// table builder at all.  base_col is then the base state's whole table, `col_doubles` entries CS doubles apart.
template <bool LDS_CONSTS = false, int PB = 0, int CS = 2, bool PAIR_PLANES = true, bool HAND = false>
__device__ __forceinline__ void lr_propose_rj(const lr_step_args& a, int c, int lane, lr_seg_scratch* scratch_p,
                                              uint64_t it, lr_rj_state& s, lr_rj_prop& p, double2* table,
                                              int table_es, const lr_rj_draws* pre = nullptr,
                                              const double* br_lds = nullptr, const double* logbr_lds = nullptr,
                                              const double* base_col = nullptr, int col_doubles = 0, double base_const = 0.0,
                                              lr_table_hand* hand = nullptr, int hand_epoch = 0) {
    static_assert(CS == 2 || (PB > 0 && !PAIR_PLANES), "a column of its own: one-pass builder, no pair planes");
    static_assert(!HAND || PB > 0, "a helper wave runs the one-pass builder");
    lr_seg_scratch& scratch = *scratch_p;
    const lr_mcmc_config& cfg = a.cfg;
    const int n_bins = cfg.n_bins;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const double L = s.L, M = s.M, tL = s.tL, tM = s.tM;
    const int eL = s.eL, eM = s.eM, KL = s.KL, KM = s.KM;
    double g0 = s.g0, g1 = s.g1, poi = s.poi, lg0 = s.lg0, lg1 = s.lg1, lpoi = s.lpoi;
    const double priorPoiA = s.priorPoi;

    // The wave-uniform draws of the iteration in ONE Philox call (a block costs 40 quarter-rate 32-bit multiplies
    // whether one lane needs it or all 64): lane 0 its acceptance uniform, lane 1 the move selector, lanes 2..3 the
    // two RJ pairs.  Same (iteration, purpose, index) addresses as separate calls would use, so the stream is
    // unchanged.  The logarithm of the acceptance uniform rides along in the packed log of the rates below and
    // waits with the proposal until it is decided, one step later.
    // Lanes 0..31 draw the multiplier pairs of their rates in the same call (at most LR_KMAX = 32 rates; used by
    // 40 % of the iterations, free in the others), the wave-uniform draws sit in lanes LR_UD_LANE + 0..3.
    lr_u2 ud{0.0, 0.0};
    if (!pre) {
        const int ul = lane - LR_UD_LANE;
        const uint32_t purpose = (ul < 0) ? LR_P_MULT : ((ul == 0) ? LR_P_ACCEPT : (ul == 1 ? LR_P_MOVE : LR_P_RJ));
        ud = lr_pair(rng, it, purpose, ul < 0 ? (uint32_t)lane : (ul == 3 ? 1u : 0u));
    }
    const double u_next = pre ? 1.0 : lr_bcast(ud.a, LR_UD_LANE);
    LR_SSTAMP(2);

    double pL = L, pM = M, ptL = tL, ptM = tM;
    int peL = eL, peM = eM, PKL = KL, PKM = KM;
    double hasting = 0.0, priorPoi = 0.0;
    int gibbs = 0, invalid = 0, move_kind;
    const double sample_shift_mu = cfg.const_death_rate ? 0.0 : 0.5;
    const double b_freq = cfg.const_death_rate ? 0.7 : 0.4, d_freq = 0.8;
    const double fL = cfg.update_fraction, fM = cfg.const_death_rate ? 1.0 : cfg.update_fraction;
    const lr_u2 r = pre ? lr_u2{pre->r_a, pre->r_b} : lr_u2{lr_bcast(ud.a, LR_UD_LANE + 1), lr_bcast(ud.b, LR_UD_LANE + 1)};
    if (r.a < b_freq) {
        if (r.b < .5 || KL == 1) {
            if (pre) {
                hasting = lr_wave_multiplier_pre(pL, KL, pre->da < fL, pre->x, pre->m, lane);
            } else {
                hasting = lr_wave_multiplier(pL, KL, ud.a < fL, ud.b, a.mult_l, lane);
            }
            move_kind = 0;
        } else {
            peL = lr_wave_edges(tL, 0);  // update_times leaves the times unchanged (LRF:178-195)
            move_kind = 1;
        }
    } else if (r.a < d_freq) {
        if (r.b < .5 || KM == 1) {
            if (pre) {
                hasting = lr_wave_multiplier_pre(pM, KM, pre->da < fM, pre->x, pre->m, lane);
            } else {
                hasting = lr_wave_multiplier(pM, KM, ud.a < fM, ud.b, a.mult_l, lane);
            }
            move_kind = 2;
        } else {
            peM = lr_wave_edges(tM, 0);
            move_kind = 3;
        }
    } else if (r.a < 0.999 && cfg.const_rates == 0) {
        // RJMCMC (LRF:71-97)
        move_kind = 4;
        const lr_u2 q = pre ? lr_u2{pre->q_a, pre->q_b} : lr_u2{lr_bcast(ud.a, LR_UD_LANE + 2), lr_bcast(ud.b, LR_UD_LANE + 2)};
        const bool sideL = q.a > sample_shift_mu;
        double R = sideL ? L : M, T = sideL ? tL : tM;
        int K = sideL ? KL : KM;
        double score = 0.0;
        const lr_u2 q2 = pre ? lr_u2{pre->q2_a, pre->q2_b} : lr_u2{lr_bcast(ud.a, LR_UD_LANE + 3), lr_bcast(ud.b, LR_UD_LANE + 3)};
        if (q.b > 0.5) {
            if (K >= LR_KMAX) {
                // device cap on the number of rates; the reference has none (LRF:29-47): the proposal is invalid FOR THAT
                // REASON (LR_INVALID_KCAP) and the engine's warning word is raised when it is DECIDED (lr_warn_kcap at the
                // Metropolis-Hastings test) - not here, where the speculative kernel also builds candidates from
                // proposals that are then rejected and never become the chain's pending proposal
                invalid = LR_INVALID_KCAP;
            } else {
                const int ind = min((int)(q2.a * K), K - 1);
                const double delta = q2.b * (lr_bcast(T, ind + 1) - lr_bcast(T, ind));
                double beta;
                if (pre) {
                    beta = pre->beta;
                } else {
                    double ga, gb;
                    lr_wave_gamma2(rng, it, LR_P_BETA_A, LR_SHAPE_BETA_RJ, LR_P_BETA_B, LR_SHAPE_BETA_RJ, lane, &ga, &gb);
                    beta = ga / (ga + gb);
                }
                score = lr_wave_add_shift(R, T, K, ind, delta, beta, lane);
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
        const double lx = lr_log(lane == 0 ? g0 : (lane == 1 ? g1 : (lane == 2 ? poi : 1.0)));
        lg0 = lr_bcast(lx, 0), lg1 = lr_bcast(lx, 1), lpoi = lr_bcast(lx, 2);
        gibbs = 1;
    }

    LR_SSTAMP(3);
    // segments of the proposal -> LDS scratch (+ log of every rate in one call)
    double logpL, logpM, log_u_next;
    lr_stage_segments(&scratch, pL, pM, peL, peM, PKL, PKM, lane, &logpL, &logpM, u_next, &log_u_next);
    if (pre) log_u_next = pre->log_u;

    // the ranks of the bins can be taken over from the state the proposal was made from while no edge moved (every
    // multiplier move and Gibbs step, most of the no-op times moves); then, if no rate changed either, so can its column
    lr_seg_cache sg{s.sgL, s.sgM, false};
    bool edges_same = false;
    if ((pre && s.sg_valid) || base_col) edges_same = PKL == KL && PKM == KM && __ballot(lane <= LR_KMAX && (peL != eL || peM != eM)) == 0ull;
    if (pre && s.sg_valid) sg.reuse = edges_same;
    const bool same_table = base_col && edges_same && (move_kind == 1 || move_kind == 3 || move_kind == 5);
    if (HAND) {
        if (lane == 0) hand->KL = PKL, hand->KM = PKM, hand->reuse = sg.reuse ? 1 : 0, hand->noop = same_table ? 1 : 0;
        LR_WAVE_LDS_ORDER();      // (a wave's LDS operations execute in order: the scratch and the fields stand before the epoch)
        if (lane == 0) __hip_atomic_store(&hand->epoch, hand_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    LR_SSTAMP(4);
    // guard against tiny time frames (LRF:290-292) and the prior of the proposal (LRF:296-304)
    double priorP = -INFINITY;
    {
        // one reduction for both processes: min over all segment lengths
        const double nL = lr_dpp_zero<0x130 /* wave_shl:1 */, 0xf, 0xf>(ptL);   // lane l <- element l+1
        const double nM = lr_dpp_zero<0x130, 0xf, 0xf>(ptM);
        const double dmin = fmin(lane < PKL ? fabs(nL - ptL) : 1e300, lane < PKM ? fabs(nM - ptM) : 1e300);
        if (__ballot(dmin <= LR_MIN_ALLOWED_T)) invalid |= LR_INVALID_GUARD;      // min <= 1  <=>  any <= 1: one compare, no reduction
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
    double constP = 0.0;
    p.table_by_helper = 0;
    if (same_table) {
        double* col = reinterpret_cast<double*>(table);
        for (int i = lane; i < col_doubles; i += LR_WAVE) col[CS * i] = base_col[CS * i];
        constP = base_const;
    } else if (HAND) {
        p.table_by_helper = 1;
    } else {
        // (the per-bin data constants from the caller's LDS copies when it keeps some)
        constP = lr_build_tables_segments<PB, CS>(&scratch, peL, peM, PKL, PKM, LDS_CONSTS ? br_lds : a.br_length,
                                                  LDS_CONSTS ? logbr_lds : a.log_br, cfg.model,
                                                  n_bins, a.n_cls, a.H, table, lane,
                                                  lr_tab_mode<LDS_CONSTS>(a, table_es), cfg.frac_birth, cfg.frac_death,
                                                  table_es, pre ? &sg : nullptr);
        s.sgL = sg.packL, s.sgM = sg.packM, s.sg_valid = (pre && sg.packL != -1) ? 1 : 0;
    }
    if (LDS_CONSTS && PAIR_PLANES && !HAND && !same_table) {
        // persistent engines: the pair planes the packed scan gathers from (lr_scan.h; a team per pair of the speculative
        // kernel leaves them to its scanner waves, which derive them for the selected columns only; a copied table
        // brings its planes along)
        LR_WAVE_LDS_ORDER();
        if (lr_tab_mode<LDS_CONSTS>(a, table_es) == LR_TAB_UNIT) lr_pair_planes_wave(reinterpret_cast<double*>(table), a.H, n_bins, lane, 0);
        else lr_pair_planes_wave_general(reinterpret_cast<double*>(table), a.H, n_bins, lane, 0);
    }
    LR_SSTAMP(6);
    s.L = pL, s.M = pM, s.tL = ptL, s.tM = ptM, s.eL = peL, s.eM = peM, s.KL = PKL, s.KM = PKM;
    s.g0 = g0, s.g1 = g1, s.poi = poi, s.lg0 = lg0, s.lg1 = lg1, s.lpoi = lpoi, s.priorPoi = priorPoi;
    p.hasting = hasting, p.prior = priorP, p.constP = constP, p.log_u = log_u_next;
    p.gibbs = gibbs, p.invalid = invalid, p.move = move_kind;
}

// One chain step on register-resident state: accept the pending proposal given its log-likelihood sum
// (lik_sum, without the model constant), write the trace row, draw the next proposal and build its lookup
// tables at `table` (global memory or LDS; see lr_chain_table for the addressing).
// mode: 0 = regular step, 1 = finish init (adopt the evaluated initial state, then propose iteration 0), 2 = regular step
// with the pending proposal taken as REJECTED whatever lik_sum is (the streaming kernel's speculation, lr_stream.hip: its
// caller knows the proposal to be neither a Gibbs step nor invalid, and sets LR_S_LIK_P once the scan's sums are in)
// HAND (four-chain kernel with helper waves): the proposal's lookup tables, pair planes and model constant are another
// wave's (lr_propose_rj's HAND) - the scalar LR_S_CONST_P of the state then belongs to that wave, see lr_chain_store_handed
template <bool LDS_CONSTS = false, int PB = 0, bool HAND = false>
__device__ __forceinline__ void lr_chain_step_core(lr_chain_regs& st, const lr_step_args& a, int mode, int c, int lane,
                                                   lr_seg_scratch* scratch_p, double lik_sum, double2* table,
                                                   int table_es = 2, const double* br_lds = nullptr,
                                                   const double* logbr_lds = nullptr, const lr_rj_draws* pre = nullptr,
                                                   lr_table_hand* hand = nullptr, int hand_epoch = 0) {
    const lr_mcmc_config& cfg = a.cfg;
    const double sc = st.sc;
    const int isc = st.isc;
    double lik_p = lr_bcast(sc, LR_S_LIK_P);
    double likA = lr_bcast(sc, LR_S_LIKA), priorA = lr_bcast(sc, LR_S_PRIORA);
    double constA = lr_bcast(sc, LR_S_CONST_A);
    // the accepted state
    lr_rj_state s;
    s.L = st.L, s.M = st.M, s.tL = st.tL, s.tM = st.tM, s.eL = st.eL, s.eM = st.eM;
    s.KL = lr_bcast_i(isc, LR_I_KL), s.KM = lr_bcast_i(isc, LR_I_KM);
    s.g0 = lr_bcast(sc, LR_S_GRATE_L), s.g1 = lr_bcast(sc, LR_S_GRATE_M), s.poi = lr_bcast(sc, LR_S_POI);
    s.lg0 = lr_bcast(sc, LR_S_LOG_G0), s.lg1 = lr_bcast(sc, LR_S_LOG_G1), s.lpoi = lr_bcast(sc, LR_S_LOG_POI);
    s.priorPoi = lr_bcast(sc, LR_S_PRIORPOIA);
    s.sgL = s.sgM = 0, s.sg_valid = 0;
    uint64_t it = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_HI) << 32);
    int n_acc = lr_bcast_i(isc, LR_I_ACCEPTED);
    uint64_t next_sample = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_HI) << 32);
    int trace_slot = lr_bcast_i(isc, LR_I_SLOT);
    LR_SSTAMP(1);

    if (mode == 1) {
        // LRF:224-230.  The initial prior uses prior_gamma's default rate b=2 (LRF:201, 227).
        likA = lik_sum + constA;
        priorA = lr_wave_prior_gamma(s.L, s.KL, LR_GAMMA_SHAPE, 2.0, lane) + lr_wave_prior_gamma(s.M, s.KM, LR_GAMMA_SHAPE, 2.0, lane);
        priorA += -a.log_T * (s.KL - 1 + s.KM - 1);
        s.priorPoi = lr_poisson_prior(s.KL, s.poi, s.lpoi) + lr_poisson_prior(s.KM, s.poi, s.lpoi);
        priorA += s.priorPoi;
    } else {
        // ---- Metropolis-Hastings accept of iteration `it` (LRF:305-319) ----
        const int gibbs = lr_bcast_i(isc, LR_I_GIBBS), invalid = lr_bcast_i(isc, LR_I_INVALID);
        const double priorP = lr_bcast(sc, LR_S_PRIOR_P), constP = lr_bcast(sc, LR_S_CONST_P);
        double lik;
        const bool ok = lr_mh_accept(gibbs, invalid, lik_sum, constP, likA, priorP, priorA, lr_bcast(sc, LR_S_HASTING),
                                     lr_bcast(sc, LR_S_LOG_U), &lik) && mode != 2;
        lik_p = invalid ? -INFINITY : lik;
        lr_warn_kcap(a.warn, invalid, lane);
        if (ok) {
            s.L = st.pL, s.M = st.pM, s.tL = st.ptL, s.tM = st.ptM, s.eL = st.peL, s.eM = st.peM;
            s.KL = lr_bcast_i(isc, LR_I_PKL), s.KM = lr_bcast_i(isc, LR_I_PKM);
            likA = lik, priorA = priorP, s.priorPoi = lr_bcast(sc, LR_S_PRIORPOI_P), constA = constP;
            n_acc += 1;
        }
        // ---- trace row (LRF:321-359) ----
        // `it % s_freq == 0` kept as a running (next sample, slot) pair: no 64-bit division on the device
        if (it == next_sample) {
            const int slot = trace_slot;
            trace_slot += 1;
            next_sample += (uint64_t)cfg.s_freq;
            if (slot < cfg.n_trace_slots) lr_write_trace_row(a, c, lane, slot, it, likA, priorA, s);
        }
        it += 1;
    }
    const double priorPoiA = s.priorPoi;
    st.L = s.L, st.M = s.M, st.tL = s.tL, st.tM = s.tM, st.eL = s.eL, st.eM = s.eM;
    const int KL = s.KL, KM = s.KM;

    // ---- propose iteration `it` (LRF:234-304) ----
    lr_rj_prop p;
    if (HAND) lr_propose_rj<LDS_CONSTS, (PB > 0 ? PB : 1), 2, true, HAND>(a, c, lane, scratch_p, it, s, p, table, table_es, pre, br_lds, logbr_lds, nullptr, 0, 0.0, hand, hand_epoch);
    else lr_propose_rj<LDS_CONSTS, PB>(a, c, lane, scratch_p, it, s, p, table, table_es, pre, br_lds, logbr_lds);

    // ---- back into the state registers ----
    st.pL = s.L, st.pM = s.M, st.ptL = s.tL, st.ptM = s.tM, st.peL = s.eL, st.peM = s.eM;
    {
        // scalar slots: branch-free select chains (a switch over the lane id runs every case under its own exec mask)
        double so = 0.0;
        so = (lane == LR_S_LIKA) ? likA : so;
        so = (lane == LR_S_PRIORA) ? priorA : so;
        so = (lane == LR_S_PRIORPOIA) ? priorPoiA : so;
        so = (lane == LR_S_GRATE_L) ? s.g0 : so;
        so = (lane == LR_S_GRATE_M) ? s.g1 : so;
        so = (lane == LR_S_POI) ? s.poi : so;
        so = (lane == LR_S_HASTING) ? p.hasting : so;
        so = (lane == LR_S_PRIOR_P) ? p.prior : so;
        so = (lane == LR_S_PRIORPOI_P) ? s.priorPoi : so;
        so = (lane == LR_S_CONST_P) ? p.constP : so;
        so = (lane == LR_S_CONST_A) ? constA : so;
        so = (lane == LR_S_LIK_P) ? lik_p : so;
        so = (lane == LR_S_LOG_G0) ? s.lg0 : so;
        so = (lane == LR_S_LOG_G1) ? s.lg1 : so;
        so = (lane == LR_S_LOG_POI) ? s.lpoi : so;
        so = (lane == LR_S_LOG_U) ? p.log_u : so;
        st.sc = so;
        int io = 0;
        io = (lane == LR_I_KL) ? KL : io;
        io = (lane == LR_I_KM) ? KM : io;
        io = (lane == LR_I_PKL) ? s.KL : io;
        io = (lane == LR_I_PKM) ? s.KM : io;
        io = (lane == LR_I_GIBBS) ? p.gibbs : io;
        io = (lane == LR_I_INVALID) ? p.invalid : io;
        io = (lane == LR_I_IT_LO) ? (int)(uint32_t)it : io;
        io = (lane == LR_I_IT_HI) ? (int)(uint32_t)(it >> 32) : io;
        io = (lane == LR_I_ACCEPTED) ? n_acc : io;
        io = (lane == LR_I_MOVE) ? p.move : io;
        io = (lane == LR_I_NEXT_LO) ? (int)(uint32_t)next_sample : io;
        io = (lane == LR_I_NEXT_HI) ? (int)(uint32_t)(next_sample >> 32) : io;
        io = (lane == LR_I_SLOT) ? trace_slot : io;
        st.isc = io;
    }
    LR_SSTAMP(7);

}

// ---- the chain step that speculates on REJECTION (four-chain kernel with helper waves: lr_persist4_kernel's SPEC) --------
// On cfg4-sized data a chain's accepted state changes in 11-13 % of its first 3000 iterations and in < 1 % of the later
// ones (scratch/exp_accept_mix.py: multiplier moves of +-10 % against a posterior ~1 % wide are rejected, the no-op
// "times" moves are accepted but change nothing).  So while the pending proposal P of iteration `it` is being scanned the
// stepper wave already stages Q = propose(A, it + 1) - the proposal the chain makes next IF P is rejected.  At the decision,
// one phase later:
//   * state unchanged (the usual case): Q IS the pending proposal, its segments already stand in the scratch, and the
//     helper wave starts the table build right at the decision instead of waiting ~1.2 us for move + staging;
//   * state changed: Q is dropped, propose(A', it + 1) runs as before (staging, hand-over, helper's build).
// Either way the wave then stages propose(A', it + 2).  A proposal lives in the slot of its iteration's PARITY (rows,
// bookkeeping scalars, scratch, hand-over word, draws): the slot of `it` is dead at the decision and takes the proposal of
// it + 2, the slot of it + 1 holds Q - nothing is copied when Q is adopted.  lr_propose_rj is a pure function of (state,
// iteration, draws): the trajectories are those of lr_chain_step_core bit for bit, whatever is dropped at a launch boundary.
struct lr_pend {
    double L[LR_ROW], M[LR_ROW], tL[LR_ROW], tM[LR_ROW];
    double sc[LR_ROW];      // LR_S_* layout: the slots of the PENDING side (lr_sc_pending_side) and LR_S_CONST_P (the helper's)
    int eL[LR_ROW], eM[LR_ROW];
    int isc[LR_ROW];        // LR_I_* layout: PKL, PKM, GIBBS, INVALID, MOVE
};
// scalar slots that describe a pending proposal: hyper-parameters and their logs ride with it (a Gibbs step's draws)
__device__ __forceinline__ bool lr_sc_pending_side(int lane) {
    constexpr unsigned m = (1u << LR_S_GRATE_L) | (1u << LR_S_GRATE_M) | (1u << LR_S_POI) | (1u << LR_S_HASTING) | (1u << LR_S_PRIOR_P) |
                           (1u << LR_S_PRIORPOI_P) | (1u << LR_S_CONST_P) | (1u << LR_S_LOG_G0) | (1u << LR_S_LOG_G1) |
                           (1u << LR_S_LOG_POI) | (1u << LR_S_LOG_U);
    return lane < 32 && ((m >> lane) & 1u);
}
__device__ __forceinline__ bool lr_isc_pending_side(int lane) {
    constexpr unsigned m = (1u << LR_I_PKL) | (1u << LR_I_PKM) | (1u << LR_I_GIBBS) | (1u << LR_I_INVALID) | (1u << LR_I_MOVE);
    return lane < 32 && ((m >> lane) & 1u);
}
// the chain's state rows (include/literate_hip.h layout) <-> the accepted side in the rows + the pending proposal in a slot.
// In the rows' scalar row the hyper-parameter slots then mean the ACCEPTED state's.
__device__ __forceinline__ void lr_pend_from_rows(const double* S, const int* I, lr_pend* P, int lane) {
    P->L[lane] = S[LR_ROW_PL * LR_ROW + lane], P->M[lane] = S[LR_ROW_PM * LR_ROW + lane];
    P->tL[lane] = S[LR_ROW_PTL * LR_ROW + lane], P->tM[lane] = S[LR_ROW_PTM * LR_ROW + lane];
    P->eL[lane] = I[LR_IROW_PEL * LR_ROW + lane], P->eM[lane] = I[LR_IROW_PEM * LR_ROW + lane];
    P->sc[lane] = S[LR_ROW_SCALARS * LR_ROW + lane], P->isc[lane] = I[LR_IROW_SCALARS * LR_ROW + lane];
}
__device__ __forceinline__ void lr_pend_to_rows(double* S, int* I, const lr_pend* P, int lane) {
    S[LR_ROW_PL * LR_ROW + lane] = P->L[lane], S[LR_ROW_PM * LR_ROW + lane] = P->M[lane];
    S[LR_ROW_PTL * LR_ROW + lane] = P->tL[lane], S[LR_ROW_PTM * LR_ROW + lane] = P->tM[lane];
    I[LR_IROW_PEL * LR_ROW + lane] = P->eL[lane], I[LR_IROW_PEM * LR_ROW + lane] = P->eM[lane];
    if (lr_sc_pending_side(lane)) S[LR_ROW_SCALARS * LR_ROW + lane] = P->sc[lane];
    if (lr_isc_pending_side(lane)) I[LR_IROW_SCALARS * LR_ROW + lane] = P->isc[lane];
}

// pend2 / scratch2 / hands2 / draws2: the chain's two slots each, indexed by the parity of the iteration a proposal is FOR.
// S / I: the chain's state rows (LDS), accepted side.  `first`: the first decision of a launch (no staged candidate yet).
// Rows are read and written where they are needed instead of living in registers for the whole step: a proposal is ~100
// live registers by itself, and what is held across it spills to scratch memory.
template <int PB>
__device__ __forceinline__ void lr_chain_step_respec(const lr_step_args& a, int c, int lane, double lik_sum, double* S, int* I,
                                                     lr_pend* pend2, lr_seg_scratch* scratch2, lr_table_hand* hands2,
                                                     const lr_draw_slot* draws2, bool first, int epoch, const double* br_lds,
                                                     const double* logbr_lds) {
    const lr_mcmc_config& cfg = a.cfg;
    const double sc = S[LR_ROW_SCALARS * LR_ROW + lane];
    const int isc = I[LR_IROW_SCALARS * LR_ROW + lane];
    const uint64_t it = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_HI) << 32);
    const int pp = (int)(it & 1);                      // slot of the pending proposal; 1 - pp: of the staged candidate
    const double psc = pend2[pp].sc[lane];
    const int pisc = pend2[pp].isc[lane];
    double likA = lr_bcast(sc, LR_S_LIKA), priorA = lr_bcast(sc, LR_S_PRIORA);
    double constA = lr_bcast(sc, LR_S_CONST_A), priorPoiA = lr_bcast(sc, LR_S_PRIORPOIA);
    int KL = lr_bcast_i(isc, LR_I_KL), KM = lr_bcast_i(isc, LR_I_KM);
    double g0 = lr_bcast(sc, LR_S_GRATE_L), g1 = lr_bcast(sc, LR_S_GRATE_M), poi = lr_bcast(sc, LR_S_POI);
    double lg0 = lr_bcast(sc, LR_S_LOG_G0), lg1 = lr_bcast(sc, LR_S_LOG_G1), lpoi = lr_bcast(sc, LR_S_LOG_POI);
    int n_acc = lr_bcast_i(isc, LR_I_ACCEPTED);
    uint64_t next_sample = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_HI) << 32);
    int trace_slot = lr_bcast_i(isc, LR_I_SLOT);
    LR_SSTAMP(1);

    // ---- Metropolis-Hastings accept of iteration `it` (LRF:305-319) ----
    const int gibbs = lr_bcast_i(pisc, LR_I_GIBBS), invalid = lr_bcast_i(pisc, LR_I_INVALID);
    const double priorP = lr_bcast(psc, LR_S_PRIOR_P), constP = lr_bcast(psc, LR_S_CONST_P);
    double lik;
    const bool ok = lr_mh_accept(gibbs, invalid, lik_sum, constP, likA, priorP, priorA, lr_bcast(psc, LR_S_HASTING),
                                 lr_bcast(psc, LR_S_LOG_U), &lik);
    const double lik_p = invalid ? -INFINITY : lik;
    lr_warn_kcap(a.warn, invalid, lane);
    // Does the decision change what the next proposal is made FROM?  Not when P is rejected, and not when an accepted P is a
    // no-op "times" move whose re-floored edges are the accepted ones (LRF:178-195): rates, times, edges, hyper-parameters
    // and the cached Poisson prior are then what they were (likA / priorA / the counter are not read by a proposal).
    bool changed = first;
    if (ok) {
        const int mv = lr_bcast_i(pisc, LR_I_MOVE);
        const int peL = pend2[pp].eL[lane], peM = pend2[pp].eM[lane];
        const int eL = I[LR_IROW_EL * LR_ROW + lane], eM = I[LR_IROW_EM * LR_ROW + lane];
        const bool same = (mv == 1 || mv == 3) && __ballot(lane <= LR_KMAX && (peL != eL || peM != eM)) == 0ull;
        changed |= !same;
        if (!same) {
            // the pending rows become the accepted ones (a no-op move's are the very same values)
            S[LR_ROW_L * LR_ROW + lane] = pend2[pp].L[lane], S[LR_ROW_M * LR_ROW + lane] = pend2[pp].M[lane];
            S[LR_ROW_TL * LR_ROW + lane] = pend2[pp].tL[lane], S[LR_ROW_TM * LR_ROW + lane] = pend2[pp].tM[lane];
            I[LR_IROW_EL * LR_ROW + lane] = peL, I[LR_IROW_EM * LR_ROW + lane] = peM;
            KL = lr_bcast_i(pisc, LR_I_PKL), KM = lr_bcast_i(pisc, LR_I_PKM);
            g0 = lr_bcast(psc, LR_S_GRATE_L), g1 = lr_bcast(psc, LR_S_GRATE_M), poi = lr_bcast(psc, LR_S_POI);
            lg0 = lr_bcast(psc, LR_S_LOG_G0), lg1 = lr_bcast(psc, LR_S_LOG_G1), lpoi = lr_bcast(psc, LR_S_LOG_POI);
        }
        likA = lik, priorA = priorP, priorPoiA = lr_bcast(psc, LR_S_PRIORPOI_P), constA = constP;
        n_acc += 1;
    }
    const uint64_t it1 = it + 1;                      // iteration of the new pending proposal: slot 1 - pp
    if (!changed && lane == 0)                        // the staged candidate IS that proposal: the helper may build its table
        __hip_atomic_store(&hands2[1 - pp].epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LR_WAVE_LDS_ORDER();
    // ---- trace row (LRF:321-359) ----
    if (it == next_sample) {
        const int slot = trace_slot;
        trace_slot += 1;
        next_sample += (uint64_t)cfg.s_freq;
        if (slot < cfg.n_trace_slots) {
            lr_rj_state s;
            s.L = S[LR_ROW_L * LR_ROW + lane], s.M = S[LR_ROW_M * LR_ROW + lane];
            s.tL = S[LR_ROW_TL * LR_ROW + lane], s.tM = S[LR_ROW_TM * LR_ROW + lane];
            s.KL = KL, s.KM = KM, s.g0 = g0, s.g1 = g1, s.poi = poi;
            lr_write_trace_row(a, c, lane, slot, it, likA, priorA, s);
        }
    }
    // the accepted side of the scalar rows (nothing below changes it)
    {
        double so = 0.0;
        so = (lane == LR_S_LIKA) ? likA : so;
        so = (lane == LR_S_PRIORA) ? priorA : so;
        so = (lane == LR_S_PRIORPOIA) ? priorPoiA : so;
        so = (lane == LR_S_CONST_A) ? constA : so;
        so = (lane == LR_S_LIK_P) ? lik_p : so;
        so = (lane == LR_S_GRATE_L) ? g0 : so;
        so = (lane == LR_S_GRATE_M) ? g1 : so;
        so = (lane == LR_S_POI) ? poi : so;
        so = (lane == LR_S_LOG_G0) ? lg0 : so;
        so = (lane == LR_S_LOG_G1) ? lg1 : so;
        so = (lane == LR_S_LOG_POI) ? lpoi : so;
        S[LR_ROW_SCALARS * LR_ROW + lane] = so;
        int io = 0;
        io = (lane == LR_I_KL) ? KL : io;
        io = (lane == LR_I_KM) ? KM : io;
        io = (lane == LR_I_IT_LO) ? (int)(uint32_t)it1 : io;
        io = (lane == LR_I_IT_HI) ? (int)(uint32_t)(it1 >> 32) : io;
        io = (lane == LR_I_ACCEPTED) ? n_acc : io;
        io = (lane == LR_I_NEXT_LO) ? (int)(uint32_t)next_sample : io;
        io = (lane == LR_I_NEXT_HI) ? (int)(uint32_t)(next_sample >> 32) : io;
        io = (lane == LR_I_SLOT) ? trace_slot : io;
        I[LR_IROW_SCALARS * LR_ROW + lane] = io;
    }
    LR_SSTAMP(2);

    // ---- k = 0 (state changed only): propose(A', it1) into slot 1 - pp, handed to the helper at once;
    //      k = 1: propose(A', it1 + 1) into slot pp, staged for the next decision.  ONE instance of the proposal code.
    int k = changed ? 0 : 1;
#pragma unroll 1
    while (true) {
        const uint64_t itk = it1 + (uint64_t)k;
        const int par = (int)(itk & 1);
        lr_rj_state sk;
        sk.L = S[LR_ROW_L * LR_ROW + lane], sk.M = S[LR_ROW_M * LR_ROW + lane];
        sk.tL = S[LR_ROW_TL * LR_ROW + lane], sk.tM = S[LR_ROW_TM * LR_ROW + lane];
        sk.eL = I[LR_IROW_EL * LR_ROW + lane], sk.eM = I[LR_IROW_EM * LR_ROW + lane];
        sk.KL = KL, sk.KM = KM, sk.g0 = g0, sk.g1 = g1, sk.poi = poi, sk.lg0 = lg0, sk.lg1 = lg1, sk.lpoi = lpoi;
        sk.priorPoi = priorPoiA;
        sk.sgL = sk.sgM = 0, sk.sg_valid = 0;
        lr_rj_prop p;
        lr_rj_draws pre;
        lr_draws_load(&draws2[par], pre, lane);
        lr_propose_rj<true, PB, 2, true, true>(a, c, lane, &scratch2[par], itk, sk, p, nullptr, 2, &pre, br_lds, logbr_lds, nullptr, 0,
                                               0.0, &hands2[par], k == 0 ? epoch : 0);
        lr_pend* Q = &pend2[par];
        Q->L[lane] = sk.L, Q->M[lane] = sk.M, Q->tL[lane] = sk.tL, Q->tM[lane] = sk.tM;
        Q->eL[lane] = sk.eL, Q->eM[lane] = sk.eM;
        double so = 0.0;
        so = (lane == LR_S_GRATE_L) ? sk.g0 : so;
        so = (lane == LR_S_GRATE_M) ? sk.g1 : so;
        so = (lane == LR_S_POI) ? sk.poi : so;
        so = (lane == LR_S_HASTING) ? p.hasting : so;
        so = (lane == LR_S_PRIOR_P) ? p.prior : so;
        so = (lane == LR_S_PRIORPOI_P) ? sk.priorPoi : so;
        so = (lane == LR_S_LOG_G0) ? sk.lg0 : so;
        so = (lane == LR_S_LOG_G1) ? sk.lg1 : so;
        so = (lane == LR_S_LOG_POI) ? sk.lpoi : so;
        so = (lane == LR_S_LOG_U) ? p.log_u : so;
        if (lane != LR_S_CONST_P) Q->sc[lane] = so;           // (LR_S_CONST_P of a slot: written by the helper that builds its table)
        int io = 0;
        io = (lane == LR_I_PKL) ? sk.KL : io;
        io = (lane == LR_I_PKM) ? sk.KM : io;
        io = (lane == LR_I_GIBBS) ? p.gibbs : io;
        io = (lane == LR_I_INVALID) ? p.invalid : io;
        io = (lane == LR_I_MOVE) ? p.move : io;
        Q->isc[lane] = io;
        LR_WAVE_LDS_ORDER();
        if (k == 1) break;
        k = 1;
    }
    LR_SSTAMP(7);
}

// the step of chain c with its state in global memory: load, sum the tile partials in tile order, step, store

// ---- parametric samplers: one iteration of DDRate.py's loop (DD:194-239; sampler 1) or of trend_rate.py's
// (trend_rate.py:160-195; sampler 2) for one chain -----------------------------------------------------------
// Same pipeline position as lr_chain_step_core: decide the pending proposal with the scanned likelihood, write the
// trace row, propose the next parameter vector, evaluate its prior and build its lookup tables.
// State: lane j < NPAR of st.L = accepted parameter j, of st.pL = proposed parameter j; scalars in st.sc / st.isc.
// `aux` = the per-bin array the rate map needs: DT (DDRate) or the normalised covariate TREND (trend_rate).
// Propose iteration `it` of a parametric sampler from the parameter vector A (lane j < npar): returns the proposal
// (lane j), its Hastings term, prior, move kind and the log of the acceptance uniform of `it`; builds its tables.
struct lr_dd_prop {
    double hasting, prior, log_u;
    int move;
};
// the state-independent draws of one iteration of a parametric sampler (what lr_propose_dd draws itself when none are
// given); a speculative engine makes them one iteration ahead
struct lr_dd_draws {
    double log_u, rr_a, rr_b, slide_u, z1;   // wave-uniform: log of the acceptance uniform, move selectors, sliding-window uniform, its normal
    double da, x, m;                         // per parameter (lane): inclusion uniform, multiplier exponent (trend: or the normal step), exp(x)
};
__device__ __forceinline__ void lr_make_dd_draws(const lr_step_args& a, int c, int lane, uint64_t it, lr_dd_draws& d) {
    const lr_mcmc_config& cfg = a.cfg;
    const bool trend = cfg.sampler == 2;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const uint32_t purpose = trend ? (lane == 0 ? LR_P_TR_ACCEPT : LR_P_TR_MOVE)
                                   : (lane == 0 ? LR_P_DD_ACCEPT : (lane == 1 ? LR_P_DD_MOVE : LR_P_DD_SLIDE));
    const lr_u2 ud = lr_pair(rng, it, purpose, 0u);
    const double lu = lr_log(lane == 0 ? ud.a : 1.0);
    d.log_u = lr_bcast(lu, 0);
    d.rr_a = lr_bcast(ud.a, 1), d.rr_b = lr_bcast(ud.b, 1), d.slide_u = lr_bcast(ud.a, 2);
    d.z1 = 0.0, d.da = 1.0, d.x = 0.0, d.m = 1.0;
    if (trend) {
        const lr_u2 u = lr_pair(rng, it, LR_P_TR_MULT, lane);
        d.da = u.a;
        if (d.rr_a < .33) {
            d.x = lr_normal(rng, it, LR_P_TR_NORM, lane);
        } else {
            d.x = a.mult_l * (u.b - .5);
            d.m = exp(d.x);
        }
    } else if (d.rr_b < 0.1 && (cfg.m_birth == 2 || cfg.m_death == 2)) {
        if (cfg.m_death == -1) d.z1 = lr_normal(rng, it, LR_P_DD_SLIDE, 1);
    } else {
        const lr_u2 u = lr_pair(rng, it, LR_P_DD_MULT, lane);
        d.da = u.a;
        d.x = a.mult_l * (u.b - .5);
        d.m = exp(d.x);
    }
}

// The lookup tables (and, PAIR_PLANES, the pair planes behind them) of a parametric sampler's parameter vector P (lane j
// holds parameter j): DDRate's per-bin rates from the diversity trajectory `aux` = DT (DD:55-100) or trend_rate's from the
// normalised covariate `aux` = TREND (trend_rate.py:73-88).
template <bool LDS_CONSTS, bool PAIR_PLANES, int CS>
__device__ __forceinline__ void lr_param_tables(const lr_step_args& a, double P, const double* aux, double2* table, int mode,
                                                int table_es, int lane) {
    const lr_mcmc_config& cfg = a.cfg;
    if (cfg.sampler == 2) {
        const lr_trend_params tp = lr_trend_unpack(P);
        lr_rates_build_tables_wave<CS>([&](int b, double* br, double* dr) { lr_trend_bin_rates(tp, aux[b], cfg.m_birth, cfg.m_death, br, dr); },
                                       cfg.n_bins, a.H, table, lane, mode, cfg.frac_birth, cfg.frac_death, table_es);
    } else {
        const lr_dd_params pp = lr_dd_unpack(P);
        lr_dd_build_tables_wave<CS>(pp, aux, cfg.m_birth, cfg.m_death, cfg.n_bins, a.H, table, lane, mode, cfg.frac_birth, cfg.frac_death,
                                    table_es);
    }
    if (LDS_CONSTS && PAIR_PLANES) {
        LR_WAVE_LDS_ORDER();
        if (mode == LR_TAB_UNIT) lr_pair_planes_wave(reinterpret_cast<double*>(table), a.H, cfg.n_bins, lane, 0);
        else lr_pair_planes_wave_general(reinterpret_cast<double*>(table), a.H, cfg.n_bins, lane, 0);
    }
}

// HAND (with `hand`): the tables and their pair planes are built by a helper wave from the proposed parameter vector,
// handed over (epoch `hand_epoch`) as soon as it stands; this wave goes on with the prior.
template <bool LDS_CONSTS = false, bool PAIR_PLANES = true, int CS = 2, bool HAND = false>
__device__ __forceinline__ double lr_propose_dd(const lr_step_args& a, int c, int lane, uint64_t it, double A,
                                                lr_dd_prop& p, double2* table, int table_es,
                                                const double* aux_lds = nullptr, const lr_dd_draws* pre = nullptr,
                                                lr_table_hand* hand = nullptr, int hand_epoch = 0) {
    static_assert(CS == 2 || !PAIR_PLANES, "a column of its own carries no pair planes");
    const lr_mcmc_config& cfg = a.cfg;
    const bool trend = cfg.sampler == 2;
    const int npar = trend ? LR_TR_NPAR : LR_DD_NPAR;
    const lr_stream rng{(uint32_t)cfg.seed, (uint32_t)(cfg.chain_offset + c)};
    const double origin = cfg.t0, present = cfg.dd_present;
    const double k0 = a.dd_consts[0], log_k0 = a.dd_consts[1];
    const double* aux = LDS_CONSTS ? aux_lds : a.br_length;
    // wave-uniform draws in one Philox call (see lr_propose_rj): lane 0 the acceptance uniform, lane 1 the move
    // selector and lane 2 the sliding-window uniform of the iteration
    const uint32_t purpose = trend ? (lane == 0 ? LR_P_TR_ACCEPT : LR_P_TR_MOVE)
                                   : (lane == 0 ? LR_P_DD_ACCEPT : (lane == 1 ? LR_P_DD_MOVE : LR_P_DD_SLIDE));
    lr_u2 ud{0.0, 0.0};
    if (!pre) ud = lr_pair(rng, it, purpose, 0u);
    // log of the acceptance uniform (DD:211, trend_rate.py:176), evaluated in lane 0 only
    p.log_u = pre ? pre->log_u : lr_bcast(lr_log(lane == 0 ? ud.a : 1.0), 0);
    double P = A, hasting = 0.0;
    int move_kind;
    if (trend) {
        // trend_rate.py:165-169: 33 % additive normal step on the slopes, else the vector multiplier
        const double rr = pre ? pre->rr_a : lr_bcast(ud.a, 1);
        double f_mult, f_norm;
        lr_trend_update_freq(cfg.m_birth, cfg.m_death, lane, &f_mult, &f_norm);
        lr_u2 d{0.0, 0.0};
        if (!pre) d = lr_pair(rng, it, LR_P_TR_MULT, lane);
        const double da = pre ? pre->da : d.a;
        if (rr < .33) {
            const double z = pre ? pre->x : lr_normal(rng, it, LR_P_TR_NORM, lane);   // update_normal_nobound_vec (lib:140-146)
            if (lane < npar && da < f_norm) P = A + z * .001;
            move_kind = 1;
        } else {
            hasting = pre ? lr_wave_multiplier_pre(P, npar, da < f_mult, pre->x, pre->m, lane)
                          : lr_wave_multiplier(P, npar, da < f_mult, d.b, a.mult_l, lane);   // lib:156-165
            move_kind = 0;
        }
    } else {
        // DD:195-207
        const lr_u2 rr = pre ? lr_u2{pre->rr_a, pre->rr_b} : lr_u2{lr_bcast(ud.a, 1), lr_bcast(ud.b, 1)};
        if (rr.b < 0.1 && (cfg.m_birth == 2 || cfg.m_death == 2)) {
            // update_sliding_win(x0, m=0, M=PRESENT, d=1.5) (lib:124-128)
            double ii = lr_bcast(A, 2) + ((pre ? pre->slide_u : lr_bcast(ud.a, 2)) - .5) * 1.5;
            if (ii > present) ii = present - (ii - present);
            ii = fabs(ii);
            if (lane == 2) P = ii;
            if (cfg.m_death == -1) {
                const double z = pre ? pre->z1 : lr_normal(rng, it, LR_P_DD_SLIDE, 1);   // update_normal_nobound(k, d=0.2) (lib:136-138)
                if (lane == 1) P = A + z * 0.2;
            }
            move_kind = 1;
        } else {
            const double f = lr_dd_update_freq(cfg.m_birth, cfg.m_death, lane);
            if (pre) {
                hasting = lr_wave_multiplier_pre(P, npar, pre->da < f, pre->x, pre->m, lane);
            } else {
                const lr_u2 d = lr_pair(rng, it, LR_P_DD_MULT, lane);
                hasting = lr_wave_multiplier(P, npar, d.a < f, d.b, a.mult_l, lane);   // lib:156-165
            }
            move_kind = 0;
        }
    }
    if (HAND) {
        if (lane < 8) hand->par[lane] = P;
        LR_WAVE_LDS_ORDER();
        if (lane == 0) __hip_atomic_store(&hand->epoch, hand_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        p.prior = trend ? lr_trend_prior(P, lane) : lr_dd_prior(P, origin, present, k0, log_k0, lane);
    } else {
        if (trend) p.prior = lr_trend_prior(P, lane);
        else p.prior = lr_dd_prior(P, origin, present, k0, log_k0, lane);
        lr_param_tables<LDS_CONSTS, PAIR_PLANES, CS>(a, P, aux, table, lr_tab_mode<LDS_CONSTS>(a, table_es), table_es, lane);
    }
    p.hasting = hasting, p.move = move_kind;
    return P;
}

// acceptance rule of the parametric samplers (DD:211, trend_rate.py:176): `>`, iteration 0 always accepted
__device__ __forceinline__ bool lr_dd_accept(double lik, double likA, double priorP, double priorA, double hasting,
                                             double log_u, uint64_t it) {
    return ((lik - likA) + (priorP - priorA) + hasting > log_u) || it == 0;
}

// trace row of a parametric sampler: [it, posterior, likelihood, prior, args[npar]] (DD:221)
__device__ __forceinline__ void lr_dd_write_trace_row(const lr_step_args& a, int c, int lane, int slot, uint64_t it,
                                                      double likA, double priorA, double A) {
    const int npar = a.cfg.sampler == 2 ? LR_TR_NPAR : LR_DD_NPAR;
    double* row = a.trace + ((size_t)slot * a.cfg.n_chains + c) * LR_TRACE_W;
    double h = __longlong_as_double(0x7ff8000000000000LL);
    if (lane == 0) h = (double)it;
    if (lane == 1) h = likA + priorA;
    if (lane == 2) h = likA;
    if (lane == 3) h = priorA;
    const double Aj = __shfl(A, (lane - 4) & (LR_WAVE - 1));        // rare path (sampling only)
    if (lane >= 4 && lane < 4 + npar) h = Aj;
    for (int j = lane; j < LR_TRACE_W; j += LR_WAVE) row[j] = (j == lane) ? h : __longlong_as_double(0x7ff8000000000000LL);
}

template <bool LDS_CONSTS = false>
__device__ __forceinline__ void lr_dd_step_core(lr_chain_regs& st, const lr_step_args& a, int mode, int c, int lane,
                                                double lik_sum, double2* table, int table_es = 2,
                                                const double* aux_lds = nullptr) {
    const lr_mcmc_config& cfg = a.cfg;
    const bool trend = cfg.sampler == 2;
    const int npar = trend ? LR_TR_NPAR : LR_DD_NPAR;
    double A = st.L;
    const double sc = st.sc;
    const int isc = st.isc;
    double likA = lr_bcast(sc, LR_S_LIKA), priorA = lr_bcast(sc, LR_S_PRIORA), lik_p = lr_bcast(sc, LR_S_LIK_P);
    uint64_t it = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_IT_HI) << 32);
    int n_acc = lr_bcast_i(isc, LR_I_ACCEPTED);
    uint64_t next_sample = (uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_LO) | ((uint64_t)(uint32_t)lr_bcast_i(isc, LR_I_NEXT_HI) << 32);
    int trace_slot = lr_bcast_i(isc, LR_I_SLOT);
    if (mode == 1) {
        likA = lik_sum;                                                        // DD:184-186, trend_rate.py:150-151
        priorA = trend ? lr_trend_prior(A, lane)
                       : lr_dd_prior(A, cfg.t0, cfg.dd_present, a.dd_consts[0], a.dd_consts[1], lane);
    } else {
        const double priorP = lr_bcast(sc, LR_S_PRIOR_P);
        const double lik = lik_sum;
        const bool ok = lr_dd_accept(lik, likA, priorP, priorA, lr_bcast(sc, LR_S_HASTING), lr_bcast(sc, LR_S_LOG_U), it);
        lik_p = lik;
        if (ok) A = st.pL, likA = lik, priorA = priorP, n_acc += 1;
        if (it == next_sample) {                                               // DD:221
            const int slot = trace_slot;
            trace_slot += 1;
            next_sample += (uint64_t)cfg.s_freq;
            if (slot < cfg.n_trace_slots) lr_dd_write_trace_row(a, c, lane, slot, it, likA, priorA, A);
        }
        it += 1;
    }
    // ---- propose iteration `it` ----
    lr_dd_prop p;
    const double P = lr_propose_dd<LDS_CONSTS>(a, c, lane, it, A, p, table, table_es, aux_lds);
    st.L = A, st.pL = P;
    {
        double so = 0.0;
        so = (lane == LR_S_LIKA) ? likA : so;
        so = (lane == LR_S_PRIORA) ? priorA : so;
        so = (lane == LR_S_HASTING) ? p.hasting : so;
        so = (lane == LR_S_PRIOR_P) ? p.prior : so;
        so = (lane == LR_S_LIK_P) ? lik_p : so;
        so = (lane == LR_S_LOG_U) ? p.log_u : so;
        st.sc = so;
        int io = 0;
        io = (lane == LR_I_KL || lane == LR_I_KM || lane == LR_I_PKL || lane == LR_I_PKM) ? npar : io;
        io = (lane == LR_I_IT_LO) ? (int)(uint32_t)it : io;
        io = (lane == LR_I_IT_HI) ? (int)(uint32_t)(it >> 32) : io;
        io = (lane == LR_I_ACCEPTED) ? n_acc : io;
        io = (lane == LR_I_MOVE) ? p.move : io;
        io = (lane == LR_I_NEXT_LO) ? (int)(uint32_t)next_sample : io;
        io = (lane == LR_I_NEXT_HI) ? (int)(uint32_t)(next_sample >> 32) : io;
        io = (lane == LR_I_SLOT) ? trace_slot : io;
        st.isc = io;
    }
}

// COHERENT: the partials were stored by blocks of the SAME launch (lr_stream.hip) - read at the agent's point of coherence
template <int Q, bool COHERENT = false>
__device__ __forceinline__ double lr_partials_round(const double* row, int tiles, int t0, double part) {
    double v[Q];
    if (COHERENT) {
        // unconditional loads from clamped addresses, all issued before the first is looked at (behind a branch, or with
        // the selects between them, every one of these atomic loads waits for the one before)
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        unsigned long long raw[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) raw[q] = __hip_atomic_load((gu64*)(row + min(t0 + q * LR_WAVE, tiles - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = (t0 + q * LR_WAVE < tiles) ? __longlong_as_double((long long)raw[q]) : 0.0;
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int t = t0 + q * LR_WAVE;
            v[q] = t < tiles ? row[t] : 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) part += v[q];
    return part;
}

// sum over the tiles lane, lane + 64, ... of one chain's row of partials, in ascending order
template <int Q, bool COHERENT = false>
__device__ __forceinline__ double lr_sum_tile_partials(const double* row, int tiles, int lane) {
    double part = 0.0;
    int t = lane;
    for (; t - lane + 8 * LR_WAVE < tiles; t += Q * LR_WAVE) part = lr_partials_round<Q, COHERENT>(row, tiles, t, part);
    if (t - lane < tiles) part = lr_partials_round<8, COHERENT>(row, tiles, t, part);
    return part;
}

template <int Q = 16>
__device__ __forceinline__ void lr_chain_step_body(const lr_step_args& a, int mode, int c, int lane,
                                                   lr_seg_scratch* scratch_p) {
    LR_SSTAMP(0);
    // the lane's tile partials, added in tile order.  Every round of loads is a round trip to memory (the partials come from
    // blocks on every XCD), so the rounds are made as few as possible: Q loads in flight per lane (32 in the step kernel; 16
    // in the fused kernel, whose scan blocks 32 would cost a resident wave per SIMD), absent tiles entering as + 0.0 (a no-op
    // in the sum) - one or two rounds for the scan's <= 2048 tiles - and the chain's partials are ONE ROW (lr_tile_stride):
    // a wave's load is 512 contiguous bytes.  (History, 16 chains x ~2000 tiles, in-kernel stamps: a load-add-load chain 16 us,
    // half the scan it follows; rounds of 16, 4 and 1 loads 12; one round of 32, but tile-major - every lane's load a line
    // of its own, 2048 lines per wave - 6.3; as it is now 2.1.)
    const double part = lr_sum_tile_partials<Q>(a.partials + (size_t)c * lr_tile_stride(a.tiles), a.tiles, lane);
    lr_chain_regs st;   // (loaded behind the partials: their registers are free again by then)
    double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
    int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
    lr_chain_load(st, S, I, lane);
    const double lik_sum = lr_wave_sum(part);
    if (a.cfg.sampler != 0) lr_dd_step_core(st, a, mode, c, lane, lik_sum, lr_chain_table(a, c), lr_tab_es(a.unit, a.H));
    else lr_chain_step_core(st, a, mode, c, lane, scratch_p, lik_sum, lr_chain_table(a, c), lr_tab_es(a.unit, a.H));
    lr_chain_store(st, S, I, lane);
}

