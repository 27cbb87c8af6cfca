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
#include "lr_engine.h"
#include "lr_internal.h"
#include "lr_scan.h"
#include "lr_step.h"

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
        score = lr_wave_multiplier(R, k, ff, u, 2.0 * lr_log(mult_d), lane);
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
#define LR_STEP_WAVES (LR_SCAN_THREADS / LR_WAVE)

// chains [chain_base, chain_base + n_sub): one wave per chain, LR_STEP_WAVES chains per block
__global__ __launch_bounds__(LR_SCAN_THREADS) void lr_chain_step_kernel(lr_step_args a, int mode, int chain_base,
                                                                        int n_sub) {
    __shared__ lr_seg_scratch scratch[LR_STEP_WAVES];
    const int wave = threadIdx.x / LR_WAVE, lane = threadIdx.x & (LR_WAVE - 1);
    const int j = blockIdx.x * LR_STEP_WAVES + wave;
    if (j < n_sub) lr_chain_step_body<32>(a, mode, chain_base + j, lane, &scratch[wave]);
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
    const int tile_stride = lr_tile_stride(a.tiles);
    double* partials = const_cast<double*>(a.partials) + (size_t)f.scan_base * tile_stride;
    if (UNIT)
        lr_scan_unit_body<CB, H>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables, f.scan_n, f.chunk,
                                 partials, tile_stride);
    else
        lr_scan_fast_body<CB, H>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables, f.scan_n, f.chunk,
                                 partials, tile_stride);
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
#ifndef LR_PERSIST_MINWAVES
#define LR_PERSIST_MINWAVES 4
#endif

// One pass of the packed lineages against the pair tables in global memory: the launch-based twin of the persistent
// scan, used where the engines need the sums outside their kernels (the initial state's likelihood, LRF:224-226, and
// the lr_mcmc_time_scan hook).  Block (tile, pair): partials[2 pair .. 2 pair + 1][tile].
template <int H, bool GENERAL>
__global__ __launch_bounds__(256) void lr_pairscan_kernel(lr_packed_lineages pk, long long n8, const double2* __restrict__ tables,
                                                          int n_chains, int n_bins, int tiles, double* __restrict__ partials) {
    constexpr int ENT = GENERAL ? 2 : 1;                       // double2 per pair entry in global memory
    __shared__ double2 tab[LR_UNIT_PLANES * H];                       // S, E (and their slopes) + the pair planes (lr_scan.h)
    __shared__ double red[4][2];
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int tile = blockIdx.x, pair = blockIdx.y;
    const double2* src = tables + (size_t)pair * (2 * H * ENT);
    for (int i = tid; i < 2 * H * ENT; i += 256) tab[GENERAL ? lr_pairgen_lds_entry(i, H) : i] = src[i];
    __syncthreads();
    if (GENERAL) lr_pair_planes_block_general(tab, H, n_bins, tid, 256);
    else lr_pair_planes_block(tab, H, n_bins, tid, 256);
    __syncthreads();
    const long long per = (n8 + tiles - 1) / tiles;
    const long long g0 = min((long long)tile * per, n8), g1 = min(g0 + per, n8);
    double acc0 = 0.0, acc1 = 0.0;
    lr_persist_scan<H, GENERAL>(reinterpret_cast<const char*>(tab), pk, g0, g1 - g0, tid, 256, &acc0, &acc1);
    const double s0 = lr_wave_sum(acc0), s1 = lr_wave_sum(acc1);
    if (lane == 0) red[wave][0] = s0, red[wave][1] = s1;
    __syncthreads();
    if (tid < 2 && 2 * pair + tid < n_chains)
        partials[(size_t)(2 * pair + tid) * lr_tile_stride(tiles) + tile] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// The chain step of the persistent kernels lives in functions of its own (lr_persist4_steppers, lr_persist2_steppers:
// the stepper waves' whole launch; lr_persist_step: one step, for the diagnostic build): its ~120 live registers then do
// not add to the scan loop's, and both fit the 128-VGPR budget of 4 waves per SIMD.
// Every pointer is LDS-typed: the argument block (copied to LDS once per launch; reading it through the generic pointer
// to global memory costs an L2 round trip per field), the per-wave scratch, the state rows and the pair table.
#ifndef LR_P4_STEPPER_DRAWS
#define LR_P4_STEPPER_DRAWS 0   /* helper form, 1: a stepper makes the draws of its chain's NEXT step itself, behind its step
                                   (measured: cfg4 6.66 us per iteration against 6.36 - EXPERIMENTS.md, round 5) */
#endif
#ifndef LR_P4_LAST_SUMS
#define LR_P4_LAST_SUMS 1    /* four-chain kernel: the last scanner wave to finish reduces the block's scan sums */
#endif
#ifndef LR_P4_DRAW_AHEAD
#define LR_P4_DRAW_AHEAD 1   /* four-chain kernel: the RJ sampler's draws are made one phase ahead by scanner waves */
#endif
typedef __attribute__((address_space(3))) double lr_lds_f64;
typedef __attribute__((address_space(3))) int lr_lds_i32;
// PB: bins per lane of the one-pass table builder for the kernel's table size (0: choose at run time), ES: the builders'
// `so` - both known to the calling kernel at compile time, so the builder dispatch and the layout switches fold away
// SAMPLER: -1 = both chain steps compiled in, chosen at run time (two-chain kernel); 0 = the RJ sampler's only; 1 = the
// parametric samplers' only (four-chain kernel: the launch picks the instantiation - a stepper function that carries one
// step is a third smaller, and the kernel's instruction footprint is shared by two CUs' instruction cache)
// HAND: the proposal's tables are built by a helper wave from the segments this wave hands over (`hand`, epoch `hand_epoch`)
template <int PB, int ES, bool PRE = false, int SAMPLER = -1, bool HAND = false>
__device__ __forceinline__ void lr_persist_step_body(const __attribute__((address_space(3))) lr_step_args* a3, int c, int lane,
                                                     __attribute__((address_space(3))) lr_seg_scratch* scratch3,
                                                     lr_lds_f64* st_f64, lr_lds_i32* st_i32, double lik, lr_lds_f64* table3,
                                                     lr_lds_f64* br3 /* [2][LR_H_WIDE]: br_length, log br_length */,
                                                     const lr_draw_slot* draws = nullptr /* of the iteration proposed now, made ahead */,
                                                     lr_table_hand* hand = nullptr, int hand_epoch = 0) {
    constexpr int table_es = ES;
    const lr_step_args& a = *(const lr_step_args*)a3;
    const double* br_lds = (const double*)br3;
    LR_SSTAMP(0);
    lr_chain_regs st;
    lr_chain_load(st, (double*)st_f64, (int*)st_i32, lane);
    if (SAMPLER == 1 || (SAMPLER == -1 && a.cfg.sampler != 0))
        lr_dd_step_core<true>(st, a, 0, c, lane, lik, reinterpret_cast<double2*>((double*)table3), table_es, br_lds);
    else if (PRE) {
        lr_rj_draws pre;
        lr_draws_load(draws, pre, lane);
        lr_chain_step_core<true, PB, HAND>(st, a, 0, c, lane, (lr_seg_scratch*)scratch3, lik, reinterpret_cast<double2*>((double*)table3),
                                           table_es, br_lds, br_lds + LR_H_WIDE, &pre, hand, hand_epoch);
    } else
        lr_chain_step_core<true, PB>(st, a, 0, c, lane, (lr_seg_scratch*)scratch3, lik, reinterpret_cast<double2*>((double*)table3),
                                     table_es, br_lds, br_lds + LR_H_WIDE);
    if (HAND) lr_chain_store_handed(st, (double*)st_f64, (int*)st_i32, lane);
    else lr_chain_store(st, (double*)st_f64, (int*)st_i32, lane);
    LR_SSTAMP(8);
}

// The four-chain kernel's chain step that speculates on rejection (lr_chain_step_respec): the chain's two scratch / hand-over /
// draw slots (parity of the iteration a proposal is for) and its staged candidate Q live in LDS beside the state rows.
template <int PB>
__device__ __forceinline__ void lr_persist_step_respec(const __attribute__((address_space(3))) lr_step_args* a3, int c, int lane,
                                                       __attribute__((address_space(3))) lr_seg_scratch* scratch2,
                                                       lr_lds_f64* st_f64, lr_lds_i32* st_i32, double lik, lr_lds_f64* br3,
                                                       const lr_draw_slot* draws2, lr_table_hand* hands2, lr_pend* pend2, bool first,
                                                       int epoch) {
    const lr_step_args& a = *(const lr_step_args*)a3;
    const double* br_lds = (const double*)br3;
    LR_SSTAMP(0);
    lr_chain_step_respec<PB>(a, c, lane, lik, (double*)st_f64, (int*)st_i32, pend2, (lr_seg_scratch*)scratch2, hands2, draws2, first,
                             epoch, br_lds, br_lds + LR_H_WIDE);
    LR_SSTAMP(8);
}

// End of a wave's share of scan number `scans_done` in the four-chain kernel: the lanes' sums into slot `slot` of
// `part`, count in, and - the wave that arrives LAST of the NA scanning waves - add the block's sums up, per lane over the
// slots in slot order, then across the lanes (the same order whoever is last), into out[0..1].
template <int NA>
__device__ __forceinline__ void lr_p4_leave_sums(double2 (*part)[LR_WAVE], int* arrived, double* out, int slot, int lane,
                                                 int& scans_done, double s0, double s1) {
    part[slot][lane] = make_double2(s0, s1);
    int prev = 0;
    if (lane == 0) prev = __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev == NA * (scans_done + 1) - 1) {
        asm volatile("" ::: "memory");     // the sums are read after the count was seen (a wave's LDS operations execute in order)
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int w = 0; w < NA; ++w) {
            const double2 v = part[w][lane];
            a0 += v.x, a1 += v.y;
        }
        a0 = lr_wave_sum(a0), a1 = lr_wave_sum(a1);
        if (lane == 0) out[0] = a0, out[1] = a1;
    }
    ++scans_done;
}

template <int PB, int ES>
__device__ __attribute__((noinline)) void lr_persist_step(const __attribute__((address_space(3))) lr_step_args* a3, int c, int lane,
                                                          __attribute__((address_space(3))) lr_seg_scratch* scratch3,
                                                          lr_lds_f64* st_f64, lr_lds_i32* st_i32, double lik, lr_lds_f64* table3,
                                                          lr_lds_f64* br3) {
    lr_persist_step_body<PB, ES>(a3, c, lane, scratch3, st_f64, st_i32, lik, table3, br3);
}

// ---- FLOW (-DLR_P4_FLOW=1; helper form): the phases of the four-chain kernel without their block-wide barrier -------------
// In-kernel stamps of the step-less kernel (profiles/r05_p4_wave_stamps.txt): the twelve scanner waves of a phase start
// together behind the barrier, the oldest are done after 1.9-2.2 us, the youngest after 2.5-2.65, the wave that arrives last
// adds the block's sums (~0.3 us), then 0.3-0.5 us of barrier - about 1 us of a 3.15 us phase is waiting for one another.
// Here every role runs on what it really needs:
//   a scanning wave starts the scan of pair p's proposal q when both columns of its table stand (tab_epoch[p] >= 2 (q + 1));
//   the wave that arrives LAST at the end of that scan (arrived[p] == NA (q + 1)) adds the sums and publishes sums_epoch[p] = q + 1;
//   a stepper decides proposal q of its chain when sums_epoch[p] >= q + 1; its helper builds proposal q + 1's column behind
//   the hand-over and bumps tab_epoch[p].
// No wait can cycle: a wave finishes scan (p, q) before it waits for (p, q + 1), whose table needs the step that needs the
// sums of (p, q).  The sums are added in slot order by whoever is last: the same doubles as with barriers.  A wait that
// does not end within ~0.25 s raises the engine's status word and ends the launch (every other wait then ends too).
#ifndef LR_P4_FLOW
#define LR_P4_FLOW 0
#endif
struct lr_p4_flow_lds {
    double2 part[2][14][LR_WAVE];                                // [pair][scanning wave][lane]: the lanes' sums of the pair's current scan
    int arrived[2], sums_epoch[2], tab_epoch[2];
    int abort, pad_;
};
__device__ __forceinline__ bool lr_flow_wait(const int* flag, int target, int* abort) {
    for (unsigned int spins = 0;; ++spins) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
        if (spins > (1u << 21) || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    return true;
}

// The stepper waves' whole launch in the four-chain kernel as ONE call: a call per step costs the callee-saved
// registers' round trip through scratch memory every time (24 dwords x 64 lanes out and back: 0.5 + 0.2 us of a 3.6 us
// step, measured with in-kernel stamps), a call per launch costs it once.  The function keeps the step's own register
// allocation (the reason the step is not inlined into the kernel) and executes the same barriers as the scanner
// waves' loop in lr_persist4_kernel.
//   st_f64 / st_i32: the four chains' state rows; red: [pair][wave][chain of the pair] scan sums; tab: the two pair
//   tables, tab_doubles apart
//   HELP: waves 2, 3 are helper waves - hands[wave] is this stepper's hand-over to wave 2 + wave (lr_persist4_kernel)
//   SPEC: the step speculates on rejection (lr_chain_step_respec) - scratch3 / draws / hands are then [chain of the block][parity]
//   and pend [chain of the block][parity]
template <int PB, int ES, int NW, int SAMPLER, bool HELP, bool SPEC = false, bool FLOW = false>
__device__ __attribute__((noinline)) void lr_persist4_steppers(const __attribute__((address_space(3))) lr_step_args* a3, int c0,
                                                               int n_chains, int wave, int lane,
                                                               __attribute__((address_space(3))) lr_seg_scratch* scratch3,
                                                               lr_lds_f64* st_f64, lr_lds_i32* st_i32, lr_lds_f64* red,
                                                               lr_lds_f64* tab, int tab_doubles, lr_lds_f64* br3, long long n_iters,
                                                               const lr_draw_slot* draws /* [4]: made ahead (RJ sampler) */,
                                                               lr_table_hand* hands /* [2] */, lr_pend* pend = nullptr,
                                                               lr_p4_flow_lds* fl = nullptr) {
    static_assert(!HELP || (LR_P4_DRAW_AHEAD != 0 && ES == 2 && SAMPLER == 0), "helper waves: RJ sampler at unit resolution");
    static_assert(!SPEC || HELP, "speculation on rejection: the form with helper waves");
    for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
#ifdef LR_DIAG
            const unsigned long long dq0 = wall_clock64();
#endif
            const int c = c0 + 2 * ph + wave;
            // FLOW: the sums of this pair's scan number `iter` instead of a barrier
            if (FLOW && !lr_flow_wait(&fl->sums_epoch[ph], (int)iter + 1, &fl->abort)) return;
#ifdef LR_P4_NOSTEP
            if (false) {        // (timing experiment: the phases without their chain steps - the scan alone; results are void)
#else
            if (c < n_chains) {
#endif
                double lik = 0.0;
#pragma unroll
                for (int w2 = 2; w2 < (LR_P4_LAST_SUMS != 0 && ES == 2 /* unit resolution: the block's sums in slot 2 */ ? 3 : NW); ++w2) lik += red[(ph * NW + w2) * 2 + wave];
                if (SPEC) {
                    const int ch = 2 * ph + wave;
                    lr_persist_step_respec<(PB > 0 ? PB : 1)>(a3, c, lane, scratch3 + 2 * ch, st_f64 + ch * (LR_STATE_ROWS * LR_ROW),
                                                st_i32 + ch * (LR_ISTATE_ROWS * LR_ROW), lik, br3, draws + 2 * ch, hands + 2 * ch,
                                                pend + 2 * ch, iter == 0, (int)((2 * iter + ph + 1) & 0x3fffffff));
                } else
                lr_persist_step_body<PB, ES, LR_P4_DRAW_AHEAD != 0 && ES == 2 /* unit resolution */ && SAMPLER == 0, SAMPLER, HELP>(
                    a3, c, lane, scratch3, st_f64 + (2 * ph + wave) * (LR_STATE_ROWS * LR_ROW),
                    st_i32 + (2 * ph + wave) * (LR_ISTATE_ROWS * LR_ROW), lik, tab + ph * tab_doubles + wave, br3,
                    draws + (2 * ph + wave), HELP ? hands + wave : nullptr, (int)((2 * iter + ph + 1) & 0x3fffffff));
                if (HELP && !SPEC && LR_P4_STEPPER_DRAWS) {
                    // (experiment, off) The state-independent draws of this chain's NEXT step (iteration IT + 1, used two
                    // phases on) by the stepper itself, which is done 1.7 us into a 3.15 us phase: on the scanner waves 4, 5
                    // the Philox call sits BEHIND their scans (the four-chain kernel with its steps compiled out: 3.53 us per
                    // phase with the draws, 2.97 without).  Same (iteration, purpose, index) addresses: the stream does not
                    // change - and cfg4 runs 5 % SLOWER: more work on the steppers' SIMDs, once more.
                    const int* Ic = (const int*)(st_i32 + (2 * ph + wave) * (LR_ISTATE_ROWS * LR_ROW)) + LR_IROW_SCALARS * LR_ROW;
                    LR_WAVE_LDS_ORDER();
                    const unsigned long long itn = ((unsigned long long)(unsigned int)Ic[LR_I_IT_HI] << 32 | (unsigned int)Ic[LR_I_IT_LO]) + 1ull;
                    lr_spec_draw_both(*(const lr_step_args*)a3, c, lane, itn, const_cast<lr_draw_slot*>(draws) + (2 * ph + wave));
                }
            }
#ifdef LR_DIAG
            const unsigned long long dq1 = wall_clock64();
#endif
            if (!FLOW) __syncthreads();
#ifdef LR_DIAG
            if (lane == 0 && blockIdx.x < 64) {
                atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 0], dq1 - dq0);
                atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 1], wall_clock64() - dq1);
            }
#endif
        }
    }
}

// The same for the two-chain kernel, whose waves 0 and 1 scan their share like every other wave and then step: their
// whole launch - scan, reduction, barrier, step, barrier - is one call.
template <int H, int T, int PB>
__device__ __attribute__((noinline)) void lr_persist2_steppers(const __attribute__((address_space(3))) lr_step_args* a3, int c, int wave,
                                                               int lane, __attribute__((address_space(3))) lr_seg_scratch* scratch3,
                                                               lr_lds_f64* st_f64, lr_lds_i32* st_i32, lr_lds_f64* red /* [T/64][2] */,
                                                               lr_lds_f64* tab, lr_lds_f64* br3, const uint4* __restrict__ idx8,
                                                               long long n8w, long long n_iters, int prio_shift, int grp) {
    const int tid = wave * LR_WAVE + lane;
    for (long long iter = 0; iter < n_iters; ++iter) {
        if (prio_shift > 0) {
            if (((wall_clock64() >> prio_shift) + grp) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        double acc0 = 0.0, acc1 = 0.0;
        lr_scan_tail tail;
        lr_persist_scan_pair<H, 1, true>((const char*)(const double*)tab, idx8, n8w, tid, T, &acc0, &acc1, nullptr, &tail);
        const double s0 = lr_wave_sum(acc0), s1 = lr_wave_sum(acc1);
        if (lane == 0) red[wave * 2 + 0] = s0, red[wave * 2 + 1] = s1;
        lr_scan_drain(tail);
        __syncthreads();
        double lik = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < T / LR_WAVE; ++w2) lik += red[w2 * 2 + wave];
        lr_persist_step_body<PB, 2>(a3, c, lane, scratch3, st_f64, st_i32, lik, tab + wave, br3);
        __syncthreads();
    }
}

// T = threads per block: 512 (two blocks share a CU) or, when there are no more blocks than CUs anyway (at most 512
// chains), 1024 - all 16 waves of the CU scan the one pair.
template <int H, int T>
__global__ __launch_bounds__(T, LR_PERSIST_MINWAVES) void lr_persist_kernel(
    const lr_step_args* __restrict__ ap /* in global memory: taking the address of a by-value kernel argument would
                                           copy it to scratch */,
    const uint4* __restrict__ idx8, long long n8, lr_p4_shares sh, long long n_iters, int prio_shift) {
    const lr_step_args& a = *ap;
    __shared__ double2 tab[LR_UNIT_PLANES * H];  // the pair table: S' entries [0,H), E' [H,2H), pair sums behind; (.x, .y) = (chain 0, chain 1)
    __shared__ double red[T / LR_WAVE][2];
    __shared__ lr_seg_scratch scratch[2];
    // the two chains' state rows live in LDS between iterations (registers are needed by the step itself)
    __shared__ double st_f64[2][LR_STATE_ROWS * LR_ROW];
    __shared__ int st_i32[2][LR_ISTATE_ROWS * LR_ROW];
    __shared__ lr_step_args a_lds;
    __shared__ double br_lds[2][LR_H_WIDE];   // per-bin data constants of the table builders: br_length / DT / TREND, log br_length
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    if (tid < (int)(sizeof(lr_step_args) / 4)) reinterpret_cast<int*>(&a_lds)[tid] = reinterpret_cast<const int*>(ap)[tid];
    for (int b = tid; b < LR_H_WIDE; b += blockDim.x) {
        const bool in = b < ap->cfg.n_bins;
        br_lds[0][b] = (in && ap->br_length) ? ap->br_length[b] : 0.0;
        br_lds[1][b] = in ? ap->log_br[b] : 0.0;
    }
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
    lr_pair_planes_block(tab, H, a.cfg.n_bins, tid, T);
    __syncthreads();
    const char* lbase = reinterpret_cast<const char*>(tab);
    // unequal shares of the waves (see lr_persist4_kernel): older waves 0..3 take trips from the younger 4..7
    const long long n8w = sh.delta[0] != 0 ? ((n8 + T - 1) / T + sh.delta[wave]) * T : n8;
    const int grp = (blockIdx.x >> 8) & 1;
#ifdef LR_DIAG
    unsigned long long d_t0 = 0, d_t1 = 0, d_t2 = 0, d_scan = 0, d_red = 0, d_step = 0;
#endif
#ifndef LR_DIAG
    if (stepper) {
        lr_persist2_steppers<H, T, (H <= 264 ? lr_bins_per_lane(H) : 0)>(
            (const __attribute__((address_space(3))) lr_step_args*)&a_lds, c, wave, lane,
            (__attribute__((address_space(3))) lr_seg_scratch*)&scratch[wave], (lr_lds_f64*)st_f64[wave], (lr_lds_i32*)st_i32[wave],
            (lr_lds_f64*)&red[0][0], (lr_lds_f64*)reinterpret_cast<double*>(tab), (lr_lds_f64*)&br_lds[0][0], idx8, n8w, n_iters,
            prio_shift, grp);
    } else
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
        lr_scan_tail tail;
        lr_persist_scan_pair<H, 1, true>(lbase, idx8, n8w, tid, T, &acc0, &acc1, nullptr, &tail);
#ifdef LR_DIAG
        if (lane == 0 && blockIdx.x < 512) atomicAdd(&lr_diag_step[20000 + blockIdx.x * 8 + wave], wall_clock64() - d_t0);
#endif
#ifdef LR_DIAG
        d_t1 = wall_clock64();
#endif
        const double s0 = lr_wave_sum(acc0), s1 = lr_wave_sum(acc1);
        if (lane == 0) red[wave][0] = s0, red[wave][1] = s1;
        lr_scan_drain(tail);      // the idle prefetch of the scan's last trip (lr_scan.h)
        __syncthreads();  // every scan is done: sums visible, table free to be rebuilt
#ifdef LR_DIAG
        d_t2 = wall_clock64();
#endif
        if (stepper) {
            double lik = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < T / LR_WAVE; ++w2) lik += red[w2][wave];
            lr_persist_step<(H <= 264 ? lr_bins_per_lane(H) : 0), 2>(
                (const __attribute__((address_space(3))) lr_step_args*)&a_lds, c, lane,
                (__attribute__((address_space(3))) lr_seg_scratch*)&scratch[wave], (lr_lds_f64*)st_f64[wave],
                (lr_lds_i32*)st_i32[wave], lik, (lr_lds_f64*)(reinterpret_cast<double*>(tab) + wave), (lr_lds_f64*)&br_lds[0][0]);
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
#ifndef LR_P4_THREADS
#define LR_P4_THREADS 1024
#endif
#ifndef LR_P4_UNROLL
#define LR_P4_UNROLL 1
#endif
#ifndef LR_P4_AGE_PRIO
#define LR_P4_AGE_PRIO 0
#endif
#ifndef LR_P4_DRAW_WAVE
#define LR_P4_DRAW_WAVE 2    /* SPEC form: first of the two scanner slots (wave 4 + slot) that make the draws ahead */
#endif
#ifdef LR_DIAG
#define LR_PSTAMP(k) if (threadIdx.x == 0 && blockIdx.x < 64) lr_diag_step[28672 + blockIdx.x * 8 + (k)] = wall_clock64()
#else
#define LR_PSTAMP(k)
#endif

// HELP (RJ sampler at unit resolution): waves 2, 3 - the oldest waves of the two SIMDs that carry no stepper - are HELPER
// waves and twelve waves scan.  A stepper hands the segments of its proposal over through LDS as soon as they stand
// (lr_propose_rj's HAND, ~1.2 us into a 3.2 us step) and goes on with guard, prior and state; its helper builds the table
// with its pair planes meanwhile: the serial path of a phase is load + move + staging + table instead of the whole step,
// and every SIMD carries one wave of the step and three scanners.  Before the hand-over arrives a helper scores the first
// groups of the scan (sh.help_trips trips of the 128 helper lanes; the scanners stride over the rest), and the draws of the
// OTHER pair's next step are one Philox call on each of the two oldest scanner waves (four waves, a call each, otherwise).
// dynamic LDS of the SPEC form
struct lr_p4_spec_lds {
    lr_seg_scratch scratch[8];
    lr_draw_slot draws[8];
    lr_pend pend[8];
    lr_table_hand hands[8];
};

#define LR_P4_SPEC_LDS_BYTES ((sizeof(lr_p4_spec_lds) + 255) / 256 * 256)      /* the FLOW arrays follow the SPEC ones in dynamic LDS */

// SPEC (with HELP): the steppers speculate on REJECTION (lr_chain_step_respec, lr_step.h): the proposal a chain makes next
// if its pending one is rejected is staged one iteration early, so that a helper starts the table build at the DECISION
// (~0.4 us into a phase) instead of after move + staging (~1.2 us); scratch, hand-over and draw slots are then per chain
// and parity of the iteration, the draws are made two iterations ahead.
template <int H, bool GENERAL, bool PARAM /* a parametric sampler's chain step (DDRate, trend_rate) instead of the RJ sampler's */,
          bool HELP = false, bool SPEC = false>
__global__ __launch_bounds__(LR_P4_THREADS, LR_P4_THREADS / 256) void lr_persist4_kernel(const lr_step_args* __restrict__ ap /* in global memory: a by-value argument struct measured
                                                                       0.2 us per launch faster, but one instantiation then kept a copy of it in scratch
                                                                       memory and read its fields from there in every phase */,
                                                                       lr_packed_lineages pk, long long n8,
                                                                       lr_p4_shares sh, long long n_iters, char* carry_all) {
    static_assert(!HELP || (!GENERAL && !PARAM), "helper waves: RJ sampler at unit resolution");
    static_assert(!SPEC || HELP, "speculation on rejection: the form with helper waves");
    const lr_step_args& a = *ap;
    LR_PSTAMP(0);      // entry (LR_DIAG: wall-clock stamps of a launch's stages, blocks < 64; scratch/diag_p4_launch.py)
    constexpr int NW = LR_P4_THREADS / LR_WAVE;          // 16 waves: 2 steppers + 14 scanners (HELP: 2 + 2 helpers + 12)
    constexpr int W0 = HELP ? 4 : 2;                      // first scanner wave
    constexpr int NS = NW - W0;                           // scanner waves
    constexpr int LR_P4_SCANNERS = NS * LR_WAVE;
    constexpr int ENT = GENERAL ? 2 : 1;                  // double2 per pair-table entry (LR_TAB_PAIRGEN / LR_TAB_UNIT)
    constexpr int ES = GENERAL ? 6 * H : 2;               // the builders' `so`: doubles from a value to its slope in the LDS image
    __shared__ double2 tab[2][LR_UNIT_PLANES * H];        // pair tables: S, E (general times: and their slopes) + the pair planes
    __shared__ double red[2][NW][2];                      // [pair][scanner wave][chain of the pair]; LAST_SUMS: slot [2] = the block's sums
    // Unit resolution: every scanner lane's two accumulators of the current scan, and the count of scanner waves that
    // have left theirs: the wave that arrives LAST adds them up (one wave's reduction instead of fourteen: the kernel is
    // bound by instruction issue) - per lane over the waves in wave order, then across the lanes: the same order whoever
    // is last.  (On general times the scans are the longer side of a phase and the serial tail costs more than it saves.)
    constexpr bool LAST_SUMS = LR_P4_LAST_SUMS != 0 && !GENERAL;
    __shared__ double2 part[LAST_SUMS ? NW - 2 : 1][LR_WAVE];
    __shared__ int arrived;
    // SPEC: hand-over, scratch, draw slots [chain][parity of the iteration the proposal is for] and the staged candidates live in
    // DYNAMIC LDS (lr_p4_spec_lds, behind everything static): the pair tables must stay where a ds_read's 16-bit offset
    // field reaches them - placed behind 45 KB more of static arrays their base no longer folded into the gathers, and the
    // scan loop grew eight address adds per trip
    __shared__ lr_table_hand hands_s[SPEC ? 1 : 2];
    __shared__ lr_seg_scratch scratch_s[SPEC ? 1 : 2];
    extern __shared__ double2 p4_dyn[];
    lr_p4_spec_lds* const xs = reinterpret_cast<lr_p4_spec_lds*>(p4_dyn);
    lr_table_hand* const hands = SPEC ? xs->hands : hands_s;
    lr_seg_scratch* const scratch = SPEC ? xs->scratch : scratch_s;
    lr_pend* const pend = xs->pend;
    constexpr bool FLOW = LR_P4_FLOW != 0 && HELP;                   // (the phases without their barrier: lr_p4_flow_lds)
    lr_p4_flow_lds* const fl = reinterpret_cast<lr_p4_flow_lds*>(reinterpret_cast<char*>(p4_dyn) + (SPEC ? LR_P4_SPEC_LDS_BYTES : 0));
    __shared__ double st_f64[4][LR_STATE_ROWS * LR_ROW];
    __shared__ int st_i32[4][LR_ISTATE_ROWS * LR_ROW];
    __shared__ lr_step_args a_lds;
    __shared__ double br_lds[2][LR_H_WIDE];   // per-bin data constants of the table builders: br_length / DT / TREND, log br_length
    // The state-independent draws of a chain's next proposal (acceptance uniform and its logarithm, move selectors,
    // multiplier exponents and factors, the split move's beta variate), made one phase ahead by the scanner waves that
    // finish first: the steppers are the kernel's critical path (one wave's instruction stream per chain step), the
    // scanners of a phase wait 1-2 us at its barrier.  Same (iteration, purpose, index) Philox addresses as the draws
    // made inside the step: the stream does not change.  RJ sampler only; the parametric samplers draw in the step.
    __shared__ lr_draw_slot draws_s[SPEC ? 1 : 4];
    lr_draw_slot* const draws = SPEC ? xs->draws : draws_s;
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    if (tid < (int)(sizeof(lr_step_args) / 4)) reinterpret_cast<int*>(&a_lds)[tid] = reinterpret_cast<const int*>(ap)[tid];
    for (int b = tid; b < LR_H_WIDE; b += blockDim.x) {
        const bool in = b < ap->cfg.n_bins;
        br_lds[0][b] = (in && ap->br_length) ? ap->br_length[b] : 0.0;
        br_lds[1][b] = in ? ap->log_br[b] : 0.0;
    }
    // (unit resolution only: on general times the scan loops are the longer side of a phase and have nothing to spare)
    constexpr bool draw_ahead = LR_P4_DRAW_AHEAD && !GENERAL && !PARAM;
    // draw duty of scanner wave 2 + q, q < 4, for the pair `pr` that has just been scanned: part q >> 1 of chain q & 1
    // (SPEC: `ahead` iterations beyond the pending one - 2 in the loop, 1 and 2 in the prologue -, by scanner waves
    // qoff, qoff + 1, into the slot of that iteration's parity)
    // (HELP: waves 4, 5 - the oldest scanners - sit on the steppers' SIMDs and end their scans last of all scanners (in-kernel
    // stamps: 3.1 us of a 3.15 us phase against 2.2-2.9 for the others) - and still the draws cost least there: by waves 6, 7
    // (SIMDs 2, 3, beside the helpers) cfg4 ran 6.37-6.40 us per iteration against 6.25, A/B twice on one box; under SPEC, whose
    // helpers build early, the other way round: 7.05 against 6.87)
    auto draw_duty = [&](int pr, int ahead = SPEC ? 2 : 1, int qoff = SPEC ? LR_P4_DRAW_WAVE : 0, unsigned long long it_given = 0ull) {
        const int q = wave - W0 - qoff, k = q & 1, ch = 2 * pr + k;      // (the oldest scanner waves: the first to finish)
        if (!draw_ahead || q < 0 || q >= (HELP ? 2 : 4) || (int)(blockIdx.x * 4) + ch >= ap->cfg.n_chains) return;
#if defined(LR_P4_NOSTEP) && LR_P4_NOSTEP >= 2
        return;                    // (timing experiment: not even the draws)
#endif
        const int* I = st_i32[ch] + LR_IROW_SCALARS * LR_ROW;
        // (FLOW: the iteration is given - the state row of a chain may still be in its stepper's hands)
        const unsigned long long it = it_given ? it_given
                                               : ((unsigned long long)(unsigned int)I[LR_I_IT_HI] << 32 | (unsigned int)I[LR_I_IT_LO]) + (unsigned long long)ahead;
        lr_draw_slot* slot = SPEC ? &draws[2 * ch + (int)(it & 1ull)] : &draws[ch];
        // (HELP: one wave per chain, one Philox call for both parts)
        if (HELP) lr_spec_draw_both(a_lds, (int)(blockIdx.x * 4) + ch, lane, it, slot);
        else lr_spec_draw_part(a_lds, (int)(blockIdx.x * 4) + ch, lane, it, slot, q >> 1);
    };
    // a helper wave's table duty of a phase whose steppers advance pair `ph`: once the stepper has handed them over, the
    // tables of its chain of pair ph
    // (SPEC: the slots of the chain's NEW pending iteration, it0 + iter + 1 - `par`)
    auto help_duty = [&](int ph, int epoch, int par = 0) {
        const int k = wave - 2;
        const int ch = 2 * ph + k;
        // (FLOW: a column that is not built - no such chain - still counts as standing)
        auto column_stands = [&]() {
            if (FLOW) {
                LR_WAVE_LDS_ORDER();
                if (lane == 0) __hip_atomic_fetch_add(&fl->tab_epoch[ph], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        if ((int)(blockIdx.x * 4) + ch >= ap->cfg.n_chains) { column_stands(); return; }
#ifdef LR_P4_NOSTEP
        column_stands();
        return;
#endif
        lr_table_hand* hand = SPEC ? &hands[2 * ch + par] : &hands[k];
#ifdef LR_DIAG
        const unsigned long long dh0 = wall_clock64();
#endif
        if (FLOW) {
            // (bounded, and ended by any other wait's time-out: a stepper that gave up hands nothing over)
            for (unsigned int spins = 0; __hip_atomic_load(&hand->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch; ++spins) {
                if (spins > (1u << 21) || __hip_atomic_load(&fl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                    __hip_atomic_store(&fl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        } else
        while (__hip_atomic_load(&hand->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
#ifdef LR_DIAG
        if (lane == 0 && blockIdx.x < 64) atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 2], wall_clock64() - dh0);
#endif
        const lr_seg_scratch* sc = SPEC ? &scratch[2 * ch + par] : &scratch[k];
        const int eL = lane <= LR_KMAX ? sc->edge[0][lane] : 0, eM = lane <= LR_KMAX ? sc->edge[1][lane] : 0;
        double* tabd = reinterpret_cast<double*>(tab[ph]) + k;
        const double constP = lr_build_tables_segments<(H <= 264 ? lr_bins_per_lane(H) : 1), 2>(
            sc, eL, eM, hand->KL, hand->KM, br_lds[0], br_lds[1], a_lds.cfg.model, a_lds.cfg.n_bins, a_lds.n_cls, a_lds.H,
            reinterpret_cast<double2*>(tabd), lane, LR_TAB_UNIT, a_lds.cfg.frac_birth, a_lds.cfg.frac_death, ES, nullptr);
        LR_WAVE_LDS_ORDER();
        lr_pair_planes_wave(tabd, H, a_lds.cfg.n_bins, lane, 0);
        if (lane == 0) {
            if (SPEC) pend[2 * ch + par].sc[LR_S_CONST_P] = constP;
            else st_f64[ch][LR_ROW_SCALARS * LR_ROW + LR_S_CONST_P] = constP;
        }
        column_stands();
    };
    if (tid == 0) arrived = 0;
    if (tid < (SPEC ? 8 : 2)) hands[tid].epoch = 0;
    for (int i = tid; i < 2 * NW * 2; i += LR_P4_THREADS) (&red[0][0][0])[i] = 0.0;
    int scans_done = 0;
    // a scanning wave's end of scan number `scans_done` for pair `pr` (HELP: the helper waves score a share too - slots NS,
    // NS + 1)
    constexpr int NA = NS + (HELP ? 2 : 0);
    auto leave_sums = [&](int pr, double s0, double s1) {
        if (LAST_SUMS) {
            lr_p4_leave_sums<NA>(part, &arrived, &red[pr][2][0], wave >= W0 ? wave - W0 : NS + wave - 2, lane, scans_done, s0, s1);
        } else {
            // (-DLR_P4_LAST_SUMS=0: every scanning wave adds its own lanes up; the stepper adds the waves' sums in wave order)
            s0 = lr_wave_sum(s0), s1 = lr_wave_sum(s1);
            if (lane == 0) red[pr][wave][0] = s0, red[pr][wave][1] = s1;
        }
    };
    const int c0 = blockIdx.x * 4;
    const int C = a.cfg.n_chains;
    if (wave < 4 && c0 + wave < C) {
        const int c = c0 + wave;
        const double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        const int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        for (int r = 0; r < LR_STATE_ROWS; ++r) st_f64[wave][r * LR_ROW + lane] = S[r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) st_i32[wave][r * LR_ROW + lane] = I[r * LR_ROW + lane];
        // SPEC: the pending proposal into the slot of its iteration's parity (the rows keep the accepted side)
        if (SPEC) lr_pend_from_rows(S, I, &pend[2 * wave + (I[LR_IROW_SCALARS * LR_ROW + LR_I_IT_LO] & 1)], lane);
    }
    double2* g0 = lr_chain_table(a, c0);
    double2* g1 = lr_chain_table(a, c0 + 2);             // tables are allocated for whole groups of cb >= 4 chains
    for (int i = tid; i < 2 * H * ENT; i += LR_P4_THREADS) {
        const int e = GENERAL ? lr_pairgen_lds_entry(i, H) : i;
        tab[0][e] = g0[i], tab[1][e] = g1[i];
    }
    __syncthreads();
    LR_PSTAMP(1);      // arguments, data constants, the four chains' state rows and the two pair tables are in LDS
    if (GENERAL) {
        lr_pair_planes_block_general(tab[0], H, a.cfg.n_bins, tid, LR_P4_THREADS);
        lr_pair_planes_block_general(tab[1], H, a.cfg.n_bins, tid, LR_P4_THREADS);
    } else {
        lr_pair_planes_block(tab[0], H, a.cfg.n_bins, tid, LR_P4_THREADS);
        lr_pair_planes_block(tab[1], H, a.cfg.n_bins, tid, LR_P4_THREADS);
    }
    __syncthreads();
    LR_PSTAMP(2);      // pair planes derived
    const bool scanner = wave >= W0;
    const bool helper = HELP && (wave == 2 || wave == 3);
    const int sid = tid - W0 * LR_WAVE;
#if LR_P4_AGE_PRIO
    // (experiment) issue priorities that offset the arbiter's oldest-first rule among the scanners of a SIMD: waves 4..7
    // priority 0, 8..11 1, 12..15 2, steppers and helpers 3 - in the step-less kernel the youngest scanners end their scans
    // 0.6 us after the oldest, and a phase waits for them
    if (!scanner) __builtin_amdgcn_s_setprio(3);
    else if (wave >= 12) __builtin_amdgcn_s_setprio(LR_P4_AGE_PRIO == 2 ? 1 : 2);
    else if (wave >= 8) __builtin_amdgcn_s_setprio(LR_P4_AGE_PRIO == 2 ? 0 : 1);
    else __builtin_amdgcn_s_setprio(0);
#endif
    // The SIMD issue arbiter serves its oldest wave first: with equal shares the scanner waves of a SIMD finish one
    // after the other (5.0 / 6.4 / 7.9 / 9.5 us per phase, measured with in-kernel stamps), the youngest runs the tail
    // alone, and SIMDs 0, 1 carry the stepper waves on top.  So the waves get unequal shares (lr_p4_shares) chosen to
    // make them finish together.  lr_pack_lineages_kernel stores the moved groups where the takers keep striding, so
    // this is only a per-wave end of the loop: a fixed partition, the summation order - and with it bitwise
    // reproducibility - stays.
    // (HELP: the first sh.help_trips * 128 groups are the helper waves'; the shares apply to the scanners' region behind them)
    const long long nh_ = HELP ? (long long)sh.help_trips * (2 * LR_WAVE) : 0;
    const int k_tot = (int)((n8 - nh_ + LR_P4_SCANNERS - 1) / LR_P4_SCANNERS);
    const int k_mine = k_tot + (scanner ? sh.delta[wave - W0] : 0);
    // equal shares (short scans): the true end, so that the ragged last trip costs only the lanes that have a group
    bool any_shift = false;
#pragma unroll
    for (int q = 0; q < NS; ++q) any_shift |= sh.delta[q] != 0;
    // HELP: the first `nh` groups are the helper waves' (equal shares behind them)
    // ([0, nh): helper waves, the rest: the scanner waves)
    const long long nh = HELP ? (long long)sh.help_trips * (2 * LR_WAVE) : 0;
    const long long n8w = any_shift ? (long long)k_mine * LR_P4_SCANNERS : n8 - nh;
    auto help_scan = [&](int pr) {
        double s0 = 0.0, s1 = 0.0;
        if (nh > 0) lr_persist_scan<H, GENERAL, 1, false, false>(reinterpret_cast<const char*>(tab[pr]), pk, 0, nh, tid - 2 * LR_WAVE, 2 * LR_WAVE, &s0, &s1);
        leave_sums(pr, s0, s1);
    };
    // prologue: pair 0's pending proposal is scanned so that phase A can step it - unless the launch before this one has
    // left the sums of that very scan (its last phase scored pair 0; lr_prepare_constants clears the flag whenever the
    // state is set from outside): the same doubles, the scan uses the same partition
    static_assert(2 * NW * 8 + 4 <= LR_P4_CARRY_BYTES, "");
    char* carry = carry_all ? carry_all + (size_t)blockIdx.x * LR_P4_CARRY_BYTES : nullptr;
    const bool carried = carry && *reinterpret_cast<const int*>(carry + 2 * NW * 8) == 1;
    // (SPEC: the first stepper phases need the draws of the iterations 1 and 2 beyond the pending one for pair 0, and of
    // iteration 1 for pair 1 (phase A's scan of pair 1 adds its iteration 2): three pairs of scanner waves, a call each)
    auto prologue_draws = [&]() {
        if (SPEC) draw_duty(0, 1, 2), draw_duty(0, 2, 6), draw_duty(1, 1, 10);     // (waves 6, 7; 10, 11; 14, 15: SIMDs 2, 3)
        else if (HELP && LR_P4_STEPPER_DRAWS) draw_duty(0, 1, 0), draw_duty(1, 1, 2);    // both pairs' first steps; later ones: the steppers
        else draw_duty(0);
    };
    if (carried) {
        if (tid < 2 * NW) (&red[0][0][0])[tid] = reinterpret_cast<const double*>(carry)[tid];
        if (scanner) prologue_draws();
    } else if (scanner) {
        double s0 = 0.0, s1 = 0.0;
        lr_scan_tail tail;
        lr_persist_scan<H, GENERAL, GENERAL ? 1 : LR_P4_UNROLL, false, true>(reinterpret_cast<const char*>(tab[0]), pk, nh, n8w, sid, LR_P4_SCANNERS, &s0, &s1, nullptr, &tail);
        if (LAST_SUMS) {
            leave_sums(0, s0, s1);
        } else {
            s0 = lr_wave_sum(s0), s1 = lr_wave_sum(s1);
            if (lane == 0) red[0][wave][0] = s0, red[0][wave][1] = s1;
        }
        lr_scan_drain(tail);
        prologue_draws();
    }
    if (helper && !carried) help_scan(0);
    // (SPEC: parity of the pending iteration of this helper's two chains at the start of the launch)
    int it0_par0 = 0, it0_par1 = 0;
    if (SPEC && helper) {
        it0_par0 = st_i32[wave - 2][LR_IROW_SCALARS * LR_ROW + LR_I_IT_LO] & 1;
        it0_par1 = st_i32[2 + wave - 2][LR_IROW_SCALARS * LR_ROW + LR_I_IT_LO] & 1;
    }
    __syncthreads();
    LR_PSTAMP(3);      // prologue done: pair 0's sums (scanned or carried), the first draws
    if (FLOW) {
        // ---- the phases without their barrier (see lr_p4_flow_lds) ----
        // pair 0's proposal 0 is scored (prologue / carried), both pairs' tables of proposal 0 stand
        static_assert(!FLOW || NA == 14, "lr_p4_flow_lds holds fourteen scanning waves' sums per pair");
        if (tid == 0) {
            fl->arrived[0] = NA, fl->arrived[1] = 0, fl->sums_epoch[0] = 1, fl->sums_epoch[1] = 0;
            fl->tab_epoch[0] = 2, fl->tab_epoch[1] = 2, fl->abort = 0;
        }
        // the pending iterations of the two chains a drawing wave serves (scanner slots 0, 1: chain k of either pair)
        unsigned long long itd0 = 0, itd1 = 0;
        constexpr int DQ = SPEC ? LR_P4_DRAW_WAVE : 0;          // first of the two scanner slots that draw
        const bool drawer = scanner && wave - W0 >= DQ && wave - W0 < DQ + 2;
        if (drawer) {
            const int* I0 = st_i32[wave - W0 - DQ] + LR_IROW_SCALARS * LR_ROW;
            const int* I1 = st_i32[2 + wave - W0 - DQ] + LR_IROW_SCALARS * LR_ROW;
            itd0 = (unsigned long long)(unsigned int)I0[LR_I_IT_HI] << 32 | (unsigned int)I0[LR_I_IT_LO];
            itd1 = (unsigned long long)(unsigned int)I1[LR_I_IT_HI] << 32 | (unsigned int)I1[LR_I_IT_LO];
        }
        __syncthreads();
        // end of a wave's share of scan `q` of pair `pr`: its lanes' sums, its count - and the sums of the block by the wave
        // that arrives last (per lane over the slots in slot order, then across the lanes: the same order whoever it is)
        auto leave_flow = [&](int pr, int q, double s0, double s1) {
            const int slot = wave >= W0 ? wave - W0 : NS + wave - 2;
            fl->part[pr][slot][lane] = make_double2(s0, s1);
            int prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(&fl->arrived[pr], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = __builtin_amdgcn_readfirstlane(prev);
            if (prev == NA * (q + 1) - 1) {
                asm volatile("" ::: "memory");
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int w = 0; w < NA; ++w) {
                    const double2 v = fl->part[pr][w][lane];
                    a0 += v.x, a1 += v.y;
                }
                a0 = lr_wave_sum(a0), a1 = lr_wave_sum(a1);
                if (lane == 0) red[pr][2][0] = a0, red[pr][2][1] = a1;
                LR_WAVE_LDS_ORDER();
                if (lane == 0) __hip_atomic_store(&fl->sums_epoch[pr], q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        if (helper) {
            for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll 1
                for (int ph = 0; ph < 2; ++ph) {
                    const int pr = 1 - ph, q = (int)iter + ph;          // pair 1's scan `iter`, then pair 0's scan `iter + 1`
                    if (!lr_flow_wait(&fl->tab_epoch[pr], 2 * (q + 1), &fl->abort)) goto flow_done;
                    double s0 = 0.0, s1 = 0.0;
                    if (nh > 0) lr_persist_scan<H, GENERAL, 1, false, false>(reinterpret_cast<const char*>(tab[pr]), pk, 0, nh, tid - 2 * LR_WAVE, 2 * LR_WAVE, &s0, &s1);
                    leave_flow(pr, q, s0, s1);
                    help_duty(ph, (int)((2 * iter + ph + 1) & 0x3fffffff), SPEC ? ((ph ? it0_par1 : it0_par0) + (int)(iter & 1) + 1) & 1 : 0);
                    if (__hip_atomic_load(&fl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) goto flow_done;
                }
            }
        } else if (!scanner) {
            lr_persist4_steppers<(H <= 264 ? lr_bins_per_lane(H) : 0), ES, NW, PARAM ? 1 : 0, HELP, SPEC, true>(
                (const __attribute__((address_space(3))) lr_step_args*)&a_lds, c0, C, wave, lane,
                (__attribute__((address_space(3))) lr_seg_scratch*)&scratch[SPEC ? 0 : wave], (lr_lds_f64*)&st_f64[0][0], (lr_lds_i32*)&st_i32[0][0],
                (lr_lds_f64*)&red[0][0][0], (lr_lds_f64*)reinterpret_cast<double*>(tab[0]), 2 * LR_UNIT_PLANES * H,
                (lr_lds_f64*)&br_lds[0][0], n_iters, &draws[0], &hands[0], &pend[0], fl);
        } else {
            for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int pr = 1 - ph, q = (int)iter + ph;
                    if (!lr_flow_wait(&fl->tab_epoch[pr], 2 * (q + 1), &fl->abort)) goto flow_done;
                    double s0 = 0.0, s1 = 0.0;
                    lr_scan_tail tail;
                    lr_persist_scan<H, GENERAL, LR_P4_UNROLL, false, true>(reinterpret_cast<const char*>(tab[pr]), pk, nh, n8w, sid, LR_P4_SCANNERS, &s0, &s1, nullptr, &tail);
                    // the draws of the step that follows this scan BEFORE this wave counts as arrived (the stepper starts on the count)
                    // (SPEC: two iterations beyond the proposal just scored - what the stepper stages at this decision)
                    if (drawer) draw_duty(pr, SPEC ? 2 : 1, DQ, (pr ? itd1 : itd0) + (unsigned long long)q + (SPEC ? 2ull : 1ull));
                    leave_flow(pr, q, s0, s1);
                    lr_scan_drain(tail);
                }
            }
        }
    flow_done:
        __syncthreads();
        if (tid == 0 && fl->abort) __hip_atomic_store(ap->warn - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // status: a wait timed out
    } else
    // phase ph of an iteration: the steppers advance pair `ph`, the scanners score pair `1 - ph`
    if (helper) {
        for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
#ifdef LR_DIAG
                const unsigned long long dq0 = wall_clock64();
#endif
                help_scan(1 - ph);
                help_duty(ph, (int)((2 * iter + ph + 1) & 0x3fffffff), SPEC ? ((ph ? it0_par1 : it0_par0) + (int)(iter & 1) + 1) & 1 : 0);
#ifdef LR_DIAG
                const unsigned long long dq1 = wall_clock64();
#endif
                __syncthreads();
#ifdef LR_DIAG
                if (lane == 0 && blockIdx.x < 64) {
                    atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 0], dq1 - dq0);
                    atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 1], wall_clock64() - dq1);
                }
#endif
            }
        }
    } else if (!scanner)
        lr_persist4_steppers<(H <= 264 ? lr_bins_per_lane(H) : 0), ES, NW, PARAM ? 1 : 0, HELP, SPEC>(
            (const __attribute__((address_space(3))) lr_step_args*)&a_lds, c0, C, wave, lane,
            (__attribute__((address_space(3))) lr_seg_scratch*)&scratch[SPEC ? 0 : wave], (lr_lds_f64*)&st_f64[0][0], (lr_lds_i32*)&st_i32[0][0],
            (lr_lds_f64*)&red[0][0][0], (lr_lds_f64*)reinterpret_cast<double*>(tab[0]), 2 * LR_UNIT_PLANES * H,
            (lr_lds_f64*)&br_lds[0][0], n_iters, &draws[0], &hands[0], &pend[0]);
    else
    for (long long iter = 0; iter < n_iters; ++iter) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#ifdef LR_DIAG
            const unsigned long long dq0 = wall_clock64();
#endif
            {
                double s0 = 0.0, s1 = 0.0;
                lr_scan_tail tail;
                lr_persist_scan<H, GENERAL, GENERAL ? 1 : LR_P4_UNROLL, false, true>(reinterpret_cast<const char*>(tab[1 - ph]), pk, nh, n8w, sid, LR_P4_SCANNERS, &s0, &s1, nullptr, &tail);
                if (LAST_SUMS) {
                    leave_sums(1 - ph, s0, s1);
                } else {
                    s0 = lr_wave_sum(s0), s1 = lr_wave_sum(s1);
                    if (lane == 0) red[1 - ph][wave][0] = s0, red[1 - ph][wave][1] = s1;
                }
                lr_scan_drain(tail);      // the idle prefetch of the scan's last trip (lr_scan.h)
                if (!(HELP && !SPEC && LR_P4_STEPPER_DRAWS)) draw_duty(1 - ph);
            }
#ifdef LR_DIAG
            const unsigned long long dq1 = wall_clock64();
#endif
            __syncthreads();
#ifdef LR_DIAG
            if (lane == 0 && blockIdx.x < 64) {
                atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 0], dq1 - dq0);
                atomicAdd(&lr_diag_step[16384 + (blockIdx.x * 16 + wave) * 4 + 1], wall_clock64() - dq1);
            }
#endif
        }
    }
    LR_PSTAMP(4);      // the launch's iterations done
    if (wave < 4 && c0 + wave < C) {
        const int c = c0 + wave;
        double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
        int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
        if (SPEC) {      // the pending proposal back into the rows (the staged candidate is dropped: the next launch re-proposes)
            lr_pend_to_rows(st_f64[wave], st_i32[wave], &pend[2 * wave + (st_i32[wave][LR_IROW_SCALARS * LR_ROW + LR_I_IT_LO] & 1)], lane);
            LR_WAVE_LDS_ORDER();
        }
        for (int r = 0; r < LR_STATE_ROWS; ++r) S[r * LR_ROW + lane] = st_f64[wave][r * LR_ROW + lane];
        for (int r = 0; r < LR_ISTATE_ROWS; ++r) I[r * LR_ROW + lane] = st_i32[wave][r * LR_ROW + lane];
    }
    for (int i = tid; i < 2 * H * ENT; i += LR_P4_THREADS) {
        const int e = GENERAL ? lr_pairgen_lds_entry(i, H) : i;
        g0[i] = tab[0][e], g1[i] = tab[1][e];
    }
    // the last phase has scored pair 0's pending proposal: its sums for the next launch's first phase
    if (carry) {
        if (tid < 2 * NW) reinterpret_cast<double*>(carry)[tid] = (&red[0][0][0])[tid];
        if (tid == 0) *reinterpret_cast<int*>(carry + 2 * NW * 8) = 1;
    }
    LR_PSTAMP(5);      // state, tables and carried sums stored (issued: the stores drain behind the last instruction)
}

__global__ void lr_store_args_kernel(lr_step_args a, lr_step_args* dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = a;
}

// log(br_length) once per engine (data constant used by models 0/1)
__global__ void lr_log_br_kernel(const double* __restrict__ br, int n_bins, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_bins) out[b] = (br && br[b] > 0.0) ? lr_log(br[b]) : 0.0;
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
                                       n_bins, a.H, lr_chain_table(a, c), lane, a.unit, cfg.frac_birth, cfg.frac_death, lr_tab_es(a.unit, a.H));
        } else {
            const lr_dd_params pp = lr_dd_unpack(A);
            lr_dd_build_tables_wave(pp, a.br_length, cfg.m_birth, cfg.m_death, n_bins, a.H, lr_chain_table(a, c), lane,
                                    a.unit, cfg.frac_birth, cfg.frac_death, lr_tab_es(a.unit, a.H));
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
                                                   a.unit, cfg.frac_birth, cfg.frac_death, lr_tab_es(a.unit, a.H));
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
    if (lane == LR_S_LOG_POI) so = lr_log(poi0);
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

// The launch-based engine scans the PACKED lineages (lr_packscan.hip) where that applies and pays: always with too few chains
// for the pipelined schedule; with more, once a pass is long against the chain step it no longer hides (the packed scan runs
// at ~3e13 evals/s - the pipelined launches on ts / te at 7e12 - but serially before the step kernel)
// (p: the plan with the PAIR tables the packed scan reads - unit resolution: the launches' own; general times: the
// pair-general layout of the persistent engines, 32-bit fixed-point fractions)
static bool lr_packscan_planned(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    if (!lr_packscan_eligible(cfg, p)) return false;
    const bool general = p.unit == LR_TAB_PAIRGEN;
    const double evals = (double)cfg->n_lineages * (double)cfg->n_chains;
    // (chains that the launches on ts / te would not pipeline either: fewer than two blocks' worth - general times: up to the
    // sixteen of the wide scan)
    return cfg->engine_mode == 7 || cfg->n_chains < (general ? 17 : 32) || evals >= (general ? 1.5e8 : 3.0e8);
}

// 0: the launches scan ts / te; 1: the packed lineages in one partition; 2: ... in partitions (long scans)
static int lr_packed_mode(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    if (!lr_packscan_planned(cfg, p)) return 0;
    const char* env = getenv("LR_PACKED_PARTS");          // 1 / 2 force it (read per call: tests switch it)
    if (env && (atoi(env) == 1 || atoi(env) == 2)) return atoi(env);
    const double evals = (double)cfg->n_lineages * (double)cfg->n_chains;
    // (scratch/exp_packed_parts.py: a tie at n C = 3.2e8, two partitions 3-7 % ahead at 3.8e8 - 6.4e8, 8-16 % beyond; general
    // times cost twice the gathers per eval)
    return evals >= (p.unit == LR_TAB_PAIRGEN ? 1.75e8 : 3.5e8) ? 2 : 1;
}

// Partition layout of the engine.  LR_PARTS (default 2) independent partitions run on their own streams so
// that the ramp-up / drain of one partition's launches overlaps the other's; each partition with at least
// 2*cb chains is software-pipelined in two halves.  All boundaries are multiples of cb.
// packed (lr_packed_mode): the launches scan the packed lineages - never pipelined (one launch scores a group against all
// chains of its partition); 2: partitions on their own streams overlap one's step kernel with the other's scan, 1: one
// partition (short scans: two streams of 5-us kernels do not overlap well - 16 chains x 1e7 lineages 20.9 us per iteration in
// two partitions against 16.4 in one; x 1e8: 54.1 against 63.9)
static int lr_partition(int n_chains, int cb, bool fused_ok, int base[LR_MAX_PARTS + 1], int hA[LR_MAX_PARTS],
                        bool pipelined[LR_MAX_PARTS], int packed = 0) {
    static const int want_parts = lr_env_int("LR_PARTS", 2);
    int parts = want_parts < 1 ? 1 : (want_parts > LR_MAX_PARTS ? LR_MAX_PARTS : want_parts);
    if (packed) fused_ok = false;
    if (packed == 1) parts = 1;
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
static int lr_persist_variant(const lr_mcmc_config* cfg, const lr_scan_plan& p, int* team_k, int* team_cpb);
static int lr_device_cus();
static int lr_plan_engine(const lr_mcmc_config* cfg, lr_scan_plan* p) {
    // (general times, 9 to 16 chains: ONE pass of the sixteen-chain scan - lr_scan_wide_kernel - instead of two halves of eight)
    int rc = lr_plan_scan(cfg->n_lineages, cfg->n_chains, cfg->n_bins, cfg->model, cfg->unit_resolution, p,
                          (!cfg->unit_resolution && cfg->n_chains > 8 && cfg->n_chains <= 16 && lr_env_int("LR_ENGINE_WIDE", 1)) ? 1 : 0);
    if (rc) return rc;
    if (cfg->model == LR_MODEL_KEIDING_DEAD && cfg->n_bins <= 258) {
        // model 3 on a persistent engine: one table class with an extant block behind the death-side entries (lr_step.h),
        // so the half-stride must hold 2 (n_bins + 2) entries
        static const int fast_H[] = {40, 72, 136, 264, LR_H_WIDE};
        lr_scan_plan q = *p;
        q.n_cls = 1, q.fast = 1, q.H = 0;
        for (int h : fast_H)
            if (2 * (cfg->n_bins + 2) <= h) {
                q.H = h;
                break;
            }
        if (q.H) {
            q.unit = cfg->unit_resolution ? LR_TAB_UNIT : LR_TAB_PAIRGEN;
            q.cb = cfg->unit_resolution ? 16 : 4;
            while (q.unit == LR_TAB_UNIT && q.cb > 2 && q.cb / 2 >= cfg->n_chains) q.cb >>= 1;
            q.tab_stride = q.unit == LR_TAB_UNIT ? q.H : 2 * q.H;
            q.groups = (cfg->n_chains + q.cb - 1) / q.cb;
            if (lr_persist_variant(cfg, q, nullptr, nullptr) != 0) {
                *p = q;
                return LR_OK;
            }
            // ... or, on the same tables, under the launch-based engine scanning the packed lineages (their death-side slots
            // point extant lineages at the extant block)
            if (lr_packscan_planned(cfg, q)) {
                *p = q;
                lr_packscan_plan(cfg, p, lr_device_cus());
                return LR_OK;
            }
        }
    }
    if (!p->unit && p->fast && p->n_cls == 1) {
        // general lineage times: a persistent engine takes them in the pair-general table layout (LR_TAB_PAIRGEN) with the
        // in-bin fractions packed as 32-bit fixed point; if none applies the plan stays the launch-based engine's
        lr_scan_plan q = *p;
        q.unit = LR_TAB_PAIRGEN, q.cb = 4;
        q.groups = (cfg->n_chains + q.cb - 1) / q.cb;
        if (lr_persist_variant(cfg, q, nullptr, nullptr) != 0) {
            *p = q;
            return LR_OK;
        }
        // ... or the launch-based engine scanning the packed lineages, on the same pair-general tables
        if (lr_packscan_planned(cfg, q)) {
            *p = q;
            lr_packscan_plan(cfg, p, lr_device_cus());
            return LR_OK;
        }
    }
    if (!p->fast && cfg->model != LR_MODEL_KEIDING_DEAD && cfg->n_bins + 2 <= LR_H_WIDE && cfg->n_bins <= 64 * lr_bins_per_lane(LR_H_WIDE)) {
        // 255..512 bins: beyond the table sizes the launch-based scans are instantiated for, but the persistent kernels
        // (two- and four-chain; the packed groups address their entries with 16 bits) take one more class, H = 520;
        // the scans outside the kernels go through lr_pairscan_kernel as they do for pair-general tables
        lr_scan_plan q = *p;
        q.n_cls = 1, q.fast = 1, q.H = LR_H_WIDE;
        q.unit = cfg->unit_resolution ? LR_TAB_UNIT : LR_TAB_PAIRGEN;
        q.cb = cfg->unit_resolution ? 16 : 4;
        while (q.unit == LR_TAB_UNIT && q.cb > 2 && q.cb / 2 >= cfg->n_chains) q.cb >>= 1;
        q.tab_stride = q.unit == LR_TAB_UNIT ? q.H : 2 * q.H;
        q.groups = (cfg->n_chains + q.cb - 1) / q.cb;
        if (lr_persist_variant(cfg, q, nullptr, nullptr) != 0) {
            *p = q;
            return LR_OK;
        }
        // ... or the launch-based engine scanning the packed lineages (two pairs of H = 520 tables per block)
        if (lr_packscan_planned(cfg, q)) {
            *p = q;
            lr_packscan_plan(cfg, p, lr_device_cus());
            return LR_OK;
        }
    }
    int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
    bool pipelined[LR_MAX_PARTS];
    const bool packed = lr_packscan_planned(cfg, *p);
    const int parts = lr_partition(cfg->n_chains, p->cb, lr_fused_supported(*p), base, hA, pipelined, lr_packed_mode(cfg, *p));
    if (!pipelined[0]) {
        // the scan over the packed lineages (unit resolution), or - too few chains for two halves, opt-in - the resident
        // streaming kernel, its tiles sized to the device's block slots
        if (packed) lr_packscan_plan(cfg, p, lr_device_cus());
        else if (parts == 1 && lr_stream_eligible(cfg, *p)) lr_stream_plan(cfg, p, lr_device_cus());
        return LR_OK;
    }
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

// Compute units of the CURRENT device (the engine is planned and created with its device current: literate_amd/engine.py),
// asked every time: a process may drive several devices.  LR_DEVICE_CUS overrides it (planner tests).  Without a device
// to ask (planning on a GPU-less host: the CPU tests) an MI355X is assumed - lr_mcmc_create cannot succeed there anyway.
static int lr_device_cus() {
    const char* v = getenv("LR_DEVICE_CUS");
    if (v && atoi(v) > 0) return atoi(v);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
    return 256;
}

// Speculative team kernel (lr_spec.h): one block per CU, a chain pair - or a single chain - owned by a team of k blocks.
// Model of an iteration in microseconds, fitted to MI355X measurements over 1k..200k lineages x k = 1, 2, 4, 8 x chains per
// team (scratch/exp_teams.py, round 3: candidates with a column of their own, no-op moves copying it): with `trips` =
// groups / k / 512 scanner lanes,
//                        a team per PAIR                                  a team per CHAIN
//     unit resolution    k = 1: max(2.95, 2.93 + 0.180 trips)             k = 1: max(2.02, 1.68 + 0.135 trips)
//                        k > 1: max(3.40, 3.20 + 0.205 trips)             k > 1: max(2.78, 2.32 + 0.125 trips) + 0.05 log2(k / 2)
//     general times      k = 1: max(2.90, 2.85 + 0.434 trips)             k = 1: max(2.05, 1.53 + 0.30 trips)
//                        k > 1: max(3.45, 2.90 + 0.478 trips)             k > 1: max(2.63, 2.15 + 0.33 trips) + 0.05 log2(k / 2)
//     DDRate / trend     k = 1: max(2.60, 2.70 + 0.18 trips)              k = 1: max(2.05, 1.55 + 0.125 trips)
//                        k > 1: max(3.05, 2.70 + 0.21 trips)              k > 1: max(2.90, 2.25 + 0.112 trips)
// (a team per chain: refitted in round 4 to the one-chain form of the scan - 8-byte gathers, half the fp64 instructions -,
// scratch/exp_teams.py 1 unit | dd | general on 32 chains; LR_PLAN_CHECK=1 checks the choice on the device in use)
// (the floor is the candidate build - shorter with one chain per CU: the table is built by a helper wave on a SIMD of its
// own, a no-op move copies its table, and the scanners need no table build behind the barrier -; a team pays the exchange
// behind its last scanner; a team per chain scans for one chain what a team per pair scans for two, so it needs the CUs:
// chains x k <= CUs).  Returns the modelled time, the best team size in *k
// (0 = not applicable) and the chains per team.
static double lr_spec_model(const lr_mcmc_config* cfg, int* k_out, bool general = false, int* cpb_out = nullptr) {
    const int cus = lr_device_cus();
    *k_out = 0;
    if (cpb_out) *cpb_out = 2;
    static const int k_env0 = lr_env_int("LR_SPEC_TEAM", 0);
    // (lr_check_cfg has refused a request that is not 1, 2, 4 or 8; a malformed LR_SPEC_TEAM is ignored)
    const int k_req = cfg->team_request & 0xff, cpb_req = (cfg->team_request >> 8) & 0xff;
    const int k_env = k_req > 0 ? k_req : ((k_env0 == 1 || k_env0 == 2 || k_env0 == 4 || k_env0 == 8) ? k_env0 : 0);
    static const int cpb_env0 = lr_env_int("LR_SPEC_CPB", 0);       // 1 / 2 force the chains per team (A/B runs)
    const int cpb_env = cpb_req > 0 ? cpb_req : cpb_env0;
    const double n8 = (double)((cfg->n_lineages + LR_GRP - 1) / LR_GRP);      // groups, for lineages sorted by birth time
    double best = 1e30;
    for (int cpb = 1; cpb <= 2; ++cpb) {
        if (cpb_env != 0 && cpb != cpb_env) continue;
        const int teams = (cfg->n_chains + cpb - 1) / cpb;
        if (teams > cus) continue;
        double best_c = 1e30;
        int k_c = 0;
        for (int k = 1; k <= LR_TEAM_MAX; k *= 2) {
            if (teams * k > cus) break;
            if (k_env > 0 && k != k_env) continue;
            // (groups beyond an XCD's 4 MB of L2 stream from HBM with one group of prefetch per lane: measured 0.32 instead of
            // 0.205 us per trip at 10M lineages; round-2 sweep)
            // (... setting in gradually: 1e6 lineages on general times - 4.6 MB of groups and fractions - ran 8.9 us per
            // iteration of 16 chains where the full penalty predicted 14.7 and the planner took the launches at 15.1: the
            // round-5 sweep, profiles/r05_packed_scan.txt)
            const double over = n8 * 16.0 * (general ? 1.0 + LR_FRAC_ARRAYS : 1.0) / 4.0e6 - 1.0;
            const double sfrac = over <= 0.0 ? 0.0 : (over >= 1.0 ? 1.0 : over);
            const bool streaming = sfrac >= 0.5;
            const double slow = 1.0 + 0.55 * sfrac;
            const double trips = slow * n8 / k / (double)((cpb == 1 ? LR_SPEC_THREADS_SINGLE : LR_SPEC_THREADS) - 256);
            // (a trip that waits for its group to arrive from HBM costs what the memory round trip costs: the one-chain form
            // of the scan, which made the L2-resident trips of a team per chain cheaper in round 4, does not shorten it -
            // the round-3 slopes stay for that regime)
            const double s1 = streaming ? 0.20 : 0.135, sk = streaming ? 0.165 : 0.125;          // unit resolution, a team per chain
            const double d1 = streaming ? 0.20 : 0.125, dk = streaming ? 0.16 : 0.112;           // parametric samplers
            const double g1 = streaming ? 0.46 : 0.30, gk = streaming ? 0.46 : 0.33;             // general times
            double t;
            if (cfg->sampler != 0) {
                if (cpb == 2) t = (k == 1) ? fmax(2.60, 2.70 + 0.18 * trips) : fmax(3.05, 2.70 + 0.21 * trips);
                else t = (k == 1) ? fmax(2.05, 1.55 + d1 * trips) : fmax(2.90, 2.25 + dk * trips);
                if (general) t += ((cpb == 1 && !streaming) ? 0.17 : 0.26) * trips;
            } else if (!general) {
                if (cpb == 2) t = (k == 1) ? fmax(2.95, 2.93 + 0.180 * trips) : fmax(3.40, 3.20 + 0.205 * trips);
                else t = (k == 1) ? fmax(2.02, 1.68 + s1 * trips) : fmax(2.78, 2.32 + sk * trips) + (k == 4 ? 0.05 : (k == 8 ? 0.10 : 0.0));
            } else {
                if (cpb == 2) t = (k == 1) ? fmax(2.90, 2.85 + 0.434 * trips) : fmax(3.45, 2.90 + 0.478 * trips);
                else t = (k == 1) ? fmax(2.05, 1.53 + g1 * trips) : fmax(2.63, 2.15 + gk * trips) + (k == 4 ? 0.05 : (k == 8 ? 0.10 : 0.0));
            }
            if (t < best_c - 0.05) best_c = t, k_c = k;
        }
        if (k_c > 0 && best_c < best) {
            best = best_c, *k_out = k_c;
            if (cpb_out) *cpb_out = cpb;
        }
    }
    return best;
}

// Persistent engine (lr_persist_kernel): needs unit-resolution tables with byte-sized indices and an
// instantiated table size.  cfg->engine_mode 1 / 2 / 3 force the launch-based / persistent / four-chain persistent
// engine (2 and 3 still need the prerequisites); auto picks the persistent kernel unless the chains are too few for the lineage count: a block
// scans ALL lineages for its two chains, so with few chains and very long inputs the tiled launch-based scan,
// which spreads one chain group over many CUs, is faster.
// Measured iteration times (us, MI355X, scratch/exp_p2_vs_p4.py: 3k..3M lineages x 768 / 1024 / 1536 chains; refitted at
// the end of round 3, the four-chain kernel with helper waves) of the two persistent kernels that take more than 256
// chain pairs, per round of 1024 chains:
//   two-chain kernel (512-thread blocks, two per CU)   f2 = 4.8 + 0.0355 per 1000 lineages
//   four-chain kernel                                  f4 = max(5.45, 3.13 + 0.0362 per 1000 lineages)
//                                                      (general times: max(8.0, 3.8 + 0.091 per 1000 lineages), round 2)
// and a remainder of at most 512 chains costs the two-chain kernel 0.62 of a round.
static double lr_model_two_chain(const lr_mcmc_config* cfg) {
    const double kn = (double)cfg->n_lineages * 1e-3;
    const int C = cfg->n_chains, rem = C % 1024;
    const double f2 = 4.8 + 0.0355 * kn;
    return f2 * ((double)(C / 1024) + (rem == 0 ? 0.0 : (rem <= 512 ? 0.62 : 1.0)));
}
static double lr_model_four_chain(const lr_mcmc_config* cfg, bool general) {
    const double kn = (double)cfg->n_lineages * 1e-3;
    const double f4 = general ? fmax(8.0, 3.8 + 0.091 * kn) : fmax(5.45, 3.13 + 0.0362 * kn);
    return f4 * (double)((cfg->n_chains + 1023) / 1024);
}

static bool lr_persist_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    static const int env = lr_env_int("LR_PERSIST", -1);   // debugging override: 0 off, 1 on
    if (env == 0 || cfg->engine_mode == 1 || cfg->engine_mode == 6 || cfg->engine_mode == 7) return false;
    if (!p.unit || cfg->n_bins + 2 > LR_H_WIDE || p.cb < 2) return false;
    const bool general = p.unit == LR_TAB_PAIRGEN;
    // the packing holds lineage indices as int32 and the scan loops address the groups by 32-bit byte offsets
    if (cfg->n_lineages >= (1ll << 31) - 1 || lr_groups_alloc(cfg->n_lineages) * 16 >= (1ll << 32)) return false;
    if (p.H != 40 && p.H != 72 && p.H != 136 && p.H != 264 && p.H != LR_H_WIDE) return false;
    if (env == 1 || cfg->engine_mode >= 2) return true;
    const double n = (double)cfg->n_lineages, c = (double)cfg->n_chains;
    // the best persistent kernel for the shape (microseconds) against the tiled launches (measured: 7e12 evals/s at unit
    // resolution, 3e12 on general times, 15.5 us of launches per iteration + 1 us per 1024 chains of the step blocks:
    // scratch/exp_plan_small.py - with a flat 14 us, 4096 chains on short inputs went to the launches at 19.5 - 23 us
    // where the two-chain kernel runs 15.7 - 19.1)
    double t_persist = lr_model_four_chain(cfg, general);
    if (!general) t_persist = fmin(t_persist, lr_model_two_chain(cfg));
    int k = 0;
    const double t_spec = (p.H > (general ? 136 : 264)) ? 1e30 : lr_spec_model(cfg, &k, general);   // few chains: a team of CUs per pair
    if (k > 0 && t_spec < t_persist) t_persist = t_spec;
    double t_launch = n * c / (general ? 3e12 : 7e12) * 1e6 + 15.5 + c / 1024.0;
    // ... or, at unit resolution, one launch over the packed lineages for all chains + the step kernel (16 chains x 1e7 / 3e7 /
    // 1e8 lineages: 16.3 / 27.0 / 63.9 us per iteration, scan alone 9.6 / 19.7 / 55.8: round 5)
    // (general times: sixteen gathers per group and pair instead of eight)
    // (only where the launches WOULD scan the packed lineages: lr_packscan_planned)
    if (lr_packscan_planned(cfg, p)) t_launch = fmin(t_launch, 11.5 + 5.0 * c / 1024.0 + n * c / (general ? 1.5e13 : 3.1e13) * 1e6);
    return t_persist <= t_launch;
}

// Which persistent kernel: 1 = two chains per 512-thread block (lr_persist_kernel), 2 = four chains per 1024-thread
// block in ping-pong (lr_persist4_kernel), 3 = the speculative team kernel (at most 256 chain pairs).  The four-chain
// block hides the chain step under the other pair's scan, so beyond 256 pairs it wins while a scan is about as long as a
// step (from ~20k lineages up, whole rounds of 1024 chains); the two-chain kernel, two blocks per CU, is ahead on short
// scans (the step is all there is) and on remainders of at most 512 chains; on very long scans the two tie.
static int lr_persist_variant(const lr_mcmc_config* cfg, const lr_scan_plan& p, int* team_k, int* team_cpb = nullptr) {
    if (team_k) *team_k = 0;
    if (team_cpb) *team_cpb = 0;
    if (!lr_persist_eligible(cfg, p)) return 0;
    static const int p4_env = lr_env_int("LR_PERSIST4", -1);
    static const int spec_env = lr_env_int("LR_SPEC", -1);
    const bool general = p.unit == LR_TAB_PAIRGEN;   // general times: the speculative (H <= 136) and four-chain kernels only
    int k = 0, cpb = 2;
    const double t_spec = (p.H > (general ? 136 : 264)) ? 1e30 : lr_spec_model(cfg, &k, general, &cpb);
    if (k > 0 && t_spec < 1e29 && cfg->engine_mode != 3 && cfg->engine_mode != 4 && spec_env != 0) {
        // the two-chain kernel with all 16 waves scanning wins only on long scans without room for a team
        // (measured at 256 pairs, 10k..1M lineages, scratch/exp_engines.py: 4.41 + 0.236 us per trip of its 1024 scanner
        // lanes, 4.1 at least; the speculative kernel is ahead up to ~200k lineages)
        const double t_wide = general ? 1e30 : fmax(4.1, 4.41 + 0.236 * (double)((cfg->n_lineages + LR_GRP - 1) / LR_GRP) / 1024.0);
        if (cfg->engine_mode == 5 || spec_env == 1 || t_spec <= t_wide) {
            if (team_k) *team_k = k;
            if (team_cpb) *team_cpb = cpb;
            return 3;
        }
    }
    if (cfg->engine_mode == 5) return 0;
    if (general) return (p.cb >= 4 && cfg->engine_mode != 4) ? 2 : 0;
    if (p.cb < 4) return 1;                 // tables are laid out per group of cb chains; a quad must not straddle
    if (cfg->engine_mode == 3) return 2;
    if (cfg->engine_mode == 4) return 1;
    if (p4_env >= 0) return p4_env ? 2 : 1;
    return lr_model_four_chain(cfg, false) < lr_model_two_chain(cfg) ? 2 : 1;
}

static int lr_check_cfg(const lr_mcmc_config* cfg) {
    if (!cfg) return LR_ERR_NULL;
    if (cfg->n_lineages < 1 || cfg->n_chains < 1 || cfg->s_freq < 1 || cfg->n_trace_slots < 0) return LR_ERR_SIZE;
    if (cfg->n_bins < 1 || cfg->n_bins > LR_MAX_BINS) return LR_ERR_SIZE;
    if (cfg->model < 0 || cfg->model > 3) return LR_ERR_MODEL;
    if (cfg->t0 != floor(cfg->t0)) return LR_ERR_T0;
    if (!(cfg->end_time > cfg->start_time)) return LR_ERR_SIZE;
    if (cfg->sampler < 0 || cfg->sampler > 2) return LR_ERR_MODEL;
    {
        const int k_req = cfg->team_request & 0xff, cpb_req = (cfg->team_request >> 8) & 0xff;
        if (cfg->team_request < 0 || cfg->team_request > 0xffff) return LR_ERR_SIZE;
        if (k_req != 0 && k_req != 1 && k_req != 2 && k_req != 4 && k_req != 8) return LR_ERR_SIZE;   // a team is 1, 2, 4 or 8 blocks (LR_TEAM_MAX)
        if (cpb_req > 2) return LR_ERR_SIZE;                                                           // ... of one chain or a pair
    }
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
    // packed lineages of the persistent engines (lr_pack.hip): groups of LR_GRP lineages of one birth bin, 16 bytes of
    // table indices each; on general times LR_FRAC_ARRAYS more uint4 arrays with the in-bin fractions; scratch of the packing
    const long long n_alloc = lr_groups_alloc(cfg->n_lineages);
    out->lineage_idx = o, o += lr_align_up64(n_alloc * 16, 256);
    out->lineage_frac = o;
    if (p.unit == LR_TAB_PAIRGEN) o += lr_align_up64(n_alloc * 16 * LR_FRAC_ARRAYS, 256);
    out->pack_tmp = o, o += lr_align_up64(lr_pack_tmp_bytes(cfg->n_lineages), 256);
    out->args_blob = o, o += 1024;   // lr_step_args of the persistent kernel
    out->tables = o, o += lr_align_up64((long long)lr_align_up64(C, p.cb < 2 ? 2 : p.cb) * p.tab_stride * 16, 256);
    out->partials = o, o += lr_align_up64((long long)lr_tile_stride(p.tiles) * C * 8, 256);
    out->trace = o, o += lr_align_up64((long long)cfg->n_trace_slots * C * LR_TRACE_W * 8, 256);
    int team_k = 0, team_cpb = 0;
    out->persistent = lr_persist_variant(cfg, p, &team_k, &team_cpb);
    out->spec_chains_per_team = team_cpb;
    {
        int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
        bool pipelined[LR_MAX_PARTS];
        const bool packed = lr_packscan_planned(cfg, p);
        const int parts = lr_partition(cfg->n_chains, p.cb, lr_fused_supported(p), base, hA, pipelined, lr_packed_mode(cfg, p));
        out->packed_scan = (out->persistent == 0 && packed) ? 1 : 0;
        out->streaming = (out->persistent == 0 && !out->packed_scan && parts == 1 && !pipelined[0] && lr_stream_eligible(cfg, p)) ? 1 : 0;
        out->reserved3 = 0;
    }
    if (p.unit == LR_TAB_PAIRGEN && out->persistent == 0 && !lr_packscan_planned(cfg, p)) return LR_ERR_STATE;   // (planned only when a kernel takes it)
    out->team_blocks = team_k;
    out->table_mode = p.unit;
    out->status = o, o += 256;   // engine status word
    // team exchange granules of the speculative kernel: [2 parities][pairs][LR_TEAM_MAX][LR_SPEC_GRANULES] x 8 bytes
    // (four-chain kernel: the scan sums a launch leaves for the next one, LR_P4_CARRY_BYTES per block - lr_persist4_kernel)
    // (streaming engine: the counters of a launch and the second table buffer)
    out->xchg = o, o += (team_k > 1) ? lr_align_up64(2ll * C * LR_TEAM_MAX * LR_SPEC_GRANULES * 8, 256)   // (room for a team per chain)
                                     : (out->persistent == 2 ? lr_align_up64((C + 3) / 4 * LR_P4_CARRY_BYTES, 256)
                                        : (out->streaming ? LR_STREAM_SYNC_BYTES + lr_align_up64((long long)lr_align_up64(C, p.cb < 2 ? 2 : p.cb) * p.tab_stride * 16, 256) : 0));
    out->total_bytes = o;
    out->table_stride = p.tab_stride;
    out->tiles = p.tiles;
    out->chains_per_block = p.cb;
    out->trace_width = LR_TRACE_W;
    {
        int base[LR_MAX_PARTS + 1], hA[LR_MAX_PARTS];
        bool pipelined[LR_MAX_PARTS];
        out->n_parts = lr_partition(cfg->n_chains, p.cb, lr_fused_supported(p), base, hA, pipelined, lr_packed_mode(cfg, p));
        out->pipelined = pipelined[0] ? 1 : 0;
    }
    {
        static const int wide_env = lr_env_int("LR_PERSIST_WIDE", -1);
        const bool wide = wide_env >= 0 ? wide_env != 0 : ((cfg->n_chains + 1) / 2 <= 256 && cfg->n_lineages >= 20000);
        out->reserved1 = out->persistent == 3 ? (team_cpb == 1 ? LR_SPEC_THREADS_SINGLE : LR_SPEC_THREADS) : (out->persistent == 2 ? 1024 : (out->persistent == 1 ? (wide ? 1024 : 512) : 0));   // threads per persistent block
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
    e->n_parts = lr_partition(cfg->n_chains, e->plan.cb, lr_fused_supported(e->plan), base, hA, pipelined,
                              lay.packed_scan ? lr_packed_mode(cfg, e->plan) : 0);
    e->persistent = lay.persistent != 0;
    e->n8 = 0;                                          // groups of packed lineages: known once lr_pack_lineages has run
    e->n8_alloc = lr_groups_alloc(cfg->n_lineages);
    for (int j = 0; j < 16; ++j) e->p4.delta[j] = 0;
    e->p4.n_slots = 8;
    e->p4_help = lr_p4_help_choice(e);                  // (lr_mcmc_describe before init; latched again by lr_set_shares)
    e->p4_spec = lr_p4_spec_choice(e);
    e->streaming = false;
    e->packed_scan = lay.packed_scan != 0;
    e->fork = nullptr;
    e->ev0 = e->ev1 = nullptr;
    for (int p = 0; p < e->n_parts; ++p) {
        lr_part& q = e->part[p];
        q.base = base[p], q.count = base[p + 1] - base[p], q.hA = hA[p], q.pipelined = pipelined[p];
        q.stream = nullptr, q.done = nullptr, q.graph_exec = nullptr, q.graph_units = 0;
    }
    {
        // the two timing events of lr_mcmc_time_steps / lr_mcmc_time_scan live as long as the engine: creating and
        // destroying them per call cost the host tens of microseconds inside every timed region
        hipError_t he = hipEventCreate(&e->ev0);
        if (he == hipSuccess) he = hipEventCreate(&e->ev1);
        if (he != hipSuccess) {
            lr_mcmc_destroy(e);
            return (int)he;
        }
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
    if (lay.streaming) {
        // the resident kernel's blocks wait for each other: it runs only where the whole grid fits at once (LR_ERR_STATE =
        // it does not: the launches take over, same plan, same results)
        const int rcq = lr_launch_stream(e, lr_make_args(e), 0, true, nullptr);
        if (rcq != LR_OK && rcq != LR_ERR_STATE) {
            lr_mcmc_destroy(e);
            return rcq;
        }
        e->streaming = rcq == LR_OK;
    }
    *out = e;
    return LR_OK;
}

lr_step_args lr_make_args(const lr_engine* e) {
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
    a.warn = (unsigned int*)(e->ws + e->lay.status) + 1;
    return a;
}

template <int H>
static int lr_launch_pairscan(const lr_engine* e, hipStream_t stream) {
    lr_packed_lineages pk;
    pk.idx8 = (const uint4*)(e->ws + e->lay.lineage_idx), pk.frac = (const uint4*)(e->ws + e->lay.lineage_frac);
    pk.fstride = e->n8_alloc;
    const dim3 grid(e->plan.tiles, (e->cfg.n_chains + 1) / 2);
    if (e->plan.unit == LR_TAB_PAIRGEN)
        hipLaunchKernelGGL((lr_pairscan_kernel<H, true>), grid, dim3(256), 0, stream, pk, e->n8,
                           (const double2*)(e->ws + e->lay.tables), e->cfg.n_chains, e->cfg.n_bins, e->plan.tiles, (double*)(e->ws + e->lay.partials));
    else
        hipLaunchKernelGGL((lr_pairscan_kernel<H, false>), grid, dim3(256), 0, stream, pk, e->n8,
                           (const double2*)(e->ws + e->lay.tables), e->cfg.n_chains, e->cfg.n_bins, e->plan.tiles, (double*)(e->ws + e->lay.partials));
    return (int)hipGetLastError();
}

static int lr_enqueue_scan_range(const lr_engine* e, int base, int count, hipStream_t stream) {
    if (e->packed_scan) return lr_launch_packscan(e, base, count, stream);      // (never pipelined: a partition's chains in one launch)
    if (e->plan.unit == LR_TAB_PAIRGEN || (e->persistent && (e->cfg.model == LR_MODEL_KEIDING_DEAD || e->plan.H == LR_H_WIDE))) {
        // pair-general tables / the extant block of model 3: the launch-based twin of the persistent scan, all chains at once
        if (base != 0 || count != e->cfg.n_chains) return LR_ERR_STATE;
        switch (e->plan.H) {
            case 40: return lr_launch_pairscan<40>(e, stream);
            case 72: return lr_launch_pairscan<72>(e, stream);
            case 136: return lr_launch_pairscan<136>(e, stream);
            case 264: return lr_launch_pairscan<264>(e, stream);
            default: return lr_launch_pairscan<LR_H_WIDE>(e, stream);
        }
    }
    return lr_launch_scan(e->plan, e->ts, e->te, e->cfg.n_lineages, e->cfg.t0, e->cfg.n_bins, e->cfg.end_time,
                          (const double2*)(e->ws + e->lay.tables) + (size_t)base * e->plan.tab_stride, count,
                          (double*)(e->ws + e->lay.partials) + (size_t)base * lr_tile_stride(e->plan.tiles), lr_tile_stride(e->plan.tiles), stream);
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
    if (lane == 0) out[0] = m, out[1] = lr_log(m);
}

// everything in the workspace that holds device addresses or derives from the data alone
static void lr_prepare_constants(const lr_engine* e, const lr_step_args& a, hipStream_t stream) {
    // a fresh or restored run starts with a clear status and warning word (a checkpoint never carries a void run on:
    // ChainEngine.save refuses to write one)
    (void)hipMemsetAsync(e->ws + e->lay.status, 0, 256, stream);
    // ... and nothing carried over from the launches before it (four-chain kernel)
    if (e->lay.persistent == 2) (void)hipMemsetAsync(e->ws + e->lay.xchg, 0, (size_t)((e->cfg.n_chains + 3) / 4) * LR_P4_CARRY_BYTES, stream);
    hipLaunchKernelGGL(lr_log_br_kernel, dim3((e->cfg.n_bins + 127) / 128), dim3(128), 0, stream, e->br_length,
                       e->cfg.n_bins, (double*)(e->ws + e->lay.bin_consts));
    if (e->cfg.sampler == 1)
        hipLaunchKernelGGL(lr_dd_consts_kernel, dim3(1), dim3(LR_WAVE), 0, stream, e->br_length, e->cfg.n_bins,
                           (double*)(e->ws + e->lay.bin_consts) + e->cfg.n_bins);
    if (e->persistent) {
        static_assert(sizeof(lr_step_args) <= 1024, "args blob too small");
        hipLaunchKernelGGL(lr_store_args_kernel, dim3(1), dim3(64), 0, stream, a, (lr_step_args*)(e->ws + e->lay.args_blob));
    }
}

// the launch-based engine's packed scan: (re)pack; an input the packing refuses (unsorted beyond LR_MAX_RUNS runs of one
// birth bin) falls back to the scan of ts / te - same plan, same partials, same results to rounding
// (general times have no such twin: their tables are the pair-general ones - the error is the caller's
static int lr_pack_for_scan(lr_engine* e, hipStream_t stream) {
    if (!e->lay.packed_scan) return LR_OK;
    const int rc = lr_pack_lineages(e, stream);
    e->packed_scan = rc == LR_OK;
    // (... nor has model 3 on these tables: the extant block is reached through the packed slots only)
    // (... nor the H = 520 class: the scans of ts / te are instantiated up to H = 264)
    return (rc != LR_OK && (e->plan.unit == LR_TAB_PAIRGEN || e->cfg.model == LR_MODEL_KEIDING_DEAD || e->plan.H == LR_H_WIDE)) ? rc : LR_OK;
}

extern "C" int lr_mcmc_restore(lr_engine* e, void* stream_) {
    if (!e) return LR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const lr_step_args a = lr_make_args(e);
    lr_prepare_constants(e, a, stream);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    if (e->persistent) rc = lr_pack_lineages(e, stream);
    if (rc) return rc;
    rc = lr_pack_for_scan(e, stream);
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
    if (e->persistent) {
        const int rcp = lr_pack_lineages(e, stream);
        if (rcp) return rcp;
    }
    {
        const int rcp = lr_pack_for_scan(e, stream);
        if (rcp) return rcp;
    }
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
        const bool general = e->plan.unit == LR_TAB_PAIRGEN;
        lr_packed_lineages pk;
        pk.idx8 = idx8, pk.frac = general ? (const uint4*)(e->ws + e->lay.lineage_frac) : nullptr, pk.fstride = e->n8_alloc;
        if (e->n8 <= 0) return LR_ERR_STATE;
        const lr_step_args* ap = (const lr_step_args*)(e->ws + e->lay.args_blob);
        const int blocks = (e->cfg.n_chains + 1) / 2;
        const bool p4 = e->lay.persistent == 2;
        const bool param = e->cfg.sampler != 0;
        static const int prio = lr_env_int("LR_PERSIST_PRIO", 12);   // clock bits per priority slice, 0 = off
        // one block per CU at most: give it the whole CU (16 waves on the one pair)
        static const int wide_env = lr_env_int("LR_PERSIST_WIDE", -1);
        const bool wide = e->lay.reserved1 == 1024;   // (short scans keep 512: the 16-wave barrier costs more than it buys)
        (void)wide_env;
        if (e->lay.persistent == 3) return lr_launch_spec(e, a, pk, n_iters, stream);
        for (int64_t done = 0; done < n_iters;) {
            const int64_t n = (n_iters - done > 4096) ? 4096 : n_iters - done;   // keep single launches short
        static const int carry_env = lr_env_int("LR_P4_CARRY", 1);   // 0: every launch scans pair 0 again (A/B runs)
        char* p4_carry = (p4 && carry_env) ? e->ws + e->lay.xchg : nullptr;
#define LR_P_LAUNCH(HH)                                                                                                       \
    if (p4) {                                                                                                                 \
        const dim3 g4((e->cfg.n_chains + 3) / 4), b4(LR_P4_THREADS);                                                          \
        if (general && param) hipLaunchKernelGGL((lr_persist4_kernel<HH, true, true>), g4, b4, 0, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry);   \
        else if (general) hipLaunchKernelGGL((lr_persist4_kernel<HH, true, false>), g4, b4, 0, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry);      \
        else if (param) hipLaunchKernelGGL((lr_persist4_kernel<HH, false, true>), g4, b4, 0, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry);        \
        else if (e->p4_help && e->p4_spec) {                                                                                   \
            /* (helper waves: H <= 264, lr_p4_help_choice; the attribute belongs to the function on the CURRENT device) */     \
            const size_t dyn_ = LR_P4_FLOW ? LR_P4_SPEC_LDS_BYTES + sizeof(lr_p4_flow_lds) : sizeof(lr_p4_spec_lds);            \
            hipError_t he_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_persist4_kernel<(HH <= 264 ? HH : 264), false, false, true, true>), \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_);                       \
            if (he_ != hipSuccess) return (int)he_;                                                                            \
            hipLaunchKernelGGL((lr_persist4_kernel<(HH <= 264 ? HH : 264), false, false, true, true>), g4, b4, dyn_, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry); \
        }                                                                                                                      \
        else if (e->p4_help) {                                                                                                 \
            if (LR_P4_FLOW) {                                                                                                  \
                hipError_t he_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_persist4_kernel<(HH <= 264 ? HH : 264), false, false, true>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(lr_p4_flow_lds)); \
                if (he_ != hipSuccess) return (int)he_;                                                                        \
            }                                                                                                                  \
            hipLaunchKernelGGL((lr_persist4_kernel<(HH <= 264 ? HH : 264), false, false, true>), g4, b4, LR_P4_FLOW ? sizeof(lr_p4_flow_lds) : 0, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry); \
        }                                                                                                                      \
        else hipLaunchKernelGGL((lr_persist4_kernel<HH, false, false>), g4, b4, 0, stream, ap, pk, e->n8, e->p4, (long long)n, p4_carry);                  \
    } else if (wide) {                                                                                                        \
        hipLaunchKernelGGL((lr_persist_kernel<HH, 1024>), dim3(blocks), dim3(1024), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, 0); \
    } else {                                                                                                                  \
        hipLaunchKernelGGL((lr_persist_kernel<HH, 512>), dim3(blocks), dim3(512), 0, stream, ap, idx8, e->n8, e->p4, (long long)n, prio); \
    }
            switch (e->plan.H) {
                case 40: LR_P_LAUNCH(40); break;
                case 72: LR_P_LAUNCH(72); break;
                case 136: LR_P_LAUNCH(136); break;
                case 264: LR_P_LAUNCH(264); break;
                default: LR_P_LAUNCH(LR_H_WIDE); break;
            }
#undef LR_P_LAUNCH
            const int rc = (int)hipGetLastError();
            if (rc) return rc;
            done += n;
        }
        return LR_OK;
    }
    if (e->streaming) return lr_launch_stream(e, a, n_iters, false, stream);
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
    int rc = (int)hipEventRecord(e->ev0, stream);
    if (rc == LR_OK) rc = lr_mcmc_steps(e, n_iters, stream_);
    if (rc == LR_OK) rc = (int)hipEventRecord(e->ev1, stream);
    if (rc == LR_OK) rc = (int)hipEventSynchronize(e->ev1);
    float ms = 0.f;
    if (rc == LR_OK) rc = (int)hipEventElapsedTime(&ms, e->ev0, e->ev1);
    *total_ms = ms;
    return rc;
}

extern "C" int lr_mcmc_time_scan(lr_engine* e, int32_t reps, float* avg_ms, void* stream_) {
    if (!e || !avg_ms) return LR_ERR_NULL;
    if (!e->initialised) return LR_ERR_STATE;
    if (reps < 1) return LR_ERR_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    int rc = lr_enqueue_scan(e, stream);  // warm
    if (rc == LR_OK) rc = (int)hipEventRecord(e->ev0, stream);
    for (int i = 0; i < reps && rc == LR_OK; ++i) rc = lr_enqueue_scan(e, stream);
    if (rc == LR_OK) rc = (int)hipEventRecord(e->ev1, stream);
    if (rc == LR_OK) rc = (int)hipEventSynchronize(e->ev1);
    float ms = 0.f;
    if (rc == LR_OK) rc = (int)hipEventElapsedTime(&ms, e->ev0, e->ev1);
    *avg_ms = ms / reps;
    return rc;
}

extern "C" int lr_mcmc_status(lr_engine* e, int32_t* status, void* stream_) {
    if (!e || !status) return LR_ERR_NULL;
    unsigned int v = 0;
    hipError_t he = hipMemcpyAsync(&v, e->ws + e->lay.status, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream_);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream_);
    *status = (int32_t)v;
    return (int)he;
}

extern "C" int lr_mcmc_warnings(lr_engine* e, int32_t* warnings, void* stream_) {
    if (!e || !warnings) return LR_ERR_NULL;
    unsigned int v = 0;
    hipError_t he = hipMemcpyAsync(&v, e->ws + e->lay.status + 4, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream_);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream_);
    *warnings = (int32_t)v;
    return (int)he;
}

// name of the kernel lr_mcmc_steps spends its time in, as rocprofv3's kernel trace prints it (without arguments)
extern "C" int lr_mcmc_describe(const lr_engine* e, char* buf, int32_t n) {
    if (!e || !buf) return LR_ERR_NULL;
    if (n < 64) return LR_ERR_SIZE;
    if (e->persistent) {
        const char* gen = e->plan.unit == LR_TAB_PAIRGEN ? "true" : "false";
        if (e->lay.persistent == 3)
            snprintf(buf, (size_t)n, "lr_spec_kernel<%d, %d, %s, %s, %d>", e->plan.H, e->lay.reserved1, e->cfg.sampler == 0 ? "true" : "false", gen,
                     lr_spec_mode(e));
        else if (e->lay.persistent == 2)
            snprintf(buf, (size_t)n, "lr_persist4_kernel<%d, %s, %s, %s, %s>", e->plan.H, gen, e->cfg.sampler != 0 ? "true" : "false",
                     e->p4_help ? "true" : "false", (e->p4_help && e->p4_spec) ? "true" : "false");
        else snprintf(buf, (size_t)n, "lr_persist_kernel<%d, %d>", e->plan.H, e->lay.reserved1);
    } else if (e->packed_scan) {
        snprintf(buf, (size_t)n, "lr_packscan_kernel<%d, %d, %s>", lr_packscan_pairs(e->plan, e->cfg.n_chains), e->plan.H,
                 e->plan.unit == LR_TAB_PAIRGEN ? "true" : "false");
    } else if (e->streaming) {
        snprintf(buf, (size_t)n, "lr_stream_kernel<%d, %d, %s>", e->plan.cb, e->plan.H, e->plan.unit ? "true" : "false");
    } else if (e->part[0].pipelined) {
        snprintf(buf, (size_t)n, "lr_fused_iter_kernel<%d, %d, %s>", e->plan.cb, e->plan.H, e->plan.unit ? "true" : "false");
    } else if (e->plan.fast && !e->plan.unit && e->plan.cb == 16) {
        snprintf(buf, (size_t)n, "lr_scan_wide_kernel<%d>", e->plan.H);
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
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return LR_OK;
}

#ifdef LR_DIAG
extern "C" int lr_diag_dump_step(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_step), (size_t)n_words * 8);
}
extern "C" int lr_diag_dump_seg(unsigned long long* host_out, int n_words, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_seg), (size_t)n_words * 8);
    if (reset) {
        static unsigned long long zeros[64 * 16];
        rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(lr_diag_seg), zeros, sizeof(zeros));
    }
    return rc;
}
#endif
