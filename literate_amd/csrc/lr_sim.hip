// lr_sim.hip - discrete-time birth-death lineage simulator on the device.
//
// The scheme of both reference simulators (simulateRateABC.v2.py:103-234 `Simulator.simulate`, notebook 4
// `Simulator.run_simulation`): per step t every living lineage draws one uniform r; r < lambda_t spawns a lineage
// born at t, lambda_t <= r < lambda_t + mu_t kills it at t; lineages born at t first act at t + 1.  Rates per step
// are given arrays or diversity dependent (functions of the living count at the start of the step).
//
// One launch pair per step: a one-thread kernel snapshots the counters and fixes the step's rates, then a
// grid-stride kernel visits the lineages that existed at the snapshot.  Newborn slots come from one
// wave-aggregated atomic per wave; the newborns of a step are exchangeable and their draws are addressed by
// (seed, slot, step), so the multiset of (birth, death) pairs is independent of the order the slots are handed out
// (checked bit for bit against oracle/sim_oracle.py).  HBM-bound streaming over ts/te (16 B per living lineage and
// step), no LDS, no MFMA.
#include <hip/hip_runtime.h>

#include "../../include/literate_hip.h"
#include "lr_device.h"

#define LR_P_SIM 24

struct lr_sim_step {
    long long n_cur;      // lineages existing at the start of the step
    double lt, mt;        // per-step birth / death probabilities
};

// counters: [0] lineages allocated, [1] living, [2] overflow flag, [3] unused
__global__ void lr_sim_init_kernel(double* __restrict__ ts, double* __restrict__ te, long long n_start, int n_steps,
                                   long long* __restrict__ counters) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_start) ts[i] = 0.0, te[i] = (double)n_steps;
    if (i == 0) counters[0] = n_start, counters[1] = n_start, counters[2] = 0, counters[3] = 0;
}

__global__ void lr_sim_prepare_kernel(const double* __restrict__ lam_steps, const double* __restrict__ mu_steps, int t,
                                      int mode, double l0, double m0, double K, double scale, long long capacity,
                                      long long* __restrict__ counters, lr_sim_step* __restrict__ step,
                                      long long* __restrict__ alive_trace) {
    // after an overflow the allocation counter runs past the capacity (the dropped births still took slot numbers):
    // clamp it here, so that the next step never visits - or hands out - a slot outside ts/te
    if (counters[0] > capacity) counters[0] = capacity;
    const long long alive = counters[1];
    double lt, mt;
    if (mode == 0) {
        lt = lam_steps[t], mt = mu_steps[t];
    } else {
        const double D = (double)alive;
        double la, mu;
        if (mode == 1) la = fmax(0.0, l0 - l0 * D / K), mu = fmax(0.0, m0 + m0 * D / K);            // nb4 DD generator
        else la = fmax(0.0, l0 - (l0 - m0) * D / K), mu = fmax(0.0, m0 + (l0 - m0) * D / K);        // ABC:153-154
        lt = la / scale, mt = mu / scale;
    }
    step->n_cur = counters[0], step->lt = lt, step->mt = mt;
    if (alive_trace) alive_trace[t] = alive;
}

// One block works in rounds of 64 items per thread: births are first only recorded (one bit per item), then the block
// takes ONE contiguous range of newborn slots (block scan + one atomic) and ONE update of the living count - two
// atomics per block and round instead of one per wave and trip on the same two addresses.
__global__ __launch_bounds__(256) void lr_sim_step_kernel(double* __restrict__ ts, double* __restrict__ te, int t,
                                                          int n_steps, long long capacity, unsigned long long seed,
                                                          const lr_sim_step* __restrict__ step,
                                                          long long* __restrict__ counters) {
    __shared__ int wave_tot[4];
    __shared__ long long base_s;
    const long long n_cur = step->n_cur;
    const double lt = step->lt, mt = step->mt;
    const double extant = (double)n_steps;
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long first = (long long)blockIdx.x * blockDim.x + tid;
    for (long long r0 = 0; r0 < n_cur; r0 += 64 * stride) {     // uniform trip count for the whole grid
        unsigned long long births = 0ull;
        int n_birth = 0, n_death = 0;
        for (int k = 0; k < 64; ++k) {
            const long long i = r0 + (long long)k * stride + first;
            if (i >= n_cur) break;
            if (te[i] == extant) {
                const lr_stream rng{(uint32_t)seed, (uint32_t)i};
                const double r = lr_pair(rng, (uint64_t)t, LR_P_SIM, 0).a;
                if (r < lt) births |= 1ull << k, n_birth += 1;
                else if (r < lt + mt) te[i] = (double)t, n_death += 1;
            }
        }
        // block-wide exclusive scan of n_birth, block totals of births and deaths
        int incl = n_birth;
        for (int d = 1; d < LR_WAVE; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        int deaths_w = n_death;
        for (int d = LR_WAVE / 2; d >= 1; d >>= 1) deaths_w += __shfl_xor(deaths_w, d);
        if (lane == LR_WAVE - 1) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wave_tot[w];
            total += wave_tot[w];
        }
        if (lane == 0 && deaths_w) atomicAdd((unsigned long long*)&counters[1], (unsigned long long)(long long)(-deaths_w));
        if (tid == 0) {
            long long base = 0;
            if (total) {
                base = (long long)atomicAdd((unsigned long long*)&counters[0], (unsigned long long)total);
                atomicAdd((unsigned long long*)&counters[1], (unsigned long long)total);
            }
            base_s = base;
        }
        __syncthreads();
        long long slot = base_s + before + (incl - n_birth);
        while (births) {
            births &= births - 1;
            if (slot < capacity) ts[slot] = (double)t, te[slot] = extant;
            else counters[2] = 1;
            ++slot;
        }
        __syncthreads();   // base_s / wave_tot are reused by the next round
    }
}

// after an overflow the allocation counter may exceed the capacity: clamp what the caller reads
__global__ void lr_sim_finish_kernel(long long capacity, long long* __restrict__ counters) {
    if (counters[0] > capacity) counters[0] = capacity;
}

extern "C" int lr_simulate_bd(const double* lam_steps, const double* mu_steps, int32_t n_steps, int32_t mode, double l0,
                              double m0, double K, double scale, int64_t n_start, int64_t capacity, uint64_t seed,
                              double* ts, double* te, int64_t* counters, int64_t* alive_trace, void* workspace,
                              int64_t workspace_bytes, void* stream_) {
    if (!ts || !te || !counters || !workspace) return LR_ERR_NULL;
    if (mode < 0 || mode > 2) return LR_ERR_MODEL;
    if (mode == 0 && (!lam_steps || !mu_steps)) return LR_ERR_NULL;
    if (n_steps < 1 || n_start < 1 || capacity < n_start) return LR_ERR_SIZE;
    if (mode != 0 && (!(K > 0.0) || !(scale > 0.0))) return LR_ERR_SIZE;
    if (workspace_bytes < (int64_t)sizeof(lr_sim_step)) return LR_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    lr_sim_step* step = (lr_sim_step*)workspace;
    hipLaunchKernelGGL(lr_sim_init_kernel, dim3((unsigned)((n_start + 255) / 256)), dim3(256), 0, stream, ts, te,
                       (long long)n_start, n_steps, (long long*)counters);
    long long blocks = (capacity + 255) / 256;
    if (blocks > 2048) blocks = 2048;     // 8 blocks per CU, 64 items per thread and round
    for (int t = 0; t < n_steps; ++t) {
        hipLaunchKernelGGL(lr_sim_prepare_kernel, dim3(1), dim3(1), 0, stream, lam_steps, mu_steps, t, mode, l0, m0, K,
                           scale, (long long)capacity, (long long*)counters, step, (long long*)alive_trace);
        hipLaunchKernelGGL(lr_sim_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ts, te, t, n_steps,
                           (long long)capacity, (unsigned long long)seed, step, (long long*)counters);
    }
    hipLaunchKernelGGL(lr_sim_finish_kernel, dim3(1), dim3(1), 0, stream, (long long)capacity, (long long*)counters);
    return (int)hipGetLastError();
}
