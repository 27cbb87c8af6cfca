// lr_pack.hip - the lineages as the persistent engines scan them.
//
// A lineage needs two table entries per chain pair: its birth bin's and its death bin's (lr_device.h: S and E tables).
// The engines keep the lineages ordered by (birth bin, te) - ChainEngine sorts them once - so consecutive lineages share
// their birth bin for long runs (100k lineages over 128 bins: ~780 per bin) and, inside a run, die in the same or a
// neighbouring bin.  The packing makes both explicit:
//
//   group = up to 14 CONSECUTIVE lineages of ONE birth bin in LR_SLOTS = 7 slots, 16 bytes:
//           a 16-bit header (birth index a << 4 | number of lineages), then seven 16-bit entry BYTE OFFSETS (index << 4) into the block's
//           pair table
//   slot  = ONE lineage (entry H + j: its death entry E[j]) or TWO consecutive lineages of the run with death entries
//           j and j + d, 0 <= d <= 3 at unit resolution (entry (2 + d) H + j: the pair-sum plane E[j] + E[j + d]) /
//           d = 0 on general times (entry 2 H + j: the doubled plane); unused slots point at E[0] = 0
//
// A lane of the scan loads one group (16 bytes, coalesced), gathers S[a] ONCE and one entry per slot: 8 gathers for up to
// 14 lineages instead of 28.  Every lineage still enters through its own (birth bin, death bin) - S[a] `count` times, its
// death entry alone or inside the pair sum of its slot - the sort order is what lets the gathers be shared; the pair-sum
// planes are derived from the chain's own E plane in LDS every time a table is built (lr_scan.h), so global memory and
// the launch-based engine know nothing of them.  The caller's order is kept (the likelihood is a sum; only its
// rounding depends on the order), which makes the packing four prefix scans:
//   run start of lineage i      = max-scan of (i if a_i != a_{i-1} else 0)
//   stretch start of lineage i  = max-scan of (0 if i can share a slot with i - 1 else i); greedy pairs from a stretch's
//                                 start: the lineages at even positions of their stretch are the slot HEADS
//   slot of a head inside its run = sum-scan of the head flags, minus its value at the run start
//   group of a head             = sum-scan of (head and slot % 7 == 0) - 1
// so the number of groups depends on the data: <= ceil(N / 7) + runs (all singles), ~ N / 14 + runs for sorted input
// (cfg4: 49,953 pairs and 94 singles), runs <= n_bins + 2.
// On general (non-integer) times the in-bin fractions travel beside the entries as 32-bit fixed point in LR_FRAC_ARRAYS
// = 3 more arrays of uint4: the slots' fe' = ceil te - te (a pair: the mean of its two, which the doubled slope plane
// turns back into their sum) and the SUM of the group's fs = ts - floor ts as one exact double.
//
// The scanner-wave shares of the persistent kernels (lr_p4_shares; equal by default) are a permutation of whole groups
// and are applied here; they depend on the group count, so they are fixed here too, after the scans.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "lr_engine.h"

static int lr_env_int_pack(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

__device__ __forceinline__ int lr_birth_index(double s, double t0, int n_bins) {
    return min(max(__double2int_rz(floor(s) - t0), -1), n_bins) + 1;
}
__device__ __forceinline__ int lr_death_index(double e, double t0, int n_bins) {
    return min(max(__double2int_rz(ceil(e) - t0), 0), n_bins + 1);
}

// run-start candidates: i where the birth bin changes (lineage 0 included), else 0
__global__ void lr_pack_runs_kernel(const double* __restrict__ ts, long long n, double t0, int n_bins,
                                    int* __restrict__ start_cand) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = lr_birth_index(ts[i], t0, n_bins);
    const int ap = i > 0 ? lr_birth_index(ts[i - 1], t0, n_bins) : -1;
    start_cand[i] = (a != ap) ? (int)i : 0;
}

// where group g of the plain order goes under the scanner-wave shares (see lr_persist4_kernel)
__device__ __forceinline__ long long lr_share_permute(long long g, int k_tot, const lr_p4_shares& sh) {
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
    return g;
}

// ---- pair slots (lr_scan.h) ------------------------------------------------------------------------------------------
// death entry of lineage i as the packed scan addresses it (model 3: extant lineages gather the extant block, by birth bin)
__device__ __forceinline__ int lr_pack_death_entry(const double* __restrict__ ts, const double* __restrict__ te, long long i,
                                                   double t0, int n_bins, int extant_block, double end_time, bool* extant) {
    const double e = te[i];
    *extant = extant_block && e >= end_time;
    return *extant ? n_bins + 2 + lr_birth_index(ts[i], t0, n_bins) : lr_death_index(e, t0, n_bins);
}
// lineage i can share a slot with lineage i - 1: same run, both with a death entry of the window, 0 <= d <= LR_PAIR_DMAX
__device__ __forceinline__ bool lr_pack_pairable(const double* __restrict__ ts, const double* __restrict__ te, long long i,
                                                 const int* __restrict__ run_start, double t0, int n_bins, int extant_block,
                                                 double end_time, int dmax) {
    if (i == 0 || run_start[i] == (int)i) return false;
    bool x0, x1;
    const int d0 = lr_pack_death_entry(ts, te, i - 1, t0, n_bins, extant_block, end_time, &x0);
    const int d1 = lr_pack_death_entry(ts, te, i, t0, n_bins, extant_block, end_time, &x1);
    return !x0 && !x1 && d1 >= d0 && d1 - d0 <= dmax;
}
// stretch-start candidates: a stretch = a maximal sequence of lineages each pairable with its predecessor
__global__ void lr_pack_stretch_kernel(const double* __restrict__ ts, const double* __restrict__ te, long long n, double t0,
                                       int n_bins, const int* __restrict__ run_start, int extant_block, double end_time,
                                       int dmax, int* __restrict__ cand) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cand[i] = lr_pack_pairable(ts, te, i, run_start, t0, n_bins, extant_block, end_time, dmax) ? 0 : (int)i;
}
// slot heads: the lineages at even positions of their stretch (greedy pairing from the stretch start)
__global__ void lr_pack_heads_kernel(const int* __restrict__ stretch_start, long long n, int* __restrict__ head) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (((int)i - stretch_start[i]) & 1) ? 0 : 1;
}
// new-group flags: heads whose slot number inside the run is a multiple of LR_SLOTS
__global__ void lr_pack_slot_flags_kernel(const int* __restrict__ run_start, const int* __restrict__ stretch_start,
                                          const int* __restrict__ head_incl, long long n, int* __restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = !(((int)i - stretch_start[i]) & 1);
    const int slot = head_incl[i] - head_incl[run_start[i]];          // the run start is a head: slots before i in the run
    flag[i] = (head && slot % LR_SLOTS == 0) ? 1 : 0;
}
__global__ void lr_pack_slots_kernel(const double* __restrict__ ts, const double* __restrict__ te, long long n, double t0,
                                     int n_bins, int H, const int* __restrict__ run_start, const int* __restrict__ stretch_start,
                                     const int* __restrict__ head_incl, const int* __restrict__ group_incl, int permute,
                                     int k_tot, long long share_base, lr_p4_shares sh, unsigned short* __restrict__ out, int extant_block,
                                     double end_time, unsigned int* __restrict__ frac, long long fstride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (((int)i - stretch_start[i]) & 1) return;                       // second lineage of a pair: written by its head
    const int rs = run_start[i];
    const int slot_in_run = head_incl[i] - head_incl[rs];
    const int slot = slot_in_run % LR_SLOTS;
    long long g = group_incl[i] - 1;
    // (the shares apply behind the helper waves' groups [0, share_base) of the four-chain kernel's helper form)
    if (permute && g >= share_base) g = share_base + lr_share_permute(g - share_base, k_tot, sh);
    unsigned short* grp = out + g * 8;
    bool x0;
    const int d0 = lr_pack_death_entry(ts, te, i, t0, n_bins, extant_block, end_time, &x0);
    const bool pair = i + 1 < n && stretch_start[i + 1] == stretch_start[i];
    int entry = H + d0;                                                // single: the E plane
    if (pair) {
        bool x1;
        const int d1 = lr_pack_death_entry(ts, te, i + 1, t0, n_bins, extant_block, end_time, &x1);
        entry = (2 + (d1 - d0)) * H + d0;                              // pair: plane 2 + d (general times: d = 0 only)
    }
    grp[1 + slot] = (unsigned short)(entry << 4);                     // the BYTE offset of the 16-byte entry (lr_word_off16)
    // general times: fs = ts - floor ts and fe' = ceil te - te (both in [0, 1)) as 32-bit fixed point, rounded to nearest
    auto fix = [](double f) { return fmin(rint(f * 4294967296.0), 4294967295.0); };
    if (frac) {
        // the slot's fe' (model 3: an extant lineage carries fs here, see lr_step.h); a pair: the mean of its two, which
        // the doubled slope of the E2 plane turns back into their sum
        const double s0 = ts[i], e0 = te[i];
        double fe = x0 ? fix(s0 - floor(s0)) : fix(ceil(e0) - e0);
        if (pair) {
            const double e1 = te[i + 1];
            // (half-way cases to even: always up would bias the sum by a quarter ulp per pair - 4e-10 per lineage at
            // exposure rates of 8, i.e. 4e-9 of a 1.3-million-lineage log-likelihood; unbiased, the error grows with the
            // square root of the number of pairs)
            fe = fmin(rint((fe + fix(ceil(e1) - e1)) * 0.5), 4294967295.0);
        }
        frac[((size_t)(slot >> 2) * fstride + g) * 4 + (slot & 3)] = (unsigned int)fe;
    }
    if (slot == 0) {
        // header: birth index and the number of lineages in the group's (up to LR_SLOTS) slots; unused slots -> E[0] = 0
        const int a = lr_birth_index(ts[i], t0, n_bins);
        int cnt = 0, slots = 0;
        long long k = i;
        while (slots < LR_SLOTS && k < n && run_start[k] == rs) {
            const bool pr = k + 1 < n && stretch_start[k + 1] == stretch_start[k];     // k is a head by construction
            cnt += pr ? 2 : 1, k += pr ? 2 : 1, ++slots;
        }
        grp[0] = (unsigned short)((a << 4) | cnt);                         // masked with 0xfff0: the byte offset of S[a]
        for (int q = slots; q < LR_SLOTS; ++q) grp[1 + q] = (unsigned short)(H << 4);
        if (frac) {
            // the sum of the group's fs (an exact integer below 2^36) as a double: array 2, (.x, .y)
            double sum = 0.0;
            for (long long m = i; m < k; ++m) {
                const double sm = ts[m];
                sum += fix(sm - floor(sm));
            }
            unsigned int* q = frac + ((size_t)2 * fstride + g) * 4;
            q[0] = (unsigned int)__double2loint(sum), q[1] = (unsigned int)__double2hiint(sum);
        }
    }
}

struct lr_max_op {
    __host__ __device__ int operator()(int a, int b) const { return a > b ? a : b; }
};

static size_t lr_scan_tmp_bytes(long long n) {
    size_t b1 = 0, b2 = 0;
    (void)rocprim::inclusive_scan(nullptr, b1, (const int*)nullptr, (int*)nullptr, (size_t)n, lr_max_op());
    (void)rocprim::inclusive_scan(nullptr, b2, (const int*)nullptr, (int*)nullptr, (size_t)n, rocprim::plus<int>());
    return (b1 > b2 ? b1 : b2) + 256;
}

long long lr_pack_tmp_bytes(long long n) {
    // four int32 arrays (run start, stretch start, slot heads, group number) + the scans' temporary storage + the group
    // count read by the host
    return (long long)(4 * lr_align_up64(n * 4, 256) + (long long)lr_scan_tmp_bytes(n) + 256);
}

// Shares of the scanner waves given the number of groups (moved here from lr_mcmc_create: the count depends on the data)
static void lr_set_shares(lr_engine* e) {
    // Tuned on cfg4 with in-kernel stamps until the waves of a phase finish together: per wave pair (2,3) (4,5) ...
    // (14,15) of the four-chain kernel the trips beyond / short of the equal share, per 14 trips.  The SIMD arbiter
    // serves its oldest wave first and SIMDs 0, 1 also host the stepper waves, hence the shape.  Kept as fractions of
    // the trip count for other inputs; short scans (bound by the chain step) stay equal.
    static const char* env = getenv("LR_P4_SHARES");       // "d2,d4,d6,d8,d10,d12,d14" for 14 trips
    static const int env2 = lr_env_int_pack("LR_P2_SHARE", 0);   // two-chain kernel (no gain measured: off)
    for (int j = 0; j < 16; ++j) e->p4.delta[j] = 0;
    e->p4.help_trips = 0;
    e->p4_help = lr_p4_help_choice(e);
    e->p4_spec = lr_p4_spec_choice(e);
    if (e->lay.persistent == 2 && e->p4_help) {
        // a helper wave is idle until its stepper's hand-over arrives (~1.2 us into a phase, a scan trip takes ~0.3 us):
        // it scores the first groups meanwhile (the 128 helper lanes stride over [0, 128 trips), the scanners over the rest).
        // With t trips per scanner lane behind that share (groups = 128 h + 768 t) the two sides end together at about
        // h = t - 3.5; more than five never paid - the helpers scan with plain loads, one group in flight per lane
        // (scratch/exp_help_trips.py: 60k lineages 2, cfg4 5, 300k and 1M 5).  LR_P4_HELP_TRIPS overrides it.
        static const int env_t = lr_env_int_pack("LR_P4_HELP_TRIPS", -1);
        const double t = ((double)e->n8 + 512.0) / 896.0;
        int h = (int)lrint(t - 3.5);
        h = h < 0 ? 0 : (h > 5 ? 5 : h);
        e->p4.help_trips = env_t >= 0 ? env_t : h;
        if (e->p4.help_trips < 0) e->p4.help_trips = 0;
        if ((long long)e->p4.help_trips * 128 > e->n8) e->p4.help_trips = 0;
    }
    if (e->lay.persistent == 0 || e->lay.persistent == 3) {
        e->p4.n_slots = 8;       // speculative kernel / the launch-based engine's packed scan: plain layout, the groups in order
    } else if (e->lay.persistent == 2) {
        e->p4.n_slots = e->p4_help ? 12 : 14;      // (with helper waves: twelve scanners, equal shares)
        // (with 14-lineage groups a scan is ~8 trips of cfg4 and equal shares measure as fast as any: the default is
        // equal; the knob stays for experiments)
        int base[7] = {0, 0, 0, 0, 0, 0, 0};
        if (env && e->p4.n_slots == 14) sscanf(env, "%d,%d,%d,%d,%d,%d,%d", &base[0], &base[1], &base[2], &base[3], &base[4], &base[5], &base[6]);
        // helper form: "d4,d6,d8,d10,d12,d14" per wave pair (4,5) ... (14,15), trips per 9 trips of a scanner lane
        // (SIMDs 0, 1 - pairs (4,5), (8,9), (12,13) - also carry the steppers, SIMDs 2, 3 the helpers)
        static const char* env12 = getenv("LR_P4_SHARES12");
        if (env12 && e->p4.n_slots == 12) sscanf(env12, "%d,%d,%d,%d,%d,%d", &base[0], &base[1], &base[2], &base[3], &base[4], &base[5]);
        const long long behind = e->p4_help ? (long long)e->p4.help_trips * 128 : 0;
        const int per = e->p4.n_slots * 64;
        const int k_tot = (int)((e->n8 - behind + per - 1) / per);
        int sum = 0;
        for (int j = 0; j < 7; ++j) {
            int d = (k_tot >= 6) ? (int)lrint((double)base[j] * k_tot / (e->p4.n_slots == 12 ? 9.0 : 14.0)) : 0;
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
        // two-chain kernel: 8 waves, two per SIMD (16 in its wide form, four per SIMD: the oldest-first pattern of the
        // four-chain kernel - waves 0..3 / 4..7 take trips from 12..15 / 8..11, per 12 trips)
        const bool wide2 = e->lay.reserved1 == 1024;
        e->p4.n_slots = wide2 ? 16 : 8;
        const int k_tot = (int)((e->n8 + e->p4.n_slots * 64 - 1) / (e->p4.n_slots * 64));
        int d = (k_tot >= 12) ? (int)lrint((double)env2 * k_tot / 24.0) : 0;
        if (d > LR_P4_MAX_GIVE) d = LR_P4_MAX_GIVE;
        if (e->p4.n_slots == 8)
            for (int j = 0; j < 4; ++j) e->p4.delta[j] = d, e->p4.delta[4 + j] = -d;
        if (e->p4.n_slots == 16 && k_tot >= 6) {
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
    // the takers' extra trips must stay inside the zero-filled spare behind the packed groups
    const long long stride = (long long)e->p4.n_slots * 64;
    const long long k_tot = (e->n8 + stride - 1) / stride;        // (an upper bound in the helper form, whose region is shorter)
    int dmax = 0;
    for (int j = 0; j < 16; ++j) dmax = e->p4.delta[j] > dmax ? e->p4.delta[j] : dmax;
    if ((k_tot + dmax + 1) * stride + (e->p4_help ? (long long)e->p4.help_trips * 128 : 0) > e->n8_alloc)
        for (int j = 0; j < 16; ++j) e->p4.delta[j] = 0;
}

int lr_pack_lineages(lr_engine* e, hipStream_t stream) {
    const long long n = e->cfg.n_lineages;
    if (n >= (1ll << 31)) return LR_ERR_SIZE;                         // lineage numbers are scanned as int32
    const bool general = e->plan.unit == LR_TAB_PAIRGEN;
    const int extant_block = e->cfg.model == LR_MODEL_KEIDING_DEAD ? 1 : 0;
    char* tmp = e->ws + e->lay.pack_tmp;
    const long long arr = lr_align_up64(n * 4, 256);
    int* run_start = (int*)tmp;
    int* group_incl = (int*)(tmp + arr);
    int* stretch_start = (int*)(tmp + 2 * arr);
    int* head_incl = (int*)(tmp + 3 * arr);
    char* scan_tmp = tmp + 4 * arr;
    size_t scan_bytes = lr_scan_tmp_bytes(n);
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    // group_incl doubles as the scans' input buffer: candidates -> run starts, flags -> group numbers
    hipLaunchKernelGGL(lr_pack_runs_kernel, grid, blk, 0, stream, e->ts, n, e->cfg.t0, e->cfg.n_bins, group_incl);
    hipError_t he = rocprim::inclusive_scan(scan_tmp, scan_bytes, (const int*)group_incl, run_start, (size_t)n, lr_max_op(), stream);
    if (he != hipSuccess) return (int)he;
    {
        // Pair slots.  Stretches of pairable neighbours (max-scan), greedy pairs from each stretch start (heads = even
        // positions, sum-scan = slot numbers), a new group every LR_SLOTS slots of a run.  Unit resolution pairs death
        // entries up to LR_PAIR_DMAX apart, general times equal ones only.
        hipLaunchKernelGGL(lr_pack_stretch_kernel, grid, blk, 0, stream, e->ts, e->te, n, e->cfg.t0, e->cfg.n_bins,
                           (const int*)run_start, extant_block, e->cfg.end_time, general ? 0 : LR_PAIR_DMAX, group_incl);
        scan_bytes = lr_scan_tmp_bytes(n);
        he = rocprim::inclusive_scan(scan_tmp, scan_bytes, (const int*)group_incl, stretch_start, (size_t)n, lr_max_op(), stream);
        if (he != hipSuccess) return (int)he;
        hipLaunchKernelGGL(lr_pack_heads_kernel, grid, blk, 0, stream, (const int*)stretch_start, n, group_incl);
        scan_bytes = lr_scan_tmp_bytes(n);
        he = rocprim::inclusive_scan(scan_tmp, scan_bytes, (const int*)group_incl, head_incl, (size_t)n, rocprim::plus<int>(), stream);
        if (he != hipSuccess) return (int)he;
        hipLaunchKernelGGL(lr_pack_slot_flags_kernel, grid, blk, 0, stream, (const int*)run_start, (const int*)stretch_start,
                           (const int*)head_incl, n, group_incl);
    }
    scan_bytes = lr_scan_tmp_bytes(n);
    he = rocprim::inclusive_scan(scan_tmp, scan_bytes, (const int*)group_incl, group_incl, (size_t)n, rocprim::plus<int>(), stream);
    if (he != hipSuccess) return (int)he;
    int n_groups = 0;
    he = hipMemcpyAsync(&n_groups, group_incl + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (he != hipSuccess) return (int)he;
    if (n_groups < 1 || (long long)n_groups + LR_IDX_SPARE > e->n8_alloc) return LR_ERR_ORDER;   // too many runs: unsorted input
    e->n8 = n_groups;
    lr_set_shares(e);
    // zero fill (general: a zero group = sentinel entries on both sides, contribution 0), then the groups
    he = hipMemsetAsync(e->ws + e->lay.lineage_idx, 0, (size_t)e->n8_alloc * 16, stream);
    if (he == hipSuccess && general) he = hipMemsetAsync(e->ws + e->lay.lineage_frac, 0, (size_t)e->n8_alloc * 16 * LR_FRAC_ARRAYS, stream);
    if (he != hipSuccess) return (int)he;
    bool any = false;
    for (int j = 0; j < 16; ++j) any |= e->p4.delta[j] != 0;
    const long long share_base = (e->lay.persistent == 2 && e->p4_help) ? (long long)e->p4.help_trips * 128 : 0;
    const int k_tot = (int)((e->n8 - share_base + e->p4.n_slots * 64 - 1) / (e->p4.n_slots * 64));
    // (an all-zero group of the spare: birth entry 0 x count 0 and seven gathers of S[0] = 0 - contribution 0)
    hipLaunchKernelGGL(lr_pack_slots_kernel, grid, blk, 0, stream, e->ts, e->te, n, e->cfg.t0, e->cfg.n_bins, e->plan.H,
                       (const int*)run_start, (const int*)stretch_start, (const int*)head_incl, (const int*)group_incl,
                       any ? 1 : 0, k_tot, share_base, e->p4, (unsigned short*)(e->ws + e->lay.lineage_idx), extant_block, e->cfg.end_time,
                       general ? (unsigned int*)(e->ws + e->lay.lineage_frac) : nullptr, (long long)e->n8_alloc);
    return (int)hipGetLastError();
}
