// lr_stream.hip - the RJMCMC loop where HBM bounds it (few chains x very many lineages), resident on the device:
// ONE launch runs n iterations of   scan all lineages for all chains -> Metropolis-Hastings step of every chain
// (runMCMC, /root/reference/LiteRateForward.py:233-319).
//
// The launch-based engine pays a kernel boundary on either side of its chain-step kernel and runs the step (6 us) serially
// behind every scan: 8.5 of 39.5 us per iteration on 16 chains x 1e7 lineages.  Here the grid is sized to be resident all
// at once and stays for the whole call:
//   * SCANNER blocks (tiles x chain groups of them) own a fixed tile of the lineages.  Per iteration: wait until the tables
//     of the iteration are published, stage them in LDS, stream the tile (the launch-based scan's own body: same sums in
//     the same order), store the tile's partial sums, count in.
//   * STEPPER waves (one per chain, four per block, the first blocks of the grid) keep their chain's state in registers.
//     While the scan of iteration k runs, a stepper already takes the step that FOLLOWS IF THE PENDING PROPOSAL IS REJECTED
//     (the usual case: 87-99 % of the iterations) - trace row, proposal k + 1, its lookup table in the other table buffer.
//     A Gibbs step (always accepted) or an invalid proposal (always rejected) is decided before its scan ends, so for those
//     the early step is the real one.  When the last scanner block has counted in, the stepper adds its chain's row of tile
//     partials, decides, and either adopts the early step or - accepted proposal - takes the step again from the new state.
//     Then it publishes the table.  lr_propose_rj is a pure function of (state, iteration): the trajectories are those of
//     the launch-based engine bit for bit.
// A scanner block that is done with its tile stages the NEXT iteration's tables as they stand (the early ones) right away and
// stages them again only if a table was written again since (a counter the steppers move around every second write).
// Critical path of an iteration: scan -> counter -> partial sums of one row -> decision -> counter, the scanners' tables in
// LDS and their first loads of ts / te in flight while they wait.
//
// MEASURED (round 5, 16 chains, unit resolution, us per iteration; profiles/r05_stream_stamps.txt holds the in-kernel stamps):
//                              1e7 lineages   3e7     1e8
//   launches (scan + step)        38.1        96      284      <- what the planner keeps
//   this kernel, fences           100.9       170     359      (release / acquire fences in every scanner block)
//   ... atomics instead            55.3       115     305
//   ... + early staging            49.9       116     296
//   ... + a line of flags per 16 blocks, fence-free steppers, one polling lane:  51.5 - 54.5, 108 - 111, 300 - 302
// It is NOT faster, and the stamps say why.  (1) Tables another XCD's wave has written can only be read past the L2
// (agent-scope loads; the alternative, an invalidate per block, wipes the L2 for everybody): 977 blocks x 35 KB = 35 MB per
// iteration from the Infinity Cache beside the 160 MB of ts / te - the slowest block ends its tile after 36 us where the
// launched scan, whose blocks share the tables through their XCD's L2, takes 30.5 in all.  (2) A flag takes ~1 us from a
// stepper to a scanner and back, about what the two kernel boundaries it replaces cost inside a captured graph (7.6 us
// for boundaries + step kernel).  (3) A chain's first iterations accept 11-13 % of the proposals: with 16 chains 89 % of
// the iterations have at least one step to take again (5-8 us) behind the scan, as the launches' step kernel does always.
// The kernel stays as an opt-in engine (engine_mode 6 / LR_STREAM=1), bit-identical to the launches, with its tests.
//
// Publication: relaxed agent-scope counters / words; what is shared (tables, partial sums) is written and read with agent-scope
// atomic accesses, drained (s_waitcnt) before the count - NO cache-wide fence anywhere (lr_stream_io, lr_stream_write_through).
// Every wait is bounded (two seconds of the wall clock, or the engine's status word raised by someone else) and raises
// the status word.
#include <cstdlib>

#include "lr_engine.h"

// (typed for the atomics' address space)
typedef __attribute__((address_space(1))) unsigned long long lr_sgu64;
typedef __attribute__((address_space(1))) unsigned int lr_sgu32;

#ifdef LR_STREAM_STAMPS
// diagnostic builds: wall-clock stamps (10 ns) of the LAST iteration of a launch - [0..15] stepper of chain 0, [16..31]
// scanner block 0, [32..47] the last scanner block; read by lr_stream_dump_stamps (scratch/diag_stream.py)
static __device__ unsigned long long lr_stream_stamps[48 + 4 * 1024];   // ... then per scanner block: top, ready seen, scan start (tables staged), scan + reduce done
#define LR_TSTAMP(on, base, j) if ((on)) lr_stream_stamps[(base) + (j)] = wall_clock64()
extern "C" int lr_stream_dump_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_stream_stamps), sizeof(unsigned long long) * (48 + 4 * 1024));
}
#else
#define LR_TSTAMP(on, base, j)
#endif

struct lr_stream_args {
    const double* ts;
    const double* te;
    long long n, chunk;
    double t0;
    int n_bins, tiles, groups, step_blocks;
    long long n_iters;
    double2* tables2;            // the second table buffer: iteration k of a launch reads buffer k & 1 (0 = lr_step_args.tables)
    lr_stream_sync* sync;
    unsigned int* status;
};

// A stepper wave waits until *counter >= want (false = gave up, status word raised).  ONE lane polls (sixty-four lanes'
// atomic loads of one address are sixty-four requests: four polling waves kept their XCD's memory pipeline so busy that its
// scanner blocks finished 8 us behind the other XCDs'), at long intervals while many blocks are still out.
__device__ __forceinline__ bool lr_stream_wait(const unsigned long long* counter, unsigned long long want, unsigned int* status, int lane) {
    const unsigned long long t_start = wall_clock64();
    for (unsigned int spins = 0;; ++spins) {
        unsigned long long v = 0;
        if (lane == 0) v = __hip_atomic_load((lr_sgu64*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned int)v);
        if (v >= want) return true;
        if ((spins & 63u) == 63u) {
            unsigned int st = 0;
            if (lane == 0) st = __hip_atomic_load((lr_sgu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st = __builtin_amdgcn_readfirstlane(st);
            if (st != 0u || wall_clock64() - t_start > LR_SPEC_TIMEOUT_TICKS) {
                if (lane == 0) __hip_atomic_store((lr_sgu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        if (want - v > 48) __builtin_amdgcn_s_sleep(48);      // (~1.3 us)
        else __builtin_amdgcn_s_sleep(4);
    }
}

// the stepper wave of chain c writes its word into every copy (lane j: copy j)
__device__ __forceinline__ void lr_stream_publish(lr_stream_sync* sync, int c, int lane, unsigned long long word) {
    static_assert(LR_STREAM_COPIES == LR_WAVE, "a lane per copy");
    __hip_atomic_store((lr_sgu64*)&sync->word[lane][c], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A stepper wave makes the table it has just written visible to the whole device WITHOUT a fence (a release fence writes
// the XCD's whole L2 back: in-kernel stamps showed the scanner blocks of the four XCDs that host stepper blocks finishing
// 10 us behind the others): it reads its entries back (its own XCD's L2) and stores them again write-through, then drains.
__device__ __forceinline__ void lr_stream_write_through(const lr_step_args& a, double2* table, int lane) {
    double* t = reinterpret_cast<double*>(table);
    if (a.unit) {
        // a pair table: 2 H 16-byte entries, of which this chain owns one double each
        for (int j = lane; j < 2 * a.H; j += LR_WAVE)
            __hip_atomic_store((lr_sgu64*)(t + 2 * j), (unsigned long long)__double_as_longlong(t[2 * j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        for (int j = lane; j < 2 * a.tab_stride; j += LR_WAVE)
            __hip_atomic_store((lr_sgu64*)(t + j), (unsigned long long)__double_as_longlong(t[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS of a stepper block (in the dynamic LDS its scanner twins stage their tables in)
struct lr_stream_stepper_lds {
    lr_seg_scratch scratch[LR_STEP_WAVES_PER_BLOCK];
    double saved_f64[LR_STEP_WAVES_PER_BLOCK][LR_STATE_ROWS * LR_ROW];
    int saved_i32[LR_STEP_WAVES_PER_BLOCK][LR_ISTATE_ROWS * LR_ROW];
};

#ifndef LR_STREAM_Q
#define LR_STREAM_Q 16   /* loads of tile partials a stepper lane keeps in flight */
#endif

// What a scanner block shares with the stepper waves, and how it touches it.  A thousand scanner blocks using agent-scope
// fences would write back / invalidate their XCD's whole L2 a thousand times per iteration (measured: 100 us per iteration
// instead of 38): they read
// the tables and store their partial sums with agent-scope ATOMIC accesses instead (performed at the point of coherence,
// whatever the caches hold), the stores drained (s_waitcnt) before the block counts in; the steppers read the partial sums
// the same way.
struct lr_stream_io {
    static __device__ __forceinline__ double2 load16(const double2* p) {
        const lr_sgu64* q = (const lr_sgu64*)p;
        double2 v;
        v.x = __longlong_as_double((long long)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        v.y = __longlong_as_double((long long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        return v;
    }
    static __device__ __forceinline__ void store_partial(double* p, double v) {
        __hip_atomic_store((lr_sgu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// block-shared words of a scanner block
struct lr_stream_block {
    int gave_up;                   // a wait ran out
    int staged;                    // the tables of the coming iteration stand in LDS (staged early, and not written again since)
};

__device__ __forceinline__ unsigned int lr_word_early(unsigned long long w) { return (unsigned int)(w >> 40); }
__device__ __forceinline__ unsigned int lr_word_ready(unsigned long long w) { return (unsigned int)(w >> 16) & 0xffffffu; }
__device__ __forceinline__ unsigned int lr_word_changed(unsigned long long w) { return (unsigned int)w & 0xffffu; }

// Wave 0 of a scanner block, lane c < n_chains on chain c's word of the block's copy: waits until every chain's `early`
// (EARLY) or `ready` field has reached `want`; -> the lane's word as last read (lanes without a chain: 0), or gave_up set.
template <bool EARLY>
__device__ __forceinline__ unsigned long long lr_stream_wait_words(const unsigned long long* line, int n_chains, unsigned int want, unsigned int* status,
                                                                   int* gave_up) {
    const int lane = threadIdx.x;
    const bool mine = lane < n_chains;
    unsigned long long w = 0;
    const unsigned long long t_start = wall_clock64();
    for (unsigned int spins = 0;; ++spins) {
        if (mine) w = __hip_atomic_load((lr_sgu64*)(line + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(!mine || (EARLY ? lr_word_early(w) : lr_word_ready(w)) >= want)) return w;
        if ((spins & 63u) == 63u) {
            const unsigned int st = __hip_atomic_load((lr_sgu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != 0u || wall_clock64() - t_start > LR_SPEC_TIMEOUT_TICKS) {
                __hip_atomic_store((lr_sgu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) *gave_up = 1;
                return w;
            }
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

// what a scanner block does between its first loads of ts / te and the staging of the tables: waits until the steps of the
// previous iteration are decided; -> true if the tables it staged early are the final ones
struct lr_stream_table_wait : lr_stream_io {
    const unsigned long long* line;   // the block's copy of the chains' words
    int n_chains;
    unsigned int want;                // value of `ready` to wait for (0: first iteration of a launch, nothing to wait for)
    unsigned int changed_seen;        // (wave 0, lane c) chain c's `changed` as read before the early staging
    unsigned int* status;
    lr_stream_block* blk;             // in LDS
    int stamp_at, stamp_blk;          // (diagnostic builds: where to stamp, -1 = nowhere)
    __device__ __forceinline__ bool operator()() const {
        if (threadIdx.x < LR_WAVE && want != 0) {
            const unsigned long long w = lr_stream_wait_words<false>(line, n_chains, want, status, &blk->gave_up);
            LR_TSTAMP(stamp_at >= 0 && threadIdx.x == 0, stamp_at, 1);
            LR_TSTAMP(stamp_blk >= 0 && threadIdx.x == 0, stamp_blk, 1);
            const bool moved = __any(threadIdx.x < n_chains && lr_word_changed(w) != changed_seen);
            if (threadIdx.x == 0 && moved) blk->staged = 0;
        }
        __syncthreads();
        return blk->staged != 0;
    }
};

// (four blocks per CU must fit - the grid is planned on that: four waves per SIMD, 128 VGPRs)
template <int CB, int H, bool UNIT>
__global__ __launch_bounds__(LR_SCAN_THREADS, 4) void lr_stream_kernel(lr_step_args a, lr_stream_args f) {
    extern __shared__ double2 lds[];
    __shared__ lr_stream_block blk;
    const int bid = blockIdx.x, tid = threadIdx.x;
    const unsigned long long scan_blocks = (unsigned long long)f.tiles * f.groups;
    if (tid == 0) blk.gave_up = 0, blk.staged = 0;
    __syncthreads();

    if (bid >= f.step_blocks) {
        // ---- a scanner block ----
        const int sb = bid - f.step_blocks;
        const int tile = sb % f.tiles, group = sb / f.tiles;
        const int tile_stride = lr_tile_stride(a.tiles);
        double* partials = const_cast<double*>(a.partials);
        const unsigned long long* line = f.sync->word[sb % LR_STREAM_COPIES];
        unsigned int changed_seen = 0;
        for (long long k = 0; k < f.n_iters; ++k) {
            const double2* tables = (k & 1) ? f.tables2 : a.tables;
            const bool stamp = tid == 0 && k + 2 == f.n_iters && (sb == 0 || sb + 1 == (int)scan_blocks);   // (the last but one: it stages early)
            const int sbase = sb == 0 ? 16 : 32;
            (void)stamp, (void)sbase;
            LR_TSTAMP(stamp, sbase, 0);
            const bool stamp_all = tid == 0 && k + 2 == f.n_iters && sb < 1024;
            (void)stamp_all;
            LR_TSTAMP(stamp_all, 48 + 4 * sb, 0);
            lr_stream_table_wait w;
            w.line = line, w.n_chains = a.cfg.n_chains, w.want = (unsigned int)k, w.changed_seen = changed_seen, w.status = f.status, w.blk = &blk;
            w.stamp_at = stamp ? sbase : -1, w.stamp_blk = stamp_all ? 48 + 4 * sb : -1;
            if (UNIT)
                lr_scan_unit_body<CB, H, LR_UNIT_DEPTH, lr_stream_table_wait>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables, a.cfg.n_chains,
                                                                               f.chunk, partials, tile_stride, w);
            else
                lr_scan_fast_body<CB, H, LR_SCAN_THREADS, 1, 0, lr_stream_table_wait>(lds, tile, group * CB, f.ts, f.te, f.n, f.t0, f.n_bins, tables,
                                                                                      a.cfg.n_chains, f.chunk, partials, tile_stride, w);
            if (blk.gave_up) return;   // (written before the barrier inside the wait: uniform over the block)
            LR_TSTAMP(stamp, sbase, 2);
            LR_TSTAMP(stamp_all, 48 + 4 * sb, 3);
            // the tile's partial sums are wave 0's stores: drained, then counted in
            if (tid < LR_WAVE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LR_TSTAMP(stamp, sbase, 3);
                if (tid == 0) __hip_atomic_fetch_add((lr_sgu64*)&f.sync->arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (k + 1 == f.n_iters) break;
            // While the stepper waves decide: the tables of the next iteration as they stand - in all likelihood (the pending
            // proposals rejected) the final ones - staged into the LDS this iteration is done with.  `changed` is read first:
            // a table written again during or after this staging moves it, and the block stages again behind the decision.
            __syncthreads();           // (the reduction's scratch is the tables' LDS)
            if (tid < LR_WAVE) {
                changed_seen = lr_word_changed(lr_stream_wait_words<true>(line, a.cfg.n_chains, (unsigned int)(k + 1), f.status, &blk.gave_up));
                if (tid == 0) blk.staged = 1;
            }
            __syncthreads();
            if (blk.gave_up) return;
            LR_TSTAMP(stamp, sbase, 4);
            const double2* next = ((k + 1) & 1) ? f.tables2 : a.tables;
            if (UNIT) lr_stage_unit_tables<CB, H, lr_stream_io>(lds, next, group * CB, tid);
            else lr_stage_fast_tables<CB, H, LR_SCAN_THREADS, lr_stream_io>(lds, next, group * CB, min(CB, a.cfg.n_chains - group * CB), tid);
            LR_TSTAMP(stamp, sbase, 5);
        }
        return;
    }

    // ---- a stepper wave: chain c ----
    lr_stream_stepper_lds* sl = reinterpret_cast<lr_stream_stepper_lds*>(lds);
    const int wave = tid / LR_WAVE, lane = tid & (LR_WAVE - 1);
    const int c = bid * LR_STEP_WAVES_PER_BLOCK + wave;
    if (c >= a.cfg.n_chains) return;
    lr_chain_regs st;
    double* S = a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW;
    int* I = a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW;
    lr_chain_load(st, S, I, lane);
    // the state as it was before the early step, in LDS (in registers beside the step's own the wave would need ~150)
    double* S0 = sl->saved_f64[wave];
    int* I0 = sl->saved_i32[wave];
    lr_step_args a2 = a;
    a2.tables = f.tables2;
    const int es = lr_tab_es(a.unit, a.H);
    const double* row = a.partials + (size_t)c * lr_tile_stride(a.tiles);
    bool alive = true;
    unsigned long long word = 0;      // what this chain has published (lr_stream_sync.word)
    for (long long k = 0; k < f.n_iters; ++k) {
        double2* table_next = ((k + 1) & 1) ? lr_chain_table(a2, c) : lr_chain_table(a, c);
        const int gibbs = lr_bcast_i(st.isc, LR_I_GIBBS), invalid = lr_bcast_i(st.isc, LR_I_INVALID);
        const bool decided = gibbs != 0 || invalid != 0;   // the Metropolis-Hastings rule does not look at the likelihood
        // the early step, while the scan runs: the real one if the pending proposal is decided already, else on "rejected"
        const bool stamp = lane == 0 && c == 0 && k + 2 == f.n_iters;
        (void)stamp;
        LR_TSTAMP(stamp, 0, 0);
        lr_chain_store(st, S0, I0, lane);
        lr_chain_step_core(st, a, decided ? 0 : 2, c, lane, &sl->scratch[wave], 0.0, table_next, es);
        LR_TSTAMP(stamp, 0, 1);
        lr_stream_write_through(a, table_next, lane);
        LR_TSTAMP(stamp, 0, 2);
        word += 1ull << 40;                       // early
        lr_stream_publish(f.sync, c, lane, word);
        if (!lr_stream_wait(&f.sync->arrived, scan_blocks * (unsigned long long)(k + 1), f.status, lane)) {
            alive = false;
            break;
        }
        LR_TSTAMP(stamp, 0, 3);
        const double lik_sum = lr_wave_sum(lr_sum_tile_partials<LR_STREAM_Q, true>(row, a.tiles, lane));
        LR_TSTAMP(stamp, 0, 4);
        if (!decided) {
            double lik;
            const double* sc0 = S0 + LR_ROW_SCALARS * LR_ROW;
            const bool ok = lr_mh_accept(0, 0, lik_sum, sc0[LR_S_CONST_P], sc0[LR_S_LIKA], sc0[LR_S_PRIOR_P], sc0[LR_S_PRIORA], sc0[LR_S_HASTING],
                                         sc0[LR_S_LOG_U], &lik);
            if (!ok) {
                st.sc = (lane == LR_S_LIK_P) ? lik : st.sc;     // (what lr_chain_step_core records of a rejected proposal)
            } else {
                // accepted: the step again, from the state as it was; the table is written again between two counts of
                // `changed` (the first one visible before the first store of the table: the wave waits for it)
                word = (word & ~0xffffull) | ((word + 1) & 0xffffull);      // changed
                lr_stream_publish(f.sync, c, lane, word);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lr_chain_load(st, S0, I0, lane);
                lr_chain_step_core(st, a, 0, c, lane, &sl->scratch[wave], lik_sum, table_next, es);
                lr_stream_write_through(a, table_next, lane);
                word = (word & ~0xffffull) | ((word + 1) & 0xffffull);      // changed, again: published with `ready` below
            }
        }
        // decided: the tables of iteration k + 1 are final
        LR_TSTAMP(stamp, 0, 5);
        word += 1ull << 16;                       // ready
        lr_stream_publish(f.sync, c, lane, word);
    }
    if (!alive) return;    // (the run is void: the status word says so)
    lr_chain_store(st, S, I, lane);
    if (f.n_iters & 1) {
        // the pending proposal's table stands in the second buffer: everything outside this kernel reads the first
        const double* src = reinterpret_cast<const double*>(lr_chain_table(a2, c));
        double* dst = reinterpret_cast<double*>(lr_chain_table(a, c));
        if (a.unit) {
            // a pair table: 2 H 16-byte entries, of which this chain owns one double each
            for (int j = lane; j < 2 * a.H; j += LR_WAVE) dst[2 * j] = src[2 * j];
        } else {
            for (int j = lane; j < 2 * a.tab_stride; j += LR_WAVE) dst[j] = src[j];
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------

static int lr_stream_step_blocks(int n_chains) { return (n_chains + LR_STEP_WAVES_PER_BLOCK - 1) / LR_STEP_WAVES_PER_BLOCK; }

static size_t lr_stream_lds_bytes(const lr_scan_plan& p) {
    const size_t scratch = sizeof(lr_stream_stepper_lds);
    return (size_t)p.lds_bytes > scratch ? (size_t)p.lds_bytes : scratch;
}

bool lr_stream_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    // OPT-IN (engine_mode 6, or LR_STREAM=1 for engine_mode 0): measured, it does not beat the launches it replaces - see the
    // head of this file
    static const int env = getenv("LR_STREAM") ? atoi(getenv("LR_STREAM")) : 0;
    if (cfg->engine_mode != 6 && !(env && cfg->engine_mode == 0)) return false;
    if (cfg->sampler != 0 || cfg->n_chains > LR_STREAM_MAX_CHAINS) return false;
    if (!lr_fused_supported(p) || p.unit == LR_TAB_PAIRGEN || p.groups != 1) return false;   // (one pass of the lineages per iteration)
    if (p.H != 40 && p.H != 72 && p.H != 136 && p.H != 264) return false;
    return lr_stream_lds_bytes(p) <= 160 * 1024;
}

// resident block slots the grid may use: four 256-thread blocks per CU at most (one round of the launch-based scan's tiles),
// fewer when the tables take more than a quarter of the LDS
int lr_stream_slots(const lr_scan_plan& p, int cus) {
    int per_cu = (int)((size_t)(160 * 1024) / lr_stream_lds_bytes(p));
    if (per_cu > 4) per_cu = 4;
    return per_cu * cus;
}

// tiles of the resident grid: every slot but the steppers', shared among the chain groups
void lr_stream_plan(const lr_mcmc_config* cfg, lr_scan_plan* p, int cus) {
    const int slots = lr_stream_slots(*p, cus) - lr_stream_step_blocks(cfg->n_chains);
    long long tiles = slots / p->groups;
    const long long unit = 2 * LR_SCAN_THREADS;
    const long long max_tiles = (cfg->n_lineages + 4 * unit - 1) / (4 * unit);
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    // (an odd number of 4 KB steps from tile to tile - the blocks start every iteration at the same instant - measured: no
    // difference)
    const long long chunk = lr_align_up64((cfg->n_lineages + tiles - 1) / tiles, unit);
    p->tiles = (int)((cfg->n_lineages + chunk - 1) / chunk);
    p->chunk = chunk;
}

template <int CB, int H, bool UNIT>
static int lr_stream_launch_one(const lr_engine* e, const lr_step_args& a, const lr_stream_args& f, int grid, size_t lds_bytes, bool query,
                                hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&lr_stream_kernel<CB, H, UNIT>);
    if (lds_bytes > 64 * 1024) {
        // (per call: the attribute belongs to the function on the CURRENT device)
        const hipError_t he = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (he != hipSuccess) return (int)he;
    }
    if (query) {
        // can the whole grid be resident at once?  (it must: its blocks wait for each other)
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, LR_SCAN_THREADS, lds_bytes);
        if (he == hipSuccess) he = hipGetDevice(&dev);
        if (he == hipSuccess) he = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (he != hipSuccess) return (int)he;
        return (long long)per_cu * cus >= grid ? LR_OK : LR_ERR_STATE;
    }
    hipLaunchKernelGGL((lr_stream_kernel<CB, H, UNIT>), dim3(grid), dim3(LR_SCAN_THREADS), lds_bytes, stream, a, f);
    return (int)hipGetLastError();
}

template <int H>
static int lr_stream_launch_h(const lr_engine* e, const lr_step_args& a, const lr_stream_args& f, int grid, size_t lds_bytes, bool query,
                              hipStream_t stream) {
    if (e->plan.unit) {
        switch (e->plan.cb) {
            case 16: return lr_stream_launch_one<16, H, true>(e, a, f, grid, lds_bytes, query, stream);
            case 8: return lr_stream_launch_one<8, H, true>(e, a, f, grid, lds_bytes, query, stream);
            default: return LR_ERR_SIZE;
        }
    }
    switch (e->plan.cb) {
        case 8: return lr_stream_launch_one<8, H, false>(e, a, f, grid, lds_bytes, query, stream);
        case 4: return lr_stream_launch_one<4, H, false>(e, a, f, grid, lds_bytes, query, stream);
        default: return LR_ERR_SIZE;
    }
}

// query = true: only asks whether the grid fits the device in use (LR_OK / LR_ERR_STATE)
int lr_launch_stream(lr_engine* e, const lr_step_args& a, int64_t n_iters, bool query, hipStream_t stream) {
    lr_stream_args f;
    f.ts = e->ts, f.te = e->te, f.n = e->cfg.n_lineages, f.chunk = e->plan.chunk, f.t0 = e->cfg.t0;
    f.n_bins = e->cfg.n_bins, f.tiles = e->plan.tiles, f.groups = e->plan.groups;
    f.step_blocks = lr_stream_step_blocks(e->cfg.n_chains);
    f.sync = (lr_stream_sync*)(e->ws + e->lay.xchg);
    f.tables2 = (double2*)(e->ws + e->lay.xchg + LR_STREAM_SYNC_BYTES);
    f.status = (unsigned int*)(e->ws + e->lay.status);
    const int grid = f.step_blocks + f.tiles * f.groups;
    const size_t lds_bytes = lr_stream_lds_bytes(e->plan);
    for (int64_t done = 0; done < n_iters || query;) {
        const int64_t n = (n_iters - done > 4096) ? 4096 : n_iters - done;   // keep single launches short
        f.n_iters = n;
        if (!query) {
            // the counters start at zero in every launch
            const hipError_t he = hipMemsetAsync(f.sync, 0, LR_STREAM_SYNC_BYTES, stream);
            if (he != hipSuccess) return (int)he;
        }
        int rc;
        switch (e->plan.H) {
            case 40: rc = lr_stream_launch_h<40>(e, a, f, grid, lds_bytes, query, stream); break;
            case 72: rc = lr_stream_launch_h<72>(e, a, f, grid, lds_bytes, query, stream); break;
            case 136: rc = lr_stream_launch_h<136>(e, a, f, grid, lds_bytes, query, stream); break;
            case 264: rc = lr_stream_launch_h<264>(e, a, f, grid, lds_bytes, query, stream); break;
            default: rc = LR_ERR_SIZE;
        }
        if (rc || query) return rc;
        done += n;
    }
    return LR_OK;
}
