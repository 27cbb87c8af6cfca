// lr_engine.h - host-side engine object shared by the translation units of the RJMCMC engines
// (lr_mcmc.hip: launch-based and two- / four-chain persistent kernels, lr_spec.hip: speculative team kernel,
// lr_pack.hip: packing of the lineages for the persistent scans).
#pragma once
#include "lr_internal.h"
#include "lr_scan.h"
#include "lr_step.h"

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
    int n_slots;          // scanner waves striding over the groups: 14 (four-chain kernel; 12 with helper waves) or 8 (two-chain kernel)
    int help_trips;       // four-chain kernel with helper waves: trips of the scan each helper wave makes before its hand-over arrives
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
    bool p4_help;             // four-chain kernel: the form with helper waves - latched by lr_set_shares (init / restore), so
                              // that the form, its shares and the sums carried between launches belong together for a whole run
    bool p4_spec;             // ... whose steppers speculate on rejection (lr_chain_step_respec); latched with p4_help
    bool packed_scan;         // launch-based engine whose scan kernel reads the PACKED lineages (lr_packscan.hip) instead of ts / te:
                              // planned by lr_mcmc_query_layout (lay.packed_scan), dropped by init / restore if the packing
                              // refuses the input (unsorted beyond LR_MAX_RUNS runs)
    bool streaming;           // launch-based plan, but the iterations run inside ONE resident kernel (lr_stream.hip): latched by
                              // lr_mcmc_create, which asks the device in use whether the whole grid fits it at once
    hipEvent_t fork;
    hipEvent_t ev0, ev1;      // timing events of lr_mcmc_time_steps / lr_mcmc_time_scan, created once
};


// spare zero-filled 16-byte groups behind the packed lineages (group 0 bytes = sentinel table entries, contribution 0):
// room for the trips the four-chain kernel moves between waves and for its prefetch past the end
#define LR_P4_MAX_GIVE 64
// sized for the widest stride (16 scanner waves x 64 lanes): the takers' extra trips reach group (k_tot + give) * stride
#define LR_IDX_SPARE ((LR_P4_MAX_GIVE + 2) * 1024)
// the packing keeps the caller's order and starts a new group wherever the birth bin changes: lineages sorted by birth
// time give at most n_bins + 2 such runs; more than LR_MAX_RUNS of them (unsorted input) are refused
#define LR_MAX_RUNS (1ll << 20)

static inline long long lr_groups_alloc(long long n_lineages) {
    const long long runs = n_lineages < LR_MAX_RUNS ? n_lineages : LR_MAX_RUNS;
    // (unit resolution: LR_SLOTS slots of one or two lineages per group - all singles in the worst case)
    return (n_lineages + LR_SLOTS - 1) / LR_SLOTS + runs + LR_IDX_SPARE;
}

// four-chain kernel: bytes per block of the sums a launch leaves for the next ([16 waves][2] doubles + a valid word)
#define LR_P4_CARRY_BYTES 512

// widest table class of the persistent kernels (the launch-based scans are instantiated up to H = 264)
#define LR_H_WIDE 520

// ---- speculative team engine (lr_spec.h / lr_spec.hip) ----
#ifndef LR_SPEC_THREADS
#define LR_SPEC_THREADS 768   /* a team per pair: 12 waves = 4 candidate + 8 scanner waves, 3 per SIMD = 168 VGPRs each */
#endif
#ifndef LR_SPEC_THREADS_SINGLE
#define LR_SPEC_THREADS_SINGLE 768    /* a team per chain: 12 waves = 2 candidate + 2 helper + 8 scanner waves.  (1024 threads = 12
                                         scanner waves fit without spills - with the table build on the helper waves no role needs
                                         more than ~105 VGPRs - and take 5 % off a 100k-lineage scan on one CU, but every team
                                         exchange and every short scan got 3-8 % slower: measured round 3, not used) */
#endif
#ifndef LR_SPEC_SCAN_UNROLL
#define LR_SPEC_SCAN_UNROLL 1
#endif
#ifndef LR_SPEC_SCAN_PRIO
#define LR_SPEC_SCAN_PRIO 2   /* s_setprio of the scanner waves of a team (0 = leave the default) */
#endif
#define LR_TEAM_MAX 8
#define LR_SPEC_GRANULES 16      /* 8-byte granules reserved per block and parity: one 128-byte line */
#define LR_SPEC_TIMEOUT_TICKS 200000000ull   /* 2 s of the 100 MHz wall clock */

// Four-chain kernel: helper waves (lr_persist4_kernel's HELP) under the RJ sampler at unit resolution; LR_P4_HELP = 0: the
// form with fourteen scanner waves (A/B runs).  Evaluated ONLY by lr_set_shares, i.e. at lr_mcmc_init / lr_mcmc_restore
// (a test process that runs both forms re-initialises between them); everything else reads e->p4_help.
static inline bool lr_p4_help_choice(const lr_engine* e) {
    const char* env = getenv("LR_P4_HELP");
    if (e->lay.persistent != 2 || e->plan.unit == LR_TAB_PAIRGEN || e->cfg.sampler != 0 || e->plan.H > 264) return false;
    return env ? atoi(env) != 0 : true;
}

// ... and whether its steppers speculate on rejection (lr_persist4_kernel's SPEC): LR_P4_SPEC = 1.  OFF by default: the
// trajectories are bit-identical, but cfg4 runs 7.3 us per iteration against 6.3 - a phase of the four-chain kernel is
// bound by what SIMDs 0, 1 issue (a stepper + three scanners each) and by the helper's build behind the scanners' LDS
// gathers, not by the wait for the hand-over that the speculation removes (in-kernel stamps: profiles/EXPERIMENTS.md)
static inline bool lr_p4_spec_choice(const lr_engine* e) {
    const char* env = getenv("LR_P4_SPEC");
    return lr_p4_help_choice(e) && (env ? atoi(env) != 0 : false);
}

// Which instantiation of the speculative kernel an engine runs (lr_spec.h): 0 = a team per pair; a team per chain: 1 = in
// a team of several blocks, 2 = on its own CU with the pair planes of a candidate's table derived by the scanner waves for
// the ONE table that becomes pending (short scans - at most 3 trips per scanner lane -: the helper waves' build is the
// longer path of an iteration), 3 = on its own CU with the planes derived by the helper wave that builds the table (long
// scans: the scan is).  LR_SPEC_PLANES_BY_SCANNERS = 0 / 1 overrides the rule between 2 and 3.
static inline int lr_spec_mode(const lr_engine* e) {
    if (e->lay.spec_chains_per_team != 1) return 0;
    if (e->lay.team_blocks > 1) return 1;
    const double trips = (double)e->n8 / 64.0 / 8.0;
    const char* env = getenv("LR_SPEC_PLANES_BY_SCANNERS");
    const bool by_scanners = env ? atoi(env) != 0 : trips <= 3.0;
    return by_scanners ? 2 : 3;
}

// ---- resident streaming engine (lr_stream.hip) ----
#define LR_STEP_WAVES_PER_BLOCK (LR_SCAN_THREADS / LR_WAVE)
// counters of one launch, in the workspace at lr_mcmc_layout.xchg, each on a cache line of its own; the second table buffer
// follows LR_STREAM_SYNC_BYTES behind
#define LR_STREAM_COPIES 64
#define LR_STREAM_MAX_CHAINS 16
struct lr_stream_sync {
    unsigned long long arrived;      // scanner blocks that have stored their tile's partial sums, over the launch's iterations
    unsigned long long pad0[15];
    // what the stepper wave of chain c has published, LR_STREAM_COPIES times - one 128-byte line per copy holds all chains'
    // words, and scanner block b polls copy b % LR_STREAM_COPIES (a thousand blocks polling ONE line kept the memory channel
    // that holds it so busy that the blocks still streaming took twice as long):
    //   bits 40-63  early: iterations whose early table stands (the step taken ahead)
    //   bits 16-39  ready: iterations decided - the next iteration's table is final
    //   bits  0-15  changed: +1 before and +1 behind every table written AGAIN (an accepted proposal); a scanner block
    //               that staged the early tables stages again if this moved since
    unsigned long long word[LR_STREAM_COPIES][LR_STREAM_MAX_CHAINS];
};
#define LR_STREAM_SYNC_BYTES (128 + LR_STREAM_COPIES * 128 + 128)
static_assert(sizeof(lr_stream_sync) + 128 == LR_STREAM_SYNC_BYTES && LR_STREAM_SYNC_BYTES % 256 == 0, "counter line + a line per copy");
bool lr_stream_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p);
void lr_stream_plan(const lr_mcmc_config* cfg, lr_scan_plan* p, int cus);
// n_iters iterations in launches of at most 4096; query = true only asks whether the grid fits the current device at once
int lr_launch_stream(lr_engine* e, const lr_step_args& a, int64_t n_iters, bool query, hipStream_t stream);

// ---- packed scan of the launch-based engine (lr_packscan.hip) ----
bool lr_packscan_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p);
void lr_packscan_plan(const lr_mcmc_config* cfg, lr_scan_plan* p, int cus);
int lr_packscan_pairs(const lr_scan_plan& p, int n_chains);
int lr_launch_packscan(const lr_engine* e, int base, int count, hipStream_t stream);

// lr_mcmc.hip
lr_step_args lr_make_args(const lr_engine* e);
// lr_pack.hip: (re)builds the packed lineages in the workspace and the scanner-wave shares; blocks on `stream` once
int lr_pack_lineages(lr_engine* e, hipStream_t stream);
long long lr_pack_tmp_bytes(long long n_lineages);
// lr_spec.hip: n_iters iterations of the speculative team kernel
int lr_launch_spec(lr_engine* e, const lr_step_args& a, const lr_packed_lineages& pk, int64_t n_iters, hipStream_t stream);
