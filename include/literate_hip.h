/*
 * literate_hip.h - C ABI of libliterate_hip.so: the MI355X (gfx950) kernels behind the
 * LiteRate RJMCMC birth-death likelihood path.
 *
 * The reference (dsilvestro/LiteRate) is pure Python/numpy and has no FFI layer; the seams
 * this library plugs into are the Python call signatures listed per entry point below
 * (LRF = LiteRateForward.py, lib = literate_library.py, DD = DDRate.py, BDIx =
 * other/LiteRateBDI_ext.py).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor.data_ptr()) unless it is
 *     marked "host"; the caller owns every buffer, nothing is allocated behind its back;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     asynchronous on it and returns after enqueueing;
 *   - return value: 0 = ok, <0 = invalid argument (LR_ERR_*), >0 = a hipError_t;
 *   - arithmetic is IEEE fp64, counts are int64 (as numpy in the reference);
 *   - results are bitwise reproducible: all reductions run in a fixed order, no float atomics.
 */
#ifndef LITERATE_HIP_H
#define LITERATE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_OK 0
#define LR_ERR_NULL (-1)      /* a required pointer is NULL                              */
#define LR_ERR_SIZE (-2)      /* a size / count is out of range                          */
#define LR_ERR_MODEL (-3)     /* unknown model id or missing model input                 */
#define LR_ERR_WORKSPACE (-4) /* workspace too small                                     */
#define LR_ERR_T0 (-5)        /* bin origin must be integer valued                       */
#define LR_ERR_STATE (-6)     /* engine used before lr_mcmc_init / after destroy         */
#define LR_ERR_ORDER (-7)     /* persistent engines: lineages not grouped by birth bin (sort them by ts) */

#define LR_KMAX 32     /* max number of rates per process held on the device (reference: unbounded) */
#define LR_ROW 64      /* padded row length of per-chain state arrays                                 */
#define LR_MAX_BINS 4094

/* model ids = the reference's -model_BDI (LRF:388, 421-431) */
#define LR_MODEL_BD 0       /* BDI_partial_lik, birth-death      (LRF:150-162) */
#define LR_MODEL_ID 1       /* BDI_partial_lik, immigration-death               */
#define LR_MODEL_KEIDING 2  /* BD_lik_Keiding                     (LRF:137-148) */
#define LR_MODEL_KEIDING_DEAD 3 /* Keiding, death half on te<end_time (LRF:141-142, 529-546) */

int lr_version(void);

/* ---- A1/A2: sufficient statistics ---------------------------------------------------------
 * Replaces precompute_events + get_br (lib:74-85; LRF:111-123) evaluated for n_windows windows
 * at once (the loop LRF:519-523 / lib create_bins:231-245).  Window w is [win_lo[w], win_hi[w]]:
 *   sp_events[w] = #{ts >= lo && ts <  hi}
 *   ex_events[w] = #{te >  lo && te <= hi}
 *   br_length[w] = sum_i max(0, min(te_i,hi) - max(ts_i,lo))
 * Counts are exact; br_length is summed in a fixed order.                                    */
int64_t lr_bin_events_workspace_bytes(int64_t n, int32_t n_windows);
int lr_bin_events(const double* ts, const double* te, int64_t n,
                  const double* win_lo, const double* win_hi, int32_t n_windows,
                  int64_t* sp_events, int64_t* ex_events, double* br_length,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* The same statistics for the n_bins UNIT windows [t0 + w, t0 + w + 1] the reference always bins into (LRF:519-523:
 * `for i in range(int(min(ts)), int(max(te)))`; lib create_bins:231-245), t0 integer valued: ONE pass over ts / te,
 * 16 bytes per lineage - SURVEY 8(b)'s lr_bin_events(ts, te, n, t0, n_bins, ...).  Counts are exact; br_length[w] is the
 * EXACT sum of the per-lineage overlaps get_br forms (lib:74-79), rounded to fp64 once (integer accumulation: order
 * free, bitwise reproducible); on year-resolution input it equals the reference's value bit for bit.              */
int64_t lr_bin_unit_events_workspace_bytes(int64_t n, int32_t n_bins);
int lr_bin_unit_events(const double* ts, const double* te, int64_t n, double t0, int32_t n_bins,
                       int64_t* sp_events, int64_t* ex_events, double* br_length,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ---- A3: rate index -----------------------------------------------------------------------
 * Replaces get_rate_index + fancy indexing L[indL] (LRF:125-135, 262, 306): expands K segment
 * rates to one rate per unit bin.  rates [C,kmax], times [C,kmax+1] ascending, K [C].
 * mode 0: floor(times) (LRF:262,272,277-278); mode 1: round-half-even (LRF:129, initial call
 * LRF:224-225).  Bin b of chain c gets segment j with e_j <= b < e_{j+1},
 * e_j = int(mode(times[j])) - int(mode(times[0])).                                            */
int lr_expand_rates(const double* rates, const double* times, const int32_t* K, int32_t kmax,
                    int32_t n_chains, int32_t n_bins, int32_t mode,
                    double* rate_bins /* [C,n_bins] */, void* stream);

/* ---- A4/A5/A6: batched per-lineage log-likelihood -----------------------------------------
 * Replaces calc_likelihood(L[indL], M[indM]) (LRF:226, 306, 430-431) for n_chains states at
 * once, in the per-lineage form of BD_partial_lik/get_BDlik (BDIx:124-146): one pass over
 * ts/te per group of chains, unit bins [t0+b, t0+b+1), b < n_bins, t0 integer valued
 * (= int(min ts), LRF:519).  lam_bins/mu_bins [C,n_bins] are the per-bin rates (the
 * L_acc_vec/M_acc_vec arguments).  br_length [n_bins] is required for models 0/1 (it is the
 * k = br_length_bin of LRF:154), end_time for model 3.  out_loglik [C].
 * Few states on few lineages (n <= 2^18, C <= 64, n * C <= 2^21: the reference's own use, one state per iteration) take
 * ONE launch - a block per state builds its table in LDS and walks all lineages (LR_LOGLIK_SMALL=0: never); lam_bins,
 * mu_bins and out_loglik may then as well be pinned HOST memory (device-accessible), which saves the caller two copies
 * and the synchronisation (literate_amd/ops.py LoglikSession).                                  */
int64_t lr_bd_loglik_workspace_bytes(int64_t n, int32_t n_bins, int32_t n_chains, int32_t model);
/* measurement hook: the launch shape lr_bd_loglik_batch uses for these sizes - out[0] = Cb, the chains one pass over
 * ts / te scores (the call makes ceil(n_chains / Cb) passes = 16 B x n x passes of algorithmic HBM reads), out[1] =
 * lineage tiles per pass, out[2] = table entries reserved per chain and side, out[3] = passes.  out: host int32[4]. */
int lr_bd_loglik_plan(int64_t n, int32_t n_bins, int32_t n_chains, int32_t model, int32_t* out /* host */);
int lr_bd_loglik_batch(const double* ts, const double* te, int64_t n, double t0, int32_t n_bins,
                       const double* lam_bins, const double* mu_bins, int32_t n_chains,
                       int32_t model, const double* br_length, double end_time,
                       double* out_loglik, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- A7/A8: proposal scoring with explicit draws ------------------------------------------
 * Replaces update_multiplier_freq (LRF:165-176 == lib update_multiplier_proposal_vec:156-165),
 * add_shift_RJ_weighted_mean (LRF:29-47) and remove_shift_RJ_weighted_mean (LRF:49-69), one
 * move per chain, randomness supplied by the caller:
 *   move[c] = 0 multiplier: draws[c, 0:K) = binomial mask (0/1), draws[c, kmax:kmax+K) = uniforms
 *   move[c] = 1 add shift : index[c] = interval (0..K-1), draws[c,0] = offset in it, draws[c,1] = Beta(10,10) variate
 *   move[c] = 2 remove    : index[c] = shift (1..K-1)
 * out_score = Hastings term (multiplier) or log q + log Jacobian (RJ).                        */
int lr_rj_propose_score(const double* rates /* [C,kmax] */, const double* times /* [C,kmax+1] */,
                        const int32_t* K, int32_t kmax, int32_t n_chains,
                        const int32_t* move, const int32_t* index,
                        const double* draws /* [C,2*kmax] */, double mult_d,
                        double* out_rates, double* out_times, int32_t* out_K, double* out_score,
                        void* stream);

/* ---- A10: priors --------------------------------------------------------------------------
 * out[c] = prior_gamma(rates[c,:K], a=shape, b=gamma_rate[c]) (LRF:201-202)
 *        + Poisson_prior(K[c], poi_rate[c]) if poi_rate != NULL (LRF:198-199).               */
int lr_log_priors(const double* rates, const int32_t* K, int32_t kmax, int32_t n_chains,
                  double shape, const double* gamma_rate, const double* poi_rate,
                  double* out, void* stream);

/* ---- A12: DDRate rates --------------------------------------------------------------------
 * Replaces the rate half of likelihood_function (DD:71-100): args [C,8] =
 * [l_max,k,x0,div_0,L,m_max,nuB,nuD] -> per-bin birth/death rates, niche, niche fraction
 * (each [C,n_bins]); the likelihood half is lr_bd_loglik_batch(model 2) on those rates.      */
int lr_dd_rates(const double* args, const double* DT, int32_t n_bins, int32_t n_chains,
                int32_t m_birth, int32_t m_death,
                double* birth_rates, double* death_rates, double* niche, double* niche_frac,
                void* stream);

/* ---- the reference's other rate maps onto the same per-bin likelihood ---------
 * lr_ddv2_rates replaces the rate half of DDRatev2.py likelihood_function (DDRatev2.py:73-104):
 * args [C,9] = [l_f,l_mul,k,x0,div_0,L,m_mul,nuB,nuD]; m_birth 0..2, m_death <=0 (rates of 1) / 1 / 2.
 * lr_trend_rates replaces trend_rate.py likelihood_function's rate half (trend_rate.py:73-88):
 * args [C,6] = [l_min,m_min,alpha,beta,delta,gamma], trend [n_bins] = the normalised covariate
 * (parse_trend_data, trend_rate.py:58-69).  Likelihood half: lr_bd_loglik_batch(model 2).     */
int lr_ddv2_rates(const double* args, const double* DT, int32_t n_bins, int32_t n_chains,
                  int32_t m_birth, int32_t m_death,
                  double* birth_rates, double* death_rates, double* niche, double* niche_frac,
                  void* stream);
int lr_trend_rates(const double* args, const double* trend, int32_t n_bins, int32_t n_chains,
                   int32_t const_birth, int32_t const_death,
                   double* birth_rates, double* death_rates, void* stream);

/* Binned Keiding halves of C per-bin rate vectors, out[c] = sum_b log(rate[c,b]) * events[b] - rate[c,b] * DT[b]
 * (DD:86, 101; trend_rate.py:82, 89; the same expression as BD_lik_Keiding LRF:137-148 on create_bins statistics):
 * the likelihood_birth / likelihood_death columns of the DDRate-family logs.                                       */
int lr_binned_keiding(const double* birth_rates, const double* death_rates, const int64_t* n_spec,
                      const int64_t* n_exti, const double* DT, int32_t n_bins, int32_t n_chains,
                      double* out_birth, double* out_death, void* stream);

/* ---- discrete-time birth-death lineage simulator ----------------------------------
 * The scheme of simulateRateABC.v2.py:103-234 and of notebook 4's Simulator / Population: n_start lineages born at
 * step 0; at every step t < n_steps each living lineage draws one uniform r (Philox keyed by (seed, lineage slot),
 * counter (t, 24, 0)): r < lambda_t spawns a lineage born at t, lambda_t <= r < lambda_t + mu_t kills it at t.
 * mode 0: lambda_t = lam_steps[t], mu_t = mu_steps[t] (per-step probabilities, i.e. rate / scale);
 * mode 1: notebook-4 diversity dependence, lambda = max(0, l0 - l0 D/K), mu = max(0, m0 + m0 D/K), both / scale;
 * mode 2: simulateRateABC.v2.py:153-154, lambda = max(0, l0 - (l0-m0) D/K), mu = max(0, m0 + (l0-m0) D/K), / scale;
 * D = living lineages at the start of the step.  Outputs: ts/te [capacity] = birth / death STEP of every lineage
 * (te = n_steps: extant), counters[0] = lineages, counters[1] = living at the end, counters[2] = 1 if `capacity`
 * was hit (later births were dropped); alive_trace[n_steps] (may be NULL) = D per step.  workspace: 64 bytes.   */
int lr_simulate_bd(const double* lam_steps, const double* mu_steps, int32_t n_steps, int32_t mode,
                   double l0, double m0, double K, double scale, int64_t n_start, int64_t capacity,
                   uint64_t seed, double* ts, double* te, int64_t* counters /* [4] */,
                   int64_t* alive_trace, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the text form of the logs (host only, no GPU) ---------------------------------
 * The reference writes every number through Python's csv module (LRF:334-359, DD:236-238): str(float), the shortest
 * decimal string that reads back to the same double, "24.0" / "1e-05" / "1.5e+16" by Python's rules.  lr_format_rows
 * writes n_rows tab-separated lines into `out`: row i = vals[row_start[i] .. row_start[i + 1]); a value in column
 * c < 64 of its row whose bit is set in int_cols is written as an integer; lines end "\n", or "\r\n" with
 * LR_FORMAT_CRLF in flags (csv.writer's default, which DDRate.py and trend_rate.py write).  cap must be at least
 * 26 * (number of values) + 2 * n_rows.  Returns the number of bytes written (>= 0) or LR_ERR_*.  Thread-safe.       */
#define LR_FORMAT_CRLF 1
int64_t lr_format_rows(const double* vals, const int64_t* row_start, int64_t n_rows, uint64_t int_cols, int32_t flags,
                       char* out, int64_t cap);

/* ---- A11: fused multi-chain RJMCMC --------------------------------------------------------
 * Replaces runMCMC (LRF:216-373) for n_chains independent chains.  Per iteration: one scan of
 * the lineage arrays scoring every chain's proposal, then one chain-step kernel (reduce,
 * Metropolis-Hastings accept, trace write, next proposal, next tables).  Randomness: Philox4x32-10
 * addressed by (iteration, purpose, index), keyed by (seed, chain_offset + chain).            */
typedef struct lr_mcmc_config {
    int64_t n_lineages;
    int32_t n_bins;
    int32_t n_chains;
    int32_t model;            /* LR_MODEL_*                                         */
    int32_t const_rates;      /* -const_rates       (LRF:386, 274)                  */
    int32_t const_death_rate; /* -const_death_rate  (LRF:387, 243-252)              */
    int32_t use_rate_HP;      /* -use_rate_HP       (LRF:395, 285)                  */
    int32_t s_freq;           /* -s sampling frequency (LRF:382, 321)               */
    int32_t n_trace_slots;    /* capacity of the trace buffer in samples            */
    double poisson_HP;        /* -Poisson_prior     (LRF:396, 220-221)              */
    double update_fraction;   /* -update_fraction   (LRF:399)                       */
    double t0;                /* first bin edge = int(min ts)                       */
    double start_time;        /* min(ts)  (LRF:473)                                 */
    double end_time;          /* max(te)  (LRF:474)                                 */
    uint64_t seed;
    int64_t chain_offset;     /* global index of local chain 0 (multi-GPU sharding) */
    /* unit-resolution data: 1 asserts that EVERY lineage has ts - floor(ts) == frac_birth and
     * te - (ceil(te) - 1) == frac_death (true for year-resolution input + death_jitter, i.e. every
     * dataset the reference ships: 0 and 0.5).  The fractions are then folded into the lookup
     * tables (8-byte entries, half the LDS traffic per lineage).  0 = general times.            */
    int32_t unit_resolution;
    int32_t engine_mode;      /* 0 = auto, 1 = launch-per-iteration engine (fused, pipelined), 2 = a persistent kernel
                               * (the library picks which), 3 = four chains per block, 4 = two chains per block,
                               * 5 = speculative team kernel, 6 = the launch-based plan with its iterations inside the
                               * resident streaming kernel where it applies, 7 = the launch-based engine scanning the
                               * PACKED lineages (lr_mcmc_layout.persistent / .streaming / .packed_scan tell what runs) */
    double frac_birth;
    double frac_death;
    /* ---- sampler 1: the DDRate.py Metropolis-Hastings loop (DD:124-241) on the same engine -------------
     * model must be LR_MODEL_KEIDING, br_length = DT of create_bins (lib:231-257), t0 = ORIGIN, n_bins =
     * N_TIME_BINS.  A chain's state is the parameter vector [l_max,k,x0,div_0,L,m_max,nuB,nuD] (DD:161) in
     * lanes 0..7 of the rate row; a trace row is [it, posterior, likelihood, prior, args[8]].               */
    int32_t sampler;          /* 0 = runMCMC (LRF), 1 = DDRate, 2 = trend_rate (below) */
    int32_t m_birth;          /* -m_birth (DD:25); sampler 2: -const_B flag          */
    int32_t m_death;          /* -m_death (DD:26); sampler 2: -const_D flag          */
    int32_t team_request;     /* speculative kernel: bits 0-7 blocks per team (1, 2, 4, 8), bits 8-15 chains per team (1 or
                               * 2); 0 in either field = the library chooses                                            */
    double dd_present;        /* PRESENT - as create_bins returns it (DD:36)        */
    double dd_init_death;     /* -fix_death (DD:27, 156)                            */
    /* sampler 2: the trend_rate.py loop (trend_rate.py:102-196): parameters [l_min,m_min,alpha,beta,delta,gamma],
     * br_length = the normalised covariate TREND[n_bins] (parse_trend_data, trend_rate.py:58-69), t0 / n_bins as
     * for sampler 1; a trace row is [it, posterior, likelihood, prior, args[6]].                              */
} lr_mcmc_config;

/* where things live inside the engine workspace (byte offsets), for zero-copy host views */
typedef struct lr_mcmc_layout {
    int64_t state_f64;    /* [C, LR_STATE_ROWS, LR_ROW] doubles  (rows: see LR_ROW_* below)  */
    int64_t state_i32;    /* [C, LR_ISTATE_ROWS, LR_ROW] int32                                */
    int64_t bin_consts;   /* [n_bins] doubles: log(br_length) (models 0/1)                    */
    int64_t lineage_idx;  /* [groups] 16 bytes: packed table entries of the lineages (persistent engines): a group = up to
                           * 14 consecutive lineages of one birth bin in 7 slots of one or two lineages: byte 0 birth
                           * index, byte 1 count, then seven 16-bit entry byte offsets (csrc/lr_pack.hip)                  */
    int64_t args_blob;    /* 1 KiB: kernel arguments of the persistent engine, kept in device memory          */
    int64_t tables;       /* [C, table_stride] double2                                        */
    int64_t partials;     /* [C, tiles rounded up to 16] doubles: a chain's tile partials are one row    */
    int64_t trace;        /* [n_trace_slots, C, LR_TRACE_W] doubles                           */
    int64_t total_bytes;
    int32_t table_stride; /* double2 entries per chain                                         */
    int32_t tiles;
    int32_t chains_per_block;
    int32_t trace_width;
    int32_t n_parts;      /* independent chain partitions, each on its own stream                  */
    int32_t pipelined;    /* 1: each partition runs the fused scan|step schedule over two halves   */
    int32_t persistent;   /* 0: launch-per-iteration engine; 1 / 2: persistent kernel, 2 / 4 chains per block;
                           * 3: speculative team kernel (a chain pair per team of team_blocks blocks)             */
    int32_t reserved1;    /* threads per block of the persistent kernel (512 / 1024), 0 for the launch-based engine */
    int64_t status;       /* engine status word (uint32): 0 ok, 1 = a team exchange of the speculative kernel timed out;
                           * the uint32 behind it is the warning word (lr_mcmc_warnings)                                */
    int64_t xchg;         /* partial-sum exchange granules of the speculative kernel's teams (team_blocks > 1); under the
                           * four-chain kernel (persistent == 2) the scan sums a launch leaves for the next one, 512 bytes
                           * per block - engine scratch either way: not part of a run, cleared by init / restore        */
    int32_t team_blocks;  /* blocks (= CUs) that share one chain pair, each scanning 1/team_blocks of the lineages     */
    int32_t table_mode;   /* 0 chain-major general tables, 1 unit-resolution pair tables, 2 pair-general tables (persistent
                           * engines on general lineage times: in-bin fractions packed as 32-bit fixed point)             */
    int64_t lineage_frac; /* [3][groups] uint4: fe' of the 7 slots + sum of fs (table_mode 2)                               */
    int64_t pack_tmp;     /* scratch of the lineage packing (two int32 per lineage + the scans' temporary storage)       */
    int32_t spec_chains_per_team; /* speculative kernel: chains a team of blocks owns - 2 (a pair) or 1; 0 for the other engines */
    int32_t streaming;    /* 1 (persistent == 0 only): too few chains for the pipelined schedule - the iterations run inside one
                           * RESIDENT kernel (csrc/lr_stream.hip: scanner blocks + a stepper wave per chain, the step taken
                           * ahead on the assumption that the pending proposal is rejected) wherever its grid fits the device
                           * at once; xchg then holds the launch's counters and the second table buffer                      */
    int32_t packed_scan;  /* 1 (persistent == 0 only): the launch-based engine scans the PACKED lineages (lineage_idx: 1.14 bytes
                           * per lineage at unit resolution) once per iteration for all its chains (csrc/lr_packscan.hip)
                           * instead of ts / te (16 bytes per lineage); unit-resolution data, too few chains for the
                           * pipelined schedule.  engine_mode 1 keeps the scan of ts / te                               */
    int32_t reserved3;
} lr_mcmc_layout;

/* rows of the fp64 state block (element j of a row lives in lane j of the chain's wave) */
#define LR_ROW_L 0      /* accepted birth rates       [K_l]   */
#define LR_ROW_M 1      /* accepted death rates       [K_m]   */
#define LR_ROW_TL 2     /* accepted birth shift times [K_l+1] */
#define LR_ROW_TM 3
#define LR_ROW_PL 4     /* proposed ... */
#define LR_ROW_PM 5
#define LR_ROW_PTL 6
#define LR_ROW_PTM 7
#define LR_ROW_SCALARS 8 /* see LR_S_* */
#define LR_STATE_ROWS 9
/* scalar slots inside LR_ROW_SCALARS */
#define LR_S_LIKA 0
#define LR_S_PRIORA 1
#define LR_S_PRIORPOIA 2
#define LR_S_GRATE_L 3   /* Gamma_rate[0] (LRF:222, 286) */
#define LR_S_GRATE_M 4
#define LR_S_POI 5       /* Poi_lambda_rjHP (LRF:220-221, 284) */
#define LR_S_HASTING 6   /* of the pending proposal */
#define LR_S_PRIOR_P 7
#define LR_S_PRIORPOI_P 8
#define LR_S_CONST_P 9   /* model constant of the pending proposal (model 1)   */
#define LR_S_CONST_A 10
#define LR_S_LIK_P 11    /* last evaluated proposal log-likelihood (diagnostic) */
#define LR_S_LOG_G0 12   /* cached log(Gamma_rate[0]), log(Gamma_rate[1]), log(Poi_lambda_rjHP) */
#define LR_S_LOG_G1 13
#define LR_S_LOG_POI 14
#define LR_S_LOG_U 15     /* log of the acceptance uniform of the pending proposal's iteration (drawn one step early) */
/* rows of the int32 state block */
#define LR_IROW_EL 0     /* accepted birth bin edges (ints, relative to bin 0) [K_l+1] */
#define LR_IROW_EM 1
#define LR_IROW_PEL 2
#define LR_IROW_PEM 3
#define LR_IROW_SCALARS 4 /* see LR_I_* */
#define LR_ISTATE_ROWS 5
#define LR_I_KL 0
#define LR_I_KM 1
#define LR_I_PKL 2
#define LR_I_PKM 3
#define LR_I_GIBBS 4     /* pending proposal is a Gibbs step (LRF:283-287)     */
#define LR_I_INVALID 5   /* pending proposal fails the LRF:290 guard or K cap  */
#define LR_I_IT_LO 6     /* next iteration number (64 bit)                      */
#define LR_I_IT_HI 7
#define LR_I_ACCEPTED 8  /* number of accepted proposals so far                 */
#define LR_I_MOVE 9      /* kind of the pending proposal: 0 L-mult 1 L-times 2 M-mult 3 M-times 4 RJ 5 Gibbs */
#define LR_I_NEXT_LO 10  /* next iteration number that writes a trace row (64 bit) and its slot     */
#define LR_I_NEXT_HI 11
#define LR_I_SLOT 12

/* trace row (one per chain per sample; columns 0..12 are the _mcmc.log columns LRF:496-502
 * without the adequacy triple, then the _sp_rates / _ex_rates rows LRF:354-359):
 *   [it, posterior, likelihood, prior, lambda_avg, mu_avg, K_l, K_m, root_age, death_age,
 *    gamma_rate_hp_BI, gamma_rate_hp_D, poisson_rate_hp,
 *    L[0..KMAX), tL interior [0..KMAX-1), M[0..KMAX), tM interior [0..KMAX-1)]               */
#define LR_TRACE_HEAD 13
#define LR_TRACE_W (LR_TRACE_HEAD + 2 * (2 * LR_KMAX - 1))

typedef struct lr_engine lr_engine;

int lr_mcmc_query_layout(const lr_mcmc_config* cfg /* host */, lr_mcmc_layout* out /* host */);
int lr_mcmc_create(const lr_mcmc_config* cfg /* host */, const double* ts, const double* te,
                   const double* br_length /* [n_bins], models 0/1, else NULL */,
                   void* workspace, int64_t workspace_bytes, lr_engine** out /* host */);
/* init_state: NULL = the CLI's initial state (K=1, Gamma(2,2) rates, LRF:580-583) drawn from the
 * chain's Philox stream; else device arrays L[C,kmax], M[C,kmax], tL[C,kmax+1], tM[C,kmax+1],
 * KL[C], KM[C] (the runMCMC argument, LRF:218).  Evaluates likA/priorA (LRF:224-230) and
 * prepares the proposal of iteration 0.                                                       */
int lr_mcmc_init(lr_engine* e, const double* L, const double* M, const double* tL,
                 const double* tM, const int32_t* KL, const int32_t* KM, int32_t kmax, void* stream);
int lr_mcmc_steps(lr_engine* e, int64_t n_iters, void* stream);
/* Resume (no reference counterpart: the reference cannot resume).  Between two lr_mcmc_steps
 * calls the whole run - accepted states, pending proposals and their tables, iteration counters, trace rows -
 * is the workspace; the Philox draws are addressed by (seed, chain, iteration), so a run continued from a copy of
 * the workspace is bit-identical to an uninterrupted one.  The caller copies a workspace saved from an engine
 * with the SAME configuration (hence the same lr_mcmc_layout) into this engine's workspace, then calls this
 * instead of lr_mcmc_init: it rebuilds what holds device addresses or derives from the data (argument blob,
 * log(br_length), packed lineage indices), forgets what earlier launches left in the engine's scratch regions (status,
 * warnings, carried scan sums) and leaves the chains alone.  Chain state written into the workspace from outside
 * must always be followed by this call.                                                                          */
int lr_mcmc_restore(lr_engine* e, void* stream);
/* measurement hook (bench.py roofline): average duration in ms of `reps` back-to-back launches of
 * the engine's lineage-scan kernel on `stream`, timed with HIP events recorded on that stream.
 * Blocks until the launches finish; re-scores the pending proposal, so chain state is unchanged. */
int lr_mcmc_time_scan(lr_engine* e, int32_t reps, float* avg_ms /* host */, void* stream);
/* lr_mcmc_steps(n_iters) bracketed by HIP events on `stream`; blocks; *total_ms = elapsed device time. */
int lr_mcmc_time_steps(lr_engine* e, int64_t n_iters, float* total_ms /* host */, void* stream);
/* Blocks until `stream` is idle and copies the engine status word to *status (host): 0 = ok, 1 = a team exchange of the
 * speculative kernel timed out (its blocks were not all resident within two seconds) and the run is void.          */
int lr_mcmc_status(lr_engine* e, int32_t* status /* host */, void* stream);
/* Blocks until `stream` is idle and copies the engine's warning word to *warnings (host): a bit set of LR_WARN_*.
 * LR_WARN_KCAP: at least one add-shift move (LRF:29-47) was proposed from a state that already holds LR_KMAX rates and
 * was rejected for that reason alone - the reference has no such cap, the posterior on the number of shifts is
 * truncated at LR_KMAX.  Cleared by lr_mcmc_init / lr_mcmc_restore.                                                  */
#define LR_WARN_KCAP 2
int lr_mcmc_warnings(lr_engine* e, int32_t* warnings /* host */, void* stream);
/* measurement hook: name of the kernel lr_mcmc_steps spends its time in, as a kernel trace prints it (n >= 64). */
int lr_mcmc_describe(const lr_engine* e, char* buf /* host */, int32_t n);
int lr_mcmc_destroy(lr_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* LITERATE_HIP_H */
