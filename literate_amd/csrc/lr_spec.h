// lr_spec.h - the speculative team engine: persistent kernel for shards with at most one team (a chain pair or a single
// chain) per CU.
//
// The chain step is a serial fp64 stream of ~3 us that needs the lineage scan for ONE bit (accept / reject).  So it is
// taken off the critical path by speculation: while the scanner waves score the pending proposal P, candidate waves
// build the NEXT proposal for both outcomes - Q0 from the accepted state A (P rejected) and Q1 from P (P accepted) -
// with their priors and lookup tables.  The draws are addressed by (seed, chain, iteration, purpose), so both candidates
// consume exactly the draws the sequential loop would: trajectories are those of lr_chain_step_core.  ONE barrier per
// iteration: the scanning wave that finishes last adds the sums in a fixed order, runs the Metropolis-Hastings tests and
// leaves the decision in LDS (lr_spec_deliver); behind the barrier every wave turns the roles of the four sets
// (A, P, Q0, Q1).  Every state in flight owns its lookup-table entries, so a proposal whose move changes no rate and no
// bin edge (the no-op "times" moves, the Gibbs step) copies its base state's entries instead of building them.
//
// Two kinds of team (template MODE of the kernel):
//   a team per PAIR (MODE 0): waves 0..3 are the candidate waves (chain = wave / 2, outcome = wave % 2), each with a column
//     of its own; behind the barrier the scanner waves lay the two pending columns side by side into the six-plane scan
//     table (lr_build_scan_table) - every gather then serves two chains;
//   a team per CHAIN (MODE 1, 2, 3): waves 0, 1 are the candidate waves of the one chain, waves 2, 3 their HELPER waves on
//     the two SIMDs that carry no candidate: a candidate hands its staged segments (RJ) or parameter vector (DDRate /
//     trend_rate) over through LDS as soon as they stand and goes on with guard, prior and set, its helper builds the
//     table meanwhile (and makes the draws of the iteration after the next).  Every state owns a whole scan table
//     ((.x, .y) = (the chain, 0)), so behind the barrier the scanners only switch tables.  MODE 1 is a team of several
//     blocks; MODE 2, 3 are one block on its own CU (the exchange between blocks is compiled out of them); in MODE 2
//     (short scans) the helper stops at the S and E planes and the scanner waves derive the pair planes of the table
//     that becomes pending.
//
// Teams of blocks.  With fewer teams than CUs a team is k blocks (k = 2, 4, 8; one block per CU).  Every block of the
// team runs the same candidate waves on the same state - a replicated state machine, so no table or state ever crosses a
// CU boundary - and scans its own 1/k slice of the lineages.  The only exchange is the two partial sums per block and
// iteration: 8-byte {epoch tag, 32-bit half} granules written with one agent-scope store each and swept by one wave of
// every block until all tags carry the iteration's epoch (cdna_hip_programming.md Guideline 16, form R2; parity-double-
// buffered, zeroed before every launch).  Every block adds the k partial sums in block order, so all of them take
// bit-identical decisions; only block 0 of a team writes trace rows and the final state.  Results never depend on
// dispatch order or placement; a sweep that does not complete within two seconds raises the engine's status word and ends
// the launch (every block polls that word too; later launches of the call end at entry).
#pragma once
#include "lr_scan.h"
#include "lr_step.h"

// one state "as accepted" together with the bookkeeping of the proposal that led to it, in LDS
struct lr_set {
    double L[LR_ROW], M[LR_ROW], tL[LR_ROW], tM[LR_ROW];
    int eL[LR_ROW], eM[LR_ROW];
    int sgL[LR_ROW], sgM[LR_ROW];   // the table builder's bin ranks for these edges (lr_seg_cache), valid iff isc[LR_SETI_SEG]
    double sc[16];
    int isc[8];
};
#define LR_SET_HASTING 0
#define LR_SET_PRIOR 1
#define LR_SET_PRIORPOI 2
#define LR_SET_CONST 3
#define LR_SET_G0 4
#define LR_SET_G1 5
#define LR_SET_POI 6
#define LR_SET_LG0 7
#define LR_SET_LG1 8
#define LR_SET_LPOI 9
#define LR_SET_LOG_U 10
#define LR_SETI_KL 0
#define LR_SETI_KM 1
#define LR_SETI_GIBBS 2
#define LR_SETI_INVALID 3
#define LR_SETI_MOVE 4
#define LR_SETI_SEG 5

// (lr_draw_slot, lr_draws_store / lr_draws_load, lr_spec_draw, lr_spec_draw_part: lr_step.h - the four-chain persistent
// kernel makes its draws ahead the same way)

// the same slot for a parametric sampler's draws (lr_dd_draws)
__device__ __forceinline__ void lr_dd_draws_store(lr_draw_slot* q, const lr_dd_draws& d, int lane) {
    double so = d.log_u;
    so = (lane == 1) ? d.rr_a : so;
    so = (lane == 2) ? d.rr_b : so;
    so = (lane == 3) ? d.slide_u : so;
    so = (lane == 4) ? d.z1 : so;
    if (lane < 8) q->sc[lane] = so;
    q->x[lane] = d.x, q->m[lane] = d.m, q->da[lane] = d.da;
}
__device__ __forceinline__ void lr_dd_draws_load(const lr_draw_slot* q, lr_dd_draws& d, int lane) {
    const double v = q->sc[lane & 7];
    d.log_u = lr_bcast(v, 0), d.rr_a = lr_bcast(v, 1), d.rr_b = lr_bcast(v, 2), d.slide_u = lr_bcast(v, 3), d.z1 = lr_bcast(v, 4);
    d.x = q->x[lane], d.m = q->m[lane], d.da = q->da[lane];
}

struct lr_spec_args {
    unsigned long long* xchg;     // [2 parities][n_teams][LR_TEAM_MAX][LR_SPEC_GRANULES] granules (k > 1)
    unsigned int* status;         // engine status word: 0 ok, 1 = a team exchange timed out
    int team_blocks;              // k
    int n_teams;                  // teams = chains / cpb (rounded up)
    int cpb;                      // chains per team: 2 (a pair: every table gather serves two chains) or 1 (as many teams
                                  // as chains: half the candidate work per CU and an iteration as short as ONE chain's move)
    int cand_share_q16;           // a team per chain: share (x 2^16) of a block's groups that its two candidate waves scan
                                  // once their candidates are built (long scans only: lr_launch_spec)
};

__device__ __forceinline__ void lr_set_load(const lr_set* q, lr_rj_state& s, int lane) {
    s.L = q->L[lane], s.M = q->M[lane], s.tL = q->tL[lane], s.tM = q->tM[lane];
    s.eL = q->eL[lane], s.eM = q->eM[lane];
    s.sgL = q->sgL[lane], s.sgM = q->sgM[lane];
    const double v = q->sc[lane & 15];     // the scalars in two row reads, then register broadcasts
    const int iv = q->isc[lane & 7];
    s.KL = lr_bcast_i(iv, LR_SETI_KL), s.KM = lr_bcast_i(iv, LR_SETI_KM), s.sg_valid = lr_bcast_i(iv, LR_SETI_SEG);
    s.g0 = lr_bcast(v, LR_SET_G0), s.g1 = lr_bcast(v, LR_SET_G1), s.poi = lr_bcast(v, LR_SET_POI);
    s.lg0 = lr_bcast(v, LR_SET_LG0), s.lg1 = lr_bcast(v, LR_SET_LG1), s.lpoi = lr_bcast(v, LR_SET_LPOI);
    s.priorPoi = lr_bcast(v, LR_SET_PRIORPOI);
}

// (p.table_by_helper: the rank cache, its valid flag and the model constant are the helper wave's to write)
__device__ __forceinline__ void lr_set_store(lr_set* q, const lr_rj_state& s, const lr_rj_prop& p, int lane) {
    const bool mine = !p.table_by_helper;
    q->L[lane] = s.L, q->M[lane] = s.M, q->tL[lane] = s.tL, q->tM[lane] = s.tM;
    q->eL[lane] = s.eL, q->eM[lane] = s.eM;
    if (mine) q->sgL[lane] = s.sgL, q->sgM[lane] = s.sgM;
    double so = 0.0;
    so = (lane == LR_SET_HASTING) ? p.hasting : so;
    so = (lane == LR_SET_PRIOR) ? p.prior : so;
    so = (lane == LR_SET_PRIORPOI) ? s.priorPoi : so;
    so = (lane == LR_SET_CONST) ? p.constP : so;
    so = (lane == LR_SET_G0) ? s.g0 : so;
    so = (lane == LR_SET_G1) ? s.g1 : so;
    so = (lane == LR_SET_POI) ? s.poi : so;
    so = (lane == LR_SET_LG0) ? s.lg0 : so;
    so = (lane == LR_SET_LG1) ? s.lg1 : so;
    so = (lane == LR_SET_LPOI) ? s.lpoi : so;
    so = (lane == LR_SET_LOG_U) ? p.log_u : so;
    if (lane < 16 && (mine || lane != LR_SET_CONST)) q->sc[lane] = so;
    int io = 0;
    io = (lane == LR_SETI_KL) ? s.KL : io;
    io = (lane == LR_SETI_KM) ? s.KM : io;
    io = (lane == LR_SETI_GIBBS) ? p.gibbs : io;
    io = (lane == LR_SETI_INVALID) ? p.invalid : io;
    io = (lane == LR_SETI_MOVE) ? p.move : io;
    io = (lane == LR_SETI_SEG) ? s.sg_valid : io;
    if (lane < 8 && (mine || lane != LR_SETI_SEG)) q->isc[lane] = io;
}

// chain state rows in global memory (include/literate_hip.h) -> the accepted set A and the pending set P
__device__ __forceinline__ void lr_sets_from_global(const double* S, const int* I, lr_set* A, lr_set* P, int lane) {
    A->L[lane] = S[LR_ROW_L * LR_ROW + lane], A->M[lane] = S[LR_ROW_M * LR_ROW + lane];
    A->tL[lane] = S[LR_ROW_TL * LR_ROW + lane], A->tM[lane] = S[LR_ROW_TM * LR_ROW + lane];
    P->L[lane] = S[LR_ROW_PL * LR_ROW + lane], P->M[lane] = S[LR_ROW_PM * LR_ROW + lane];
    P->tL[lane] = S[LR_ROW_PTL * LR_ROW + lane], P->tM[lane] = S[LR_ROW_PTM * LR_ROW + lane];
    A->eL[lane] = I[LR_IROW_EL * LR_ROW + lane], A->eM[lane] = I[LR_IROW_EM * LR_ROW + lane];
    P->eL[lane] = I[LR_IROW_PEL * LR_ROW + lane], P->eM[lane] = I[LR_IROW_PEM * LR_ROW + lane];
    A->sgL[lane] = A->sgM[lane] = P->sgL[lane] = P->sgM[lane] = 0;
    const double sc = S[LR_ROW_SCALARS * LR_ROW + lane];
    const int isc = I[LR_IROW_SCALARS * LR_ROW + lane];
    // hyper-parameters: a pending Gibbs step has already put its draws into the scalar row (it is always accepted), so
    // both sets take the row's values
    double a = 0.0, p = 0.0;
    a = (lane == LR_SET_PRIOR) ? lr_bcast(sc, LR_S_PRIORA) : a;
    a = (lane == LR_SET_PRIORPOI) ? lr_bcast(sc, LR_S_PRIORPOIA) : a;
    a = (lane == LR_SET_CONST) ? lr_bcast(sc, LR_S_CONST_A) : a;
    p = (lane == LR_SET_HASTING) ? lr_bcast(sc, LR_S_HASTING) : p;
    p = (lane == LR_SET_PRIOR) ? lr_bcast(sc, LR_S_PRIOR_P) : p;
    p = (lane == LR_SET_PRIORPOI) ? lr_bcast(sc, LR_S_PRIORPOI_P) : p;
    p = (lane == LR_SET_CONST) ? lr_bcast(sc, LR_S_CONST_P) : p;
    p = (lane == LR_SET_LOG_U) ? lr_bcast(sc, LR_S_LOG_U) : p;
    const double g0 = lr_bcast(sc, LR_S_GRATE_L), g1 = lr_bcast(sc, LR_S_GRATE_M), poi = lr_bcast(sc, LR_S_POI);
    const double lg0 = lr_bcast(sc, LR_S_LOG_G0), lg1 = lr_bcast(sc, LR_S_LOG_G1), lpoi = lr_bcast(sc, LR_S_LOG_POI);
    double h = 0.0;
    bool hy = true;
    switch (lane) {
        case LR_SET_G0: h = g0; break;
        case LR_SET_G1: h = g1; break;
        case LR_SET_POI: h = poi; break;
        case LR_SET_LG0: h = lg0; break;
        case LR_SET_LG1: h = lg1; break;
        case LR_SET_LPOI: h = lpoi; break;
        default: hy = false;
    }
    if (hy) a = h, p = h;
    if (lane < 16) A->sc[lane] = a, P->sc[lane] = p;
    int ia = 0, ip = 0;
    ia = (lane == LR_SETI_KL) ? lr_bcast_i(isc, LR_I_KL) : ia;
    ia = (lane == LR_SETI_KM) ? lr_bcast_i(isc, LR_I_KM) : ia;
    ip = (lane == LR_SETI_KL) ? lr_bcast_i(isc, LR_I_PKL) : ip;
    ip = (lane == LR_SETI_KM) ? lr_bcast_i(isc, LR_I_PKM) : ip;
    ip = (lane == LR_SETI_GIBBS) ? lr_bcast_i(isc, LR_I_GIBBS) : ip;
    ip = (lane == LR_SETI_INVALID) ? lr_bcast_i(isc, LR_I_INVALID) : ip;
    ip = (lane == LR_SETI_MOVE) ? lr_bcast_i(isc, LR_I_MOVE) : ip;
    if (lane < 8) A->isc[lane] = ia, P->isc[lane] = ip;
}

// bookkeeping of one chain that only its clerk wave needs (wave-uniform)
struct lr_spec_book {
    double lik_p;
    unsigned long long next_sample;
    int n_acc, trace_slot;
};

__device__ __attribute__((noinline)) void lr_sets_to_global(double* S, int* I, const lr_set* A, const lr_set* P, double likA,
                                                  unsigned long long it, const lr_spec_book& bk, bool rj, int lane) {
    S[LR_ROW_L * LR_ROW + lane] = A->L[lane], S[LR_ROW_M * LR_ROW + lane] = A->M[lane];
    S[LR_ROW_TL * LR_ROW + lane] = A->tL[lane], S[LR_ROW_TM * LR_ROW + lane] = A->tM[lane];
    S[LR_ROW_PL * LR_ROW + lane] = P->L[lane], S[LR_ROW_PM * LR_ROW + lane] = P->M[lane];
    S[LR_ROW_PTL * LR_ROW + lane] = P->tL[lane], S[LR_ROW_PTM * LR_ROW + lane] = P->tM[lane];
    I[LR_IROW_EL * LR_ROW + lane] = A->eL[lane], I[LR_IROW_EM * LR_ROW + lane] = A->eM[lane];
    I[LR_IROW_PEL * LR_ROW + lane] = P->eL[lane], I[LR_IROW_PEM * LR_ROW + lane] = P->eM[lane];
    // the scalar row carries the hyper-parameters of the pending proposal (= the accepted ones unless it is a Gibbs step)
    double so = 0.0;
    so = (lane == LR_S_LIKA) ? likA : so;
    so = (lane == LR_S_PRIORA) ? A->sc[LR_SET_PRIOR] : so;
    so = (lane == LR_S_PRIORPOIA) ? A->sc[LR_SET_PRIORPOI] : so;
    so = (lane == LR_S_GRATE_L) ? P->sc[LR_SET_G0] : so;
    so = (lane == LR_S_GRATE_M) ? P->sc[LR_SET_G1] : so;
    so = (lane == LR_S_POI) ? P->sc[LR_SET_POI] : so;
    so = (lane == LR_S_HASTING) ? P->sc[LR_SET_HASTING] : so;
    so = (lane == LR_S_PRIOR_P) ? P->sc[LR_SET_PRIOR] : so;
    so = (lane == LR_S_PRIORPOI_P) ? P->sc[LR_SET_PRIORPOI] : so;
    so = (lane == LR_S_CONST_P) ? P->sc[LR_SET_CONST] : so;
    so = (lane == LR_S_CONST_A) ? A->sc[LR_SET_CONST] : so;
    so = (lane == LR_S_LIK_P) ? bk.lik_p : so;
    so = (lane == LR_S_LOG_G0) ? P->sc[LR_SET_LG0] : so;
    so = (lane == LR_S_LOG_G1) ? P->sc[LR_SET_LG1] : so;
    so = (lane == LR_S_LOG_POI) ? P->sc[LR_SET_LPOI] : so;
    so = (lane == LR_S_LOG_U) ? P->sc[LR_SET_LOG_U] : so;
    if (!rj && lane != LR_S_LIKA && lane != LR_S_PRIORA && lane != LR_S_HASTING && lane != LR_S_PRIOR_P &&
        lane != LR_S_LIK_P && lane != LR_S_LOG_U)
        so = 0.0;                                    // the parametric samplers keep the other slots zero
    S[LR_ROW_SCALARS * LR_ROW + lane] = so;
    int io = 0;
    io = (lane == LR_I_KL) ? A->isc[LR_SETI_KL] : io;
    io = (lane == LR_I_KM) ? A->isc[LR_SETI_KM] : io;
    io = (lane == LR_I_PKL) ? P->isc[LR_SETI_KL] : io;
    io = (lane == LR_I_PKM) ? P->isc[LR_SETI_KM] : io;
    io = (lane == LR_I_GIBBS) ? P->isc[LR_SETI_GIBBS] : io;
    io = (lane == LR_I_INVALID) ? P->isc[LR_SETI_INVALID] : io;
    io = (lane == LR_I_IT_LO) ? (int)(unsigned)it : io;
    io = (lane == LR_I_IT_HI) ? (int)(unsigned)(it >> 32) : io;
    io = (lane == LR_I_ACCEPTED) ? bk.n_acc : io;
    io = (lane == LR_I_MOVE) ? P->isc[LR_SETI_MOVE] : io;
    io = (lane == LR_I_NEXT_LO) ? (int)(unsigned)bk.next_sample : io;
    io = (lane == LR_I_NEXT_HI) ? (int)(unsigned)(bk.next_sample >> 32) : io;
    io = (lane == LR_I_SLOT) ? bk.trace_slot : io;
    I[LR_IROW_SCALARS * LR_ROW + lane] = io;
}

#ifdef LR_DIAG
#define LR_XDECL() unsigned long long dg_work = 0, dg_wait1 = 0, dg_p2 = 0, dg_wait2 = 0, dg_t = 0, dg_a = 0, dg_b = 0, dg_c = 0
#define LR_XBEGIN() dg_t = wall_clock64()
#define LR_XSTAMP(acc) { const unsigned long long t_ = wall_clock64(); acc += t_ - dg_t; dg_t = t_; }
#define LR_XDUMP() if (lane == 0 && blockIdx.x < 64) { unsigned long long* o = lr_diag_step + 16384 + (blockIdx.x * 16 + wave) * 4; o[0] = dg_work, o[1] = dg_wait1, o[2] = dg_p2, o[3] = dg_wait2; unsigned long long* q = lr_diag_step + 24576 + (blockIdx.x * 16 + wave) * 4; q[0] = dg_a, q[1] = dg_b, q[2] = dg_c; }
#else
#define LR_XDECL()
#define LR_XBEGIN()
#define LR_XSTAMP(acc)
#define LR_XDUMP()
#endif
// trace row of the clerk, out of line: it runs once per s_freq iterations and must not cost the candidate loop registers
__device__ __attribute__((noinline)) void lr_spec_trace(const lr_step_args* a, int c, int lane, int slot, unsigned long long it,
                                                        double likA, const lr_set* A, int rj) {
    if (rj) {
        lr_rj_state s;
        lr_set_load(A, s, lane);
        lr_write_trace_row(*a, c, lane, slot, it, likA, A->sc[LR_SET_PRIOR], s);
    } else {
        lr_dd_write_trace_row(*a, c, lane, slot, it, likA, A->sc[LR_SET_PRIOR], A->L[lane]);
    }
}

typedef __attribute__((address_space(1))) unsigned long long lr_gu64;
typedef __attribute__((address_space(1))) unsigned int lr_gu32;

// The decisions of one iteration, left in LDS by the deciding wave before the iteration's barrier
struct lr_spec_decision {
    double lik[2];               // log-likelihood of the two pending proposals (= the accepted states' if accepted)
    double lik_p[2];             // what the state row records for them (-inf for an invalid proposal)
    int sel;                     // d0 * 2 + d1
    int pad_;
};

// the roles of a chain's four sets after a decision: accepted -> (A, P, Q0, Q1) = (P, Q1, A, Q0); rejected -> (A, Q0, P, Q1)
__device__ __forceinline__ int lr_spec_turn(int role, int acc) {
    const int oA = role & 3, oP = (role >> 2) & 3, o0 = (role >> 4) & 3, o1 = (role >> 6) & 3;
    return acc ? (oP | (o1 << 2) | (oA << 4) | (o0 << 6)) : (oA | (o0 << 2) | (oP << 4) | (o1 << 6));
}
#define LR_SPEC_ROLES0 (0 | (1 << 2) | (2 << 4) | (3 << 6))

// everything the block keeps in LDS
template <int H, int NW, int ENT>
struct lr_spec_lds {
    static constexpr bool GENERAL_ENTRIES = ENT == 2;
    static constexpr int TAB = 2 * H * ENT;          // double2 per candidate pair table (the global-memory layout)
    static constexpr int COL = 2 * H * ENT;          // doubles per column: S' entries [0,H), E' [H,2H) (general times: + slopes)
    union {
        struct {
            // A team per PAIR.  One column of lookup-table entries per state in flight, entries 1 double apart:
            // cols[chain][set] belongs to sets[chain][set] - the accepted state, the pending proposal and the two
            // candidates being built; and what the scanner waves gather from: the six planes (lr_scan.h) of the two
            // pending columns side by side, which they lay out behind the barrier (lr_build_scan_table)
            double cols[2][4][2 * H * ENT];
            double2 scan[LR_UNIT_PLANES * H];
        } pair;
        // A team per CHAIN.  One whole scan table per state in flight - the six planes, (.x, .y) = (the chain, 0) - built
        // by the candidate's helper wave with its pair planes: behind the barrier the scanner waves only switch to the
        // table of the proposal that is now pending
        double2 tabs[4][LR_UNIT_PLANES * H];
    } t;
    int final_p[2];              // set that holds each chain's pending proposal when the kernel ends
    lr_table_hand hand[2];       // a team per chain: candidate wave k -> its helper wave 2 + k
    double red[NW][2];           // per scanner wave: partial sums of the two chains
    lr_spec_decision dec[2];     // [iteration parity]: what the deciding wave found
    double likA[2];              // log-likelihood of the accepted state of the two chains
    int abort_flag;
    int scan_arrive;             // scanner waves that have delivered their sums, counted over the whole launch
    int plane_arrive;            // scanner waves that have written their share of the pair-sum planes, likewise
    lr_seg_scratch scratch[4];
    lr_set sets[2][4];
    lr_draw_slot draws[2][2];    // [chain][iteration parity]
    double br[256], logbr[256];  // per-bin data constants of the table builders (br_length / DT / TREND and log br_length)
    lr_step_args args;           // copy for the out-of-line trace writer: taking the address of the kernel argument itself
                                 // would move the whole struct from scalar registers to scratch
};

struct lr_spec_ctx {
    lr_packed_lineages pk;       // the packed lineages
    long long g0;                // this block's slice of them: first 16-byte group ...
    long long n8;                // ... and number of groups
    lr_spec_args x;
    long long n_iters;
    unsigned long long it0;      // iteration of the proposal pending at entry
    int c0, C, team, rank;
    int n_act;                   // chains of this block: 1 or 2
    long long cand_g0, cand_n;   // a team per chain: the candidate waves' share of the block's slice (behind the scanners')
};

// What a decision needs beside the scan sums: lanes 0-15 the pending proposal's scalars, 16-31 the accepted state's, of
// both chains; they stand since the last barrier, so a scanning wave fetches them BEFORE its scan (four row reads that
// are free while the scan runs; used by the deciding wave only, but which wave that will be is not known yet).
struct lr_spec_dec_in {
    double v0, v1, lA;
    int i0, i1;
};
template <class LDS>
__device__ __forceinline__ lr_spec_dec_in lr_spec_dec_fetch(const LDS& sm, const lr_spec_ctx& ctx, int role0, int role1, int lane) {
    const bool act1 = ctx.n_act > 1;
    const int cc1 = act1 ? 1 : 0;
    const int r1 = act1 ? role1 : role0;
    const lr_set* P0 = &sm.sets[0][(role0 >> 2) & 3];
    const lr_set* A0 = &sm.sets[0][role0 & 3];
    const lr_set* P1 = &sm.sets[cc1][(r1 >> 2) & 3];
    const lr_set* A1 = &sm.sets[cc1][r1 & 3];
    lr_spec_dec_in in;
    in.v0 = (lane & 16) ? A0->sc[lane & 15] : P0->sc[lane & 15];
    in.v1 = (lane & 16) ? A1->sc[lane & 15] : P1->sc[lane & 15];
    in.i0 = P0->isc[lane & 7], in.i1 = P1->isc[lane & 7];
    in.lA = sm.likA[lane & 1];
    return in;
}

// Blocks per team as the kernel's mode knows it: MODE 2, 3 (a team per chain on its own CU) have none to exchange with,
// so the exchange, its abort path and the priority games of a team are compiled out of them.
template <int MODE>
__device__ __forceinline__ int lr_spec_team_blocks(const lr_spec_args& x) {
    return (MODE == 2 || MODE == 3) ? 1 : x.team_blocks;
}

// End of a wave's scan share: its sums to LDS, and - for the wave that arrives LAST among the block's scanning waves (the
// scanner waves; a team per chain: + the two candidate waves) - the decision of the iteration: the block's sums added in
// a fixed order (in a team: published and the team's swept), the Metropolis-Hastings tests, the outcome left in LDS.
template <int H, int T, bool RJ, bool GENERAL, int MODE>
__device__ __forceinline__ void lr_spec_deliver(lr_spec_lds<H, T / LR_WAVE, GENERAL ? 2 : 1>& sm, const lr_spec_ctx& ctx, long long iter,
                                                const lr_spec_dec_in& in, double acc0, double acc1, int wave, int lane) {
    constexpr int NW = T / LR_WAVE;
    constexpr bool rj = RJ;
    constexpr bool SINGLE = MODE != 0;
    // (a team per chain under a parametric sampler: the two candidate waves scan - and count themselves in - when they
    // were given a share; the RJ sampler's never are: lr_launch_spec)
    const int n_arrive = (NW - 4) + ((SINGLE && !RJ && ctx.cand_n > 0) ? 2 : 0);
    const int k_team = lr_spec_team_blocks<MODE>(ctx.x);
    const unsigned long long it = ctx.it0 + (unsigned long long)iter;
    {
        const double s0 = lr_wave_sum(acc0), s1 = lr_wave_sum(acc1);
        if (lane == 0) sm.red[wave][0] = s0, sm.red[wave][1] = s1;
        // the scanner waves count themselves in on an LDS word (a wave's LDS operations execute in order: sums first,
        // then the count); the wave that arrives last decides
        int prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(&sm.scan_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (prev == n_arrive * ((int)iter + 1) - 1) {
            const double v0 = in.v0, v1 = in.v1, lA = in.lA;
            const int i0 = in.i0, i1 = in.i1;
            asm volatile("" ::: "memory");     // the sums are read after the count was seen
            // the block's sums: one read per lane, then a fixed pairwise tree over the scanner waves' values
            const double2 rv = *reinterpret_cast<const double2*>(&sm.red[lane < NW ? lane : NW - 1][0]);
            double t0[NW - 4], t1[NW - 4];
#pragma unroll
            for (int w = 4; w < NW; ++w) t0[w - 4] = lr_bcast(rv.x, w), t1[w - 4] = lr_bcast(rv.y, w);
#pragma unroll
            for (int n = NW - 4; n > 1; n = (n + 1) / 2) {
#pragma unroll
                for (int j = 0; j < n / 2; ++j) t0[j] = t0[2 * j] + t0[2 * j + 1], t1[j] = t1[2 * j] + t1[2 * j + 1];
                if (n & 1) t0[n / 2] = t0[n - 1], t1[n / 2] = t1[n - 1];
            }
            double sum0 = t0[0], sum1 = t1[0];
            if (SINGLE && !RJ && ctx.cand_n > 0) {
                // (a team per chain on a long scan: the two candidate waves' shares, added behind the scanner waves' tree)
                sum0 += lr_bcast(rv.x, 0), sum1 += lr_bcast(rv.y, 0);
                sum0 += lr_bcast(rv.x, 1), sum1 += lr_bcast(rv.y, 1);
            }
            bool fail = false;
            if (k_team > 1) {
                // this block's sums published as four {epoch, half} granules; then the sweep over the team's
                const unsigned int epoch = (unsigned int)iter + 1u;
                lr_gu64* slot = (lr_gu64*)(ctx.x.xchg + ((size_t)(epoch & 1u) * ctx.x.n_teams + ctx.team) * (LR_TEAM_MAX * LR_SPEC_GRANULES));
                if (lane < 4) {
                    const double v = (lane < 2) ? sum0 : sum1;
                    const unsigned int half = (lane & 1) ? (unsigned int)__double2hiint(v) : (unsigned int)__double2loint(v);
                    __hip_atomic_store(slot + ctx.rank * LR_SPEC_GRANULES + lane, ((unsigned long long)epoch << 32) | half,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const bool mine = lane < 4 * k_team;
                lr_gu64* g = slot + (lane >> 2) * LR_SPEC_GRANULES + (lane & 3);
                unsigned long long v = (unsigned long long)epoch << 32;
                const unsigned long long t_start = wall_clock64();
                for (unsigned int spins = 0;; ++spins) {
                    if (mine) v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((unsigned int)(v >> 32) == epoch)) break;
                    if ((spins & 255u) == 255u) {
                        // give up when the engine's status word is raised or after two seconds (uniform over the wave)
                        const unsigned int st = __hip_atomic_load((lr_gu32*)ctx.x.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st != 0u || wall_clock64() - t_start > LR_SPEC_TIMEOUT_TICKS) {
                            fail = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (fail) {
                    if (lane == 0) {
                        __hip_atomic_store((lr_gu32*)ctx.x.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sm.abort_flag = 1;
                    }
                } else {
                    // [block][chain] halves sit in lanes 4 b + 2 c (+1): summed in block order through register broadcasts
                    const int half = (int)(unsigned int)v;
                    sum0 = 0.0, sum1 = 0.0;
#pragma unroll
                    for (int b = 0; b < LR_TEAM_MAX; ++b)
                        if (b < k_team) {
                            sum0 += __hiloint2double(lr_bcast_i(half, 4 * b + 1), lr_bcast_i(half, 4 * b));
                            sum1 += __hiloint2double(lr_bcast_i(half, 4 * b + 3), lr_bcast_i(half, 4 * b + 2));
                        }
                }
                
            }
            if (!fail) {
                // the two Metropolis-Hastings tests (LRF:305-319; DD:204-219)
                int d[2] = {0, 0};
                double lik[2] = {0.0, 0.0}, lik_p[2] = {0.0, 0.0};
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    if (cc >= ctx.n_act) continue;
                    const double v = cc ? v1 : v0;
                    const int iv = cc ? i1 : i0;
                    const int gibbs = lr_bcast_i(iv, LR_SETI_GIBBS), invalid = lr_bcast_i(iv, LR_SETI_INVALID);
                    const double priorP = lr_bcast(v, LR_SET_PRIOR), priorA = lr_bcast(v, 16 + LR_SET_PRIOR);
                    const double hasting = lr_bcast(v, LR_SET_HASTING), log_u = lr_bcast(v, LR_SET_LOG_U);
                    const double likA = lr_bcast(lA, cc);
                    const double lik_sum = cc ? sum1 : sum0;
                    bool ok;
                    if (rj) {
                        ok = lr_mh_accept(gibbs, invalid, lik_sum, lr_bcast(v, LR_SET_CONST), likA, priorP, priorA, hasting, log_u, &lik[cc]);
                        lik_p[cc] = invalid ? -INFINITY : lik[cc];
                        lr_warn_kcap((unsigned int*)ctx.x.status + 1, invalid, lane);   // (the warning word sits behind the status word) P is the chain's own pending proposal, decided here
                    } else {
                        lik[cc] = lik_sum;
                        ok = lr_dd_accept(lik[cc], likA, priorP, priorA, hasting, log_u, it);
                        lik_p[cc] = lik[cc];
                    }
                    d[cc] = ok ? 1 : 0;
                }
                lr_spec_decision* out = &sm.dec[iter & 1];
                if (lane < 2) {
                    const double l = lane ? lik[1] : lik[0];
                    out->lik[lane] = l, out->lik_p[lane] = lane ? lik_p[1] : lik_p[0];
                    if (lane ? d[1] : d[0]) sm.likA[lane] = l;
                }
                if (lane == 2) out->sel = d[0] * 2 + d[1];
            }
        }
    }
}

// The scanner role (waves 4..NW-1): per iteration one pass over the block's slice of the lineages against the pending
// pair table and partial sums to LDS.  The wave that finishes LAST also takes the decision of the iteration: it adds up
// the block's sums (in a team: publishes them and sweeps the team's), runs the two Metropolis-Hastings tests and leaves
// the outcome in LDS - all of it BEFORE the iteration's one barrier, beside the candidate waves' tail, because a
// decision needs the scan sums and the scalars of the pending proposal and the accepted state, not the candidates being
// built.  Behind the barrier every wave reads the outcome, turns the roles of the sets and goes on: the scanners to the
// pair table of the selected candidates, the candidate waves to the next candidates.  The last two scanner waves also
// have the draw duty.
// MODE: 0 = a team per pair; a team per chain: 1 = in a team of several blocks, 2 = on its own CU, short scan (pair planes
// by the scanner waves), 3 = on its own CU, long scan (pair planes by the helper waves, as in mode 1) - a template
// parameter rather than a run-time switch: every path that is compiled in costs instruction-cache room and registers in a
// kernel of ~9000 instructions (taking a dead path out of the RJ kernels was worth 4 % on cfg2)
template <int H, int T, bool RJ, bool GENERAL, int MODE>
__device__ __forceinline__ void lr_spec_scan_role(lr_spec_lds<H, T / LR_WAVE, GENERAL ? 2 : 1>& sm, const lr_step_args& a,
                                                  const lr_spec_ctx& ctx, int tid) {
    constexpr int NW = T / LR_WAVE;
    constexpr int NSCAN = (NW - 4) * LR_WAVE;
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int k_team = lr_spec_team_blocks<MODE>(ctx.x);
    constexpr bool rj = RJ;
    const bool act1 = ctx.n_act > 1;
    // Draw duty.  A block on its own is bound by its slowest wave before the barrier, so the duty is split over the last
    // four scanner waves, two per chain; in a team the last scanner to finish also runs the exchange and must not carry
    // more than its scan, so there the last two scanner waves (the smallest scan shares) take a chain each.
    // A team per chain: the draw duty lies with the helper waves 2, 3 (lr_spec_help_role), and there is no scan table to
    // build - every state in flight has its own.
    constexpr bool SINGLE = MODE != 0, PLANES_BY_SCANNERS = MODE == 2;
    constexpr bool single = SINGLE;
    const bool split_draws = rj && k_team == 1;
    const int dch = split_draws ? (wave - (NW - 4)) >> 1 : wave - (NW - 2), dpart = (wave - (NW - 4)) & 1;
    const bool drawer = !single && dch >= 0 && dch < ctx.n_act;
    // A block on its own is bound by whichever wave reaches the barrier last, and that used to be a drawer (scan table,
    // scan, then the draws): there the drawers make their draws FIRST - they depend on nothing - while the other scanner
    // waves build the scan table, and only then scan.  In a team the last scanner also runs the exchange, so the
    // drawers scan first and draw behind the arrival.
    const bool draws_first = k_team == 1;
    const int n_build = draws_first ? (NW - 4) - (rj ? 4 : 2) : NW - 4;      // scanner waves 4 .. 4 + n_build - 1 build the scan table
    const bool builder = wave - 4 < n_build;
    const int build_id = tid - 4 * LR_WAVE;
    (void)build_id, (void)builder;
    auto draw_duty = [&](unsigned long long it_draw, unsigned long long slot) {
        if (!rj) {
            lr_dd_draws dd;
            lr_make_dd_draws(a, ctx.c0 + dch, lane, it_draw, dd);
            lr_dd_draws_store(&sm.draws[dch][slot], dd, lane);
        } else if (split_draws) {
            lr_spec_draw_part(a, ctx.c0 + dch, lane, it_draw, &sm.draws[dch][slot], dpart);
        } else {
            lr_spec_draw(a, ctx.c0 + dch, lane, it_draw, &sm.draws[dch][slot]);
        }
    };
    // every scanner lane strides over the block's slice (unequal shares between the SIMDs with and without a candidate
    // wave measured slower at every split: a scanning wave is bound by its own trip latency, not by its SIMD's issue rate)
    const long long part_g0 = ctx.g0, part_n = ctx.n8;
    const int sid = tid - 4 * LR_WAVE, n_scan = NSCAN;
    int sel = 0;
    int role0 = LR_SPEC_ROLES0, role1 = LR_SPEC_ROLES0;
    // In a team the scanners and the exchange behind them are the critical path: they may be given priority over the
    // candidate wave of their SIMD (which is older and would otherwise win every arbitration).
    if (LR_SPEC_SCAN_PRIO > 0 && k_team > 1) {
        if (wave >= (NW + 4) / 2) __builtin_amdgcn_s_setprio(LR_SPEC_SCAN_PRIO + 1);
        else __builtin_amdgcn_s_setprio(LR_SPEC_SCAN_PRIO);
    }
    // every scan starts with the same group: kept in registers, no load to wait for at the top of an iteration
    lr_first_group first;
    lr_load_first_group<GENERAL>(ctx.pk, part_g0, part_n, sid, &first);
    LR_XDECL();
    for (long long iter = 0; iter < ctx.n_iters; ++iter) {
        const unsigned long long it = ctx.it0 + (unsigned long long)iter;
        LR_XBEGIN();
        const lr_spec_dec_in dec_in = lr_spec_dec_fetch(sm, ctx, role0, role1, lane);
        double acc0 = 0.0, acc1 = 0.0;
        const char* lbase = reinterpret_cast<const char*>(single ? sm.t.tabs[(role0 >> 2) & 3] : sm.t.pair.scan);
        // (a team per chain: the one-chain form of the scan - 8-byte gathers, half the fp64 instructions)
        lr_persist_scan<H, GENERAL, GENERAL ? 1 : LR_SPEC_SCAN_UNROLL, true, false, false, SINGLE>(lbase, ctx.pk, part_g0, part_n, sid, n_scan, &acc0, &acc1, &first);
        LR_XSTAMP(dg_a);
        lr_spec_deliver<H, T, RJ, GENERAL, MODE>(sm, ctx, iter, dec_in, acc0, acc1, wave, lane);
        // draw duty after the sums are delivered: the (state independent) draws of iteration it + 2 for chain dch
        // (a block on its own has made these draws at the top of the iteration, see below)
        if (drawer && !draws_first) draw_duty(it + 2, it & 1);
        LR_XSTAMP(dg_work);
        __syncthreads();                                                     // the decision and the candidates are in
        LR_XSTAMP(dg_wait1);
        if (k_team > 1 && sm.abort_flag) return;
        sel = sm.dec[iter & 1].sel;
        role0 = lr_spec_turn(role0, sel >> 1), role1 = lr_spec_turn(role1, sel & 1);
        if (PLANES_BY_SCANNERS) {
            // a team per chain on a short scan: the pair planes of the table that is pending now, by all scanner lanes; the
            // scanner waves then wait for each other on an LDS counter
            double2* tab = sm.t.tabs[(role0 >> 2) & 3];
            if (GENERAL) lr_pair_planes_block_general(tab, H, a.cfg.n_bins, tid - 4 * LR_WAVE, NSCAN);
            else lr_pair_planes_block(tab, H, a.cfg.n_bins, tid - 4 * LR_WAVE, NSCAN);
            LR_WAVE_LDS_ORDER();
            if (lane == 0) __hip_atomic_fetch_add(&sm.plane_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int want = (NW - 4) * ((int)iter + 1);
            while (__hip_atomic_load(&sm.plane_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        if (!single) {
            // The scan table of the next iteration from the columns of the proposals now pending (every scanner wave is past
            // its scan of the old one: the barrier), both chains at once, by all scanner lanes; the scanner waves then wait
            // for each other on an LDS counter (the candidate waves are already building: no block barrier).
            // (A team per chain: nothing to do - the next scan gathers from the pending proposal's own table.)
            if (builder) {
                lr_build_scan_table<GENERAL>(sm.t.pair.scan, sm.t.pair.cols[0][(role0 >> 2) & 3], act1 ? sm.t.pair.cols[1][(role1 >> 2) & 3] : nullptr, H,
                                             a.cfg.n_bins, build_id, n_build * LR_WAVE);
                LR_WAVE_LDS_ORDER();
                if (lane == 0) __hip_atomic_fetch_add(&sm.plane_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (drawer && iter + 1 < ctx.n_iters) {
                // the draws of iteration (it + 1) + 2, i.e. what the drawers of a team make behind their next scan
                draw_duty(it + 3, (it + 1) & 1);
            }
            const int want = n_build * ((int)iter + 1);
            while (__hip_atomic_load(&sm.plane_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        LR_XSTAMP(dg_p2);
    }
    LR_XDUMP();
}

// The helper role (a team per chain: waves 2, 3, on the SIMDs that carry no candidate wave).  Per iteration
//  (i) the state-independent draws of iteration it + 2 - wave 2 the wave-uniform ones, wave 3 the per-rate multiplier draws
//      (lr_spec_draw_part; a parametric sampler's come from wave 2 alone) - into the slot the candidates of the NEXT
//      iteration read;
//  (ii) the lookup tables of candidate wave (wave - 2)'s proposal.  That wave hands its segments (RJ sampler) or its
//      parameter vector (parametric samplers) over as soon as they stand and goes on with the guard, the prior and the
//      set; this wave builds the table with its pair planes meanwhile and - RJ - writes the model constant and the rank
//      cache into the proposal's set: a candidate is two waves on two SIMDs for the longer half of its build.
// Same barrier per iteration as the other roles.
template <int H, int T, bool RJ, bool GENERAL, int MODE>
__device__ __forceinline__ void lr_spec_help_role(lr_spec_lds<H, T / LR_WAVE, GENERAL ? 2 : 1>& sm, const lr_step_args& a,
                                                  const lr_spec_ctx& ctx, int tid) {
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int k_team = lr_spec_team_blocks<MODE>(ctx.x);
    const int k = wave - 2;
    for (long long iter = 0; iter < ctx.n_iters; ++iter) {
        const unsigned long long it = ctx.it0 + (unsigned long long)iter;
        if (RJ) {
            lr_spec_draw_part(a, ctx.c0, lane, it + 2, &sm.draws[0][it & 1], k);
            lr_table_hand* hand = &sm.hand[k];
            while (__hip_atomic_load(&hand->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (int)iter + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            if (!hand->noop) {
                lr_set* out = &sm.sets[0][hand->out_idx];
                const lr_set* base = &sm.sets[0][hand->base_idx];
                const lr_seg_scratch* sc = &sm.scratch[k];
                lr_seg_cache sg{base->sgL[lane], base->sgM[lane], hand->reuse != 0};
                const int eL = lane <= LR_KMAX ? sc->edge[0][lane] : 0, eM = lane <= LR_KMAX ? sc->edge[1][lane] : 0;
                double* tabd = reinterpret_cast<double*>(sm.t.tabs[hand->out_idx]);
                const double constP = lr_build_tables_segments<lr_bins_per_lane(H), 2>(
                    sc, eL, eM, hand->KL, hand->KM, sm.br, sm.logbr, a.cfg.model, a.cfg.n_bins, a.n_cls, a.H,
                    reinterpret_cast<double2*>(tabd), lane, GENERAL ? LR_TAB_PAIRGEN : LR_TAB_UNIT, a.cfg.frac_birth,
                    a.cfg.frac_death, GENERAL ? 6 * H : 2, &sg);
                if (MODE != 2) {
                    LR_WAVE_LDS_ORDER();
                    if (GENERAL) lr_pair_planes_wave_general(tabd, H, a.cfg.n_bins, lane, 0);
                    else lr_pair_planes_wave(tabd, H, a.cfg.n_bins, lane, 0);
                }
                out->sgL[lane] = sg.packL, out->sgM[lane] = sg.packM;
                if (lane == 0) out->sc[LR_SET_CONST] = constP, out->isc[LR_SETI_SEG] = sg.packL != -1 ? 1 : 0;
            }
        } else {
            if (wave == 2) {
                lr_dd_draws dd;
                lr_make_dd_draws(a, ctx.c0, lane, it + 2, dd);
                lr_dd_draws_store(&sm.draws[0][it & 1], dd, lane);
            }
            // the lookup tables (and pair planes) of candidate wave k's proposal, from its parameter vector
            lr_table_hand* hand = &sm.hand[k];
            while (__hip_atomic_load(&hand->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (int)iter + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const double P = lane < 8 ? hand->par[lane] : 0.0;
            lr_param_tables<true, MODE != 2, 2>(a, P, sm.br, sm.t.tabs[hand->out_idx], GENERAL ? LR_TAB_PAIRGEN : LR_TAB_UNIT, GENERAL ? 6 * H : 2, lane);
        }
        __syncthreads();
        if (k_team > 1 && sm.abort_flag) return;
    }
}

// The candidate role (waves 0..3; chain = wave / 2, outcome = wave % 2): build the candidate of iteration it + 1 while
// the others scan, then - all four waves alike, each on its own SIMD - decide both chains, copy the selected pair
// table, turn the roles of the sets; waves 0 and 2 keep the books (acceptance count, trace rows, final state).
template <int H, int T, bool RJ, bool GENERAL, int MODE>
__device__ __forceinline__ void lr_spec_cand_role(lr_spec_lds<H, T / LR_WAVE, GENERAL ? 2 : 1>& sm, const lr_step_args& a,
                                                  const lr_spec_ctx& ctx, int tid) {
    constexpr int NW = T / LR_WAVE;
    constexpr bool SINGLE = MODE != 0;
    // the builders' `so`: doubles from a value to its slope - inside a column (a team per pair) / in the six-plane image
    constexpr int ES = GENERAL ? (SINGLE ? 6 * H : 2 * H) : 2;
    constexpr int COL = lr_spec_lds<H, T / LR_WAVE, GENERAL ? 2 : 1>::COL;
    // the per-bin data constants come from their LDS copies (a global load per candidate would sit on the critical path)
    const double* br_lds = sm.br;
    const double* logbr_lds = sm.logbr;
    const int lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int c = wave >> 1, k = wave & 1;
    const int c0 = ctx.c0, C = ctx.C;
    const int k_team = lr_spec_team_blocks<MODE>(ctx.x);
    constexpr bool rj = RJ;
    const bool act1 = ctx.n_act > 1;
    const bool mine_active = c < ctx.n_act;
    // run state: which of a chain's four sets plays which role (bits 0-1 A, 2-3 P, 4-5 Q0, 6-7 Q1), the accepted
    // log-likelihoods; every candidate wave tracks both chains (all four take the same decisions)
    int role0 = LR_SPEC_ROLES0, role1 = LR_SPEC_ROLES0;
    double likA0 = 0.0, likA1 = 0.0;
    likA0 = a.state_f64[((size_t)c0 * LR_STATE_ROWS + LR_ROW_SCALARS) * LR_ROW + LR_S_LIKA];
    if (act1) likA1 = a.state_f64[((size_t)(c0 + 1) * LR_STATE_ROWS + LR_ROW_SCALARS) * LR_ROW + LR_S_LIKA];
    lr_spec_book bk = {0.0, 0ull, 0, 0};
    {
        const int cm = min(c0 + c, C - 1);
        const int* Ic = a.state_i32 + ((size_t)cm * LR_ISTATE_ROWS + LR_IROW_SCALARS) * LR_ROW;
        bk.n_acc = Ic[LR_I_ACCEPTED], bk.trace_slot = Ic[LR_I_SLOT];
        bk.next_sample = (unsigned long long)(unsigned)Ic[LR_I_NEXT_LO] | ((unsigned long long)(unsigned)Ic[LR_I_NEXT_HI] << 32);
        bk.lik_p = a.state_f64[((size_t)cm * LR_STATE_ROWS + LR_ROW_SCALARS) * LR_ROW + LR_S_LIK_P];
    }
    LR_XDECL();
    for (long long iter = 0; iter < ctx.n_iters; ++iter) {
        const unsigned long long it = ctx.it0 + (unsigned long long)iter;      // iteration of the pending proposal
        LR_XBEGIN();
        // ---- phase 1: the candidate of iteration it + 1 for outcome k of chain c ----
        if (mine_active) {
            const int role = c ? role1 : role0;
            const int base_i = (role >> (k ? 2 : 0)) & 3, out_i = (role >> (k ? 6 : 4)) & 3;
            const lr_set* base = &sm.sets[c][base_i];
            lr_set* out = &sm.sets[c][out_i];
            // the candidate's own column (a team per chain: its whole scan table), and that of the state it starts from
            double2* table = SINGLE ? sm.t.tabs[out_i] : reinterpret_cast<double2*>(sm.t.pair.cols[c][out_i]);
            const double* base_col = SINGLE ? reinterpret_cast<const double*>(sm.t.tabs[base_i]) : sm.t.pair.cols[c][base_i];
            { const int lc_ = c; { const int c = c0 + lc_; (void)c; LR_SSTAMP(8); } }
            if (!rj) {
                lr_dd_prop p;
                lr_dd_draws dd;
                lr_dd_draws_load(&sm.draws[c][(it + 1) & 1], dd, lane);
                (void)base_col;
                double P;
                if (SINGLE) {
                    // (the tables are built by helper wave 2 + k from the proposed parameter vector)
                    lr_table_hand* hand = &sm.hand[k];
                    if (lane == 0) hand->out_idx = out_i, hand->base_idx = base_i;
                    P = lr_propose_dd<true, true, 2, true>(a, c0 + c, lane, it + 1, base->L[lane], p, table, ES, br_lds, &dd, hand, (int)iter + 1);
                } else {
                    P = lr_propose_dd<true, false, 1>(a, c0 + c, lane, it + 1, base->L[lane], p, table, ES, br_lds, &dd);
                }
                out->L[lane] = P;
                if (lane == 0) {
                    out->sc[LR_SET_HASTING] = p.hasting, out->sc[LR_SET_PRIOR] = p.prior, out->sc[LR_SET_LOG_U] = p.log_u;
                    out->sc[LR_SET_CONST] = 0.0;
                    out->isc[LR_SETI_GIBBS] = 0, out->isc[LR_SETI_INVALID] = 0, out->isc[LR_SETI_MOVE] = p.move;
                    out->isc[LR_SETI_KL] = out->isc[LR_SETI_KM] = base->isc[LR_SETI_KL];
                }
            } else {
                lr_rj_state s;
                lr_set_load(base, s, lane);
                { const int lc_ = c; { const int c = c0 + lc_; (void)c; LR_SSTAMP(0); } }
                lr_rj_prop p;
                lr_rj_draws d;
                lr_draws_load(&sm.draws[c][(it + 1) & 1], d, lane);
                if (SINGLE) {
                    // the table is built by helper wave 2 + k from the segments this wave stages; a no-op move copies all
                    // six planes of its base state's table
                    lr_table_hand* hand = &sm.hand[k];
                    if (lane == 0) hand->out_idx = out_i, hand->base_idx = base_i;
                    // (a copied table brings its pair planes along unless the scanner waves derive them anyway; the planes of a
                    // general-times table lie behind its slopes: S | E | 2E | slopes)
                    constexpr int n_copy = (MODE == 2 && !GENERAL) ? 2 * H : LR_UNIT_PLANES * H;
                    lr_propose_rj<true, lr_bins_per_lane(H), 2, true, true>(a, c0 + c, lane, &sm.scratch[wave], it + 1, s, p, table, ES, &d, br_lds,
                                                                            logbr_lds, base_col, n_copy, base->sc[LR_SET_CONST], hand, (int)iter + 1);
                } else {
                    lr_propose_rj<true, lr_bins_per_lane(H), 1, false>(a, c0 + c, lane, &sm.scratch[wave], it + 1, s, p, table, ES, &d, br_lds,
                                                                       logbr_lds, base_col, COL, base->sc[LR_SET_CONST]);
                }
                lr_set_store(out, s, p, lane);
                { const int lc_ = c; { const int c = c0 + lc_; (void)c; LR_SSTAMP(7); } }
            }
        }
        if (SINGLE && !RJ && ctx.cand_n > 0) {
            // a team per chain on a long scan: this wave's share of it (the tail of the block's slice), then its sums - it
            // may be the wave that arrives last and decides
            const lr_spec_dec_in dec_in = lr_spec_dec_fetch(sm, ctx, role0, role1, lane);
            double acc0 = 0.0, acc1 = 0.0;
            lr_persist_scan<H, GENERAL, 1, false, false, false, true>(reinterpret_cast<const char*>(sm.t.tabs[(role0 >> 2) & 3]), ctx.pk, ctx.cand_g0, ctx.cand_n,
                                                                      k * LR_WAVE + lane, 2 * LR_WAVE, &acc0, &acc1, nullptr);
            lr_spec_deliver<H, T, RJ, GENERAL, MODE>(sm, ctx, iter, dec_in, acc0, acc1, wave, lane);
        }
        LR_XSTAMP(dg_work);
        __syncthreads();                                                     // the decision and the candidates are in
        LR_XSTAMP(dg_wait1);
        if (k_team > 1 && sm.abort_flag) return;
        // ---- phase 2: what the deciding scanner wave found -> roles and books ----
        int d0, d1;
        {
            const lr_spec_decision* dc = &sm.dec[iter & 1];
            const double dv = (lane & 2) ? dc->lik_p[lane & 1] : dc->lik[lane & 1];     // lanes 0, 1 lik; 2, 3 lik_p
            const int sel = dc->sel;
            d0 = sel >> 1, d1 = sel & 1;
            if (d0) likA0 = lr_bcast(dv, 0);
            if (d1) likA1 = lr_bcast(dv, 1);
            bk.lik_p = c ? lr_bcast(dv, 3) : lr_bcast(dv, 2);
        }
        LR_XSTAMP(dg_b);
        role0 = lr_spec_turn(role0, d0), role1 = lr_spec_turn(role1, d1);
        LR_XSTAMP(dg_c);
        // clerk (waves 0 and 2): acceptance count and trace row of iteration `it` (LRF:321-359)
        if (k == 0 && mine_active) {
            bk.n_acc += c ? d1 : d0;
            if (it == bk.next_sample) {
                const int slot = bk.trace_slot;
                bk.trace_slot += 1;
                bk.next_sample += (unsigned long long)a.cfg.s_freq;
                if (slot < a.cfg.n_trace_slots && ctx.rank == 0) {
                    const lr_set* A = &sm.sets[c][(c ? role1 : role0) & 3];
                    const double likA = c ? likA1 : likA0;
                    lr_spec_trace(&sm.args, c0 + c, lane, slot, it, likA, A, rj ? 1 : 0);
                }
            }
        }
        LR_XSTAMP(dg_p2);
    }
    LR_XDUMP();
    if (k == 0 && mine_active && lane == 0) sm.final_p[c] = ((c ? role1 : role0) >> 2) & 3;
    // pending proposal, accepted state and scalars back to global memory, in the layout every engine shares
    if (k == 0 && mine_active && ctx.rank == 0) {
        const int role = c ? role1 : role0;
        lr_sets_to_global(a.state_f64 + (size_t)(c0 + c) * LR_STATE_ROWS * LR_ROW,
                          a.state_i32 + (size_t)(c0 + c) * LR_ISTATE_ROWS * LR_ROW, &sm.sets[c][role & 3],
                          &sm.sets[c][(role >> 2) & 3], c ? likA1 : likA0, ctx.it0 + (unsigned long long)ctx.n_iters, bk, rj, lane);
    }
}

// T threads: waves 0..3 are the candidate waves, the others scan.  The step arguments travel by value (kernarg
// segment -> scalar registers); both roles are inlined, their loops live in disjoint branches of the kernel.
template <int H, int T, bool RJ, bool GENERAL, int MODE>
__global__ __launch_bounds__(T, (T + 255) / 256) void lr_spec_kernel(lr_step_args a, lr_packed_lineages pk, long long n8,
                                                                     lr_spec_args x, long long n_iters) {
    constexpr int NW = T / LR_WAVE;
    constexpr int ENT = GENERAL ? 2 : 1;
    constexpr int COL = lr_spec_lds<H, NW, ENT>::COL;
    constexpr bool SINGLE = MODE != 0;
    constexpr int CPB = SINGLE ? 1 : 2;                   // (= x.cpb: the host picks the instantiation by it)
    static_assert(sizeof(lr_spec_lds<H, NW, ENT>) <= 160 * 1024, "the block's LDS image must fit a CU");
    __shared__ lr_spec_lds<H, NW, ENT> sm;
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int k_team = lr_spec_team_blocks<MODE>(x);
    const int team = blockIdx.x % x.n_teams, rank = blockIdx.x / x.n_teams;   // a team's blocks differ by a multiple of
    const int c0 = team * CPB;                                                 // n_teams: one XCD when 8 | n_teams
    const int C = a.cfg.n_chains;
    const int n_act = min(CPB, C - c0);
    constexpr bool rj = RJ;
    const bool act1 = n_act > 1;
    if (wave < n_act) {
        const int c = c0 + wave;
        lr_sets_from_global(a.state_f64 + (size_t)c * LR_STATE_ROWS * LR_ROW, a.state_i32 + (size_t)c * LR_ISTATE_ROWS * LR_ROW,
                            &sm.sets[wave][0], &sm.sets[wave][1], lane);
    }
    {
        // all tables start as zeros (entries no builder writes - and, with a team per chain, the unused half of every
        // entry - must not hold junk: they are gathered with count 0 / copied to the workspace at exit)
        double* z = reinterpret_cast<double*>(&sm.t);
        constexpr int N = sizeof(sm.t) / 8;
        for (int i = tid; i < N; i += T) z[i] = 0.0;
    }
    {
        // sets 2, 3 start as zeros (the parametric samplers write one row only; rows never written must not hold junk
        // that would reach the workspace at exit)
        int* z = reinterpret_cast<int*>(&sm.sets[0][0]);
        constexpr int WORDS = sizeof(lr_set) / 4;
        for (int i = tid; i < 2 * WORDS; i += T) z[2 * WORDS + i] = 0, z[6 * WORDS + i] = 0;
        if (!act1)
            for (int i = tid; i < 2 * WORDS; i += T) z[4 * WORDS + i] = 0;
    }
    for (int b = tid; b < 256; b += T) {
        const bool in = b < a.cfg.n_bins;
        sm.br[b] = (in && a.br_length) ? a.br_length[b] : 0.0;
        sm.logbr[b] = in ? a.log_br[b] : 0.0;
    }
    // a team whose exchange timed out in an earlier launch of this call has raised the status word: the run is void, and
    // the launches queued behind it end here instead of iterating on a state that was never written back
    if (tid == 0)
        sm.abort_flag = (k_team > 1 && __hip_atomic_load((lr_gu32*)x.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0,
        sm.final_p[0] = sm.final_p[1] = 1, sm.scan_arrive = 0, sm.plane_arrive = 0, sm.args = a, sm.hand[0].epoch = sm.hand[1].epoch = 0;
    if (tid < 2) sm.likA[tid] = (tid < n_act) ? a.state_f64[((size_t)(c0 + tid) * LR_STATE_ROWS + LR_ROW_SCALARS) * LR_ROW + LR_S_LIKA] : 0.0;
    lr_spec_ctx ctx;
    {
        const int* I0 = a.state_i32 + ((size_t)c0 * LR_ISTATE_ROWS + LR_IROW_SCALARS) * LR_ROW;
        // this block's slice of the packed lineage indices
        const long long per = (n8 + k_team - 1) / k_team;
        const long long g_lo = min((long long)rank * per, n8), g_hi = min(g_lo + per, n8);
        // (a team per chain: the tail of the slice is the candidate waves', scanned once their candidates stand)
        const long long n_c = (SINGLE && !RJ) ? ((g_hi - g_lo) * (long long)x.cand_share_q16) >> 16 : 0;
        ctx.pk = pk, ctx.g0 = g_lo, ctx.n8 = g_hi - g_lo - n_c, ctx.x = x, ctx.n_iters = n_iters;
        ctx.cand_g0 = g_hi - n_c, ctx.cand_n = n_c;
        ctx.it0 = (unsigned long long)(unsigned)I0[LR_I_IT_LO] | ((unsigned long long)(unsigned)I0[LR_I_IT_HI] << 32);
        ctx.c0 = c0, ctx.C = C, ctx.team = team, ctx.rank = rank, ctx.n_act = n_act;
    }
    __syncthreads();
    if (sm.abort_flag) return;
    // The pending proposals' entries (set 1) from global memory, where a chain's entries are its component of its pair
    // table, 2 doubles apart (lr_chain_table); the ACCEPTED states' tables (set 0) are not kept there: the RJ sampler
    // builds them here, once per launch, from the accepted rates and edges - the doubles their proposals were scored with.
    if (SINGLE) {
        const double* g = reinterpret_cast<const double*>(lr_chain_table(a, c0));
        for (int i = tid; i < COL; i += T) sm.t.tabs[1][GENERAL ? lr_pairgen_lds_entry(i, H) : i].x = g[2 * i];
    } else {
        for (int cc = 0; cc < n_act; ++cc) {
            const double* g = reinterpret_cast<const double*>(lr_chain_table(a, c0 + cc));
            for (int i = tid; i < COL; i += T) sm.t.pair.cols[cc][1][i] = g[2 * i];
        }
    }
    if (rj && (wave == 0 || wave == 2) && (wave >> 1) < n_act) {
        const int cc = wave >> 1;
        lr_rj_state s;
        lr_set_load(&sm.sets[cc][0], s, lane);
        double logL, logM;
        lr_stage_segments(&sm.scratch[wave], s.L, s.M, s.eL, s.eM, s.KL, s.KM, lane, &logL, &logM);
        if (SINGLE) {
            double* tabd = reinterpret_cast<double*>(sm.t.tabs[0]);
            (void)lr_build_tables_segments<lr_bins_per_lane(H), 2>(&sm.scratch[wave], s.eL, s.eM, s.KL, s.KM, sm.br, sm.logbr, a.cfg.model,
                                                                   a.cfg.n_bins, a.n_cls, a.H, reinterpret_cast<double2*>(tabd), lane,
                                                                   GENERAL ? LR_TAB_PAIRGEN : LR_TAB_UNIT, a.cfg.frac_birth, a.cfg.frac_death,
                                                                   GENERAL ? 6 * H : 2, nullptr);
            LR_WAVE_LDS_ORDER();
            if (GENERAL) lr_pair_planes_wave_general(tabd, H, a.cfg.n_bins, lane, 0);
            else lr_pair_planes_wave(tabd, H, a.cfg.n_bins, lane, 0);
        } else {
            (void)lr_build_tables_segments<lr_bins_per_lane(H), 1>(&sm.scratch[wave], s.eL, s.eM, s.KL, s.KM, sm.br, sm.logbr, a.cfg.model,
                                                                   a.cfg.n_bins, a.n_cls, a.H, reinterpret_cast<double2*>(sm.t.pair.cols[cc][0]),
                                                                   lane, GENERAL ? LR_TAB_PAIRGEN : LR_TAB_UNIT, a.cfg.frac_birth,
                                                                   a.cfg.frac_death, GENERAL ? 2 * H : 2, nullptr);
        }
    }
    __syncthreads();
    if (SINGLE) {
        // the pair planes of the pending proposal's table
        if (GENERAL) lr_pair_planes_block_general(sm.t.tabs[1], H, a.cfg.n_bins, tid, T);
        else lr_pair_planes_block(sm.t.tabs[1], H, a.cfg.n_bins, tid, T);
    } else {
        lr_build_scan_table<GENERAL>(sm.t.pair.scan, sm.t.pair.cols[0][1], act1 ? sm.t.pair.cols[1][1] : nullptr, H, a.cfg.n_bins, tid, T);
    }
    // draws of the first candidates (iteration it0 + 1); afterwards the last four scanner waves stay one iteration ahead
    // (a block on its own draws at the top of an iteration for the one after the next: it starts with two iterations' draws)
    if (wave >= NW - 2 && wave - (NW - 2) < n_act) {
        // (a team per chain: its helper waves make the draws of it0 + 2 in the first iteration)
        for (unsigned long long ahead = 1; ahead <= ((k_team == 1 && !SINGLE) ? 2ull : 1ull); ++ahead) {
            const unsigned long long itd = ctx.it0 + ahead;
            if (rj) {
                lr_spec_draw(a, c0 + (wave - (NW - 2)), lane, itd, &sm.draws[wave - (NW - 2)][itd & 1]);
            } else {
                lr_dd_draws dd;
                lr_make_dd_draws(a, c0 + (wave - (NW - 2)), lane, itd, dd);
                lr_dd_draws_store(&sm.draws[wave - (NW - 2)][itd & 1], dd, lane);
            }
        }
    }
    __syncthreads();
    if (SINGLE && (wave == 2 || wave == 3)) lr_spec_help_role<H, T, RJ, GENERAL, MODE>(sm, a, ctx, tid);
    else if (wave < 4) lr_spec_cand_role<H, T, RJ, GENERAL, MODE>(sm, a, ctx, tid);
    else lr_spec_scan_role<H, T, RJ, GENERAL, MODE>(sm, a, ctx, tid);
    __syncthreads();
    if (sm.abort_flag || rank != 0) return;
    // the pending proposals' entries back into their components of the pair tables
    if (SINGLE) {
        double* g = reinterpret_cast<double*>(lr_chain_table(a, c0));
        const double2* tab = sm.t.tabs[sm.final_p[0]];
        for (int i = tid; i < COL; i += T) g[2 * i] = tab[GENERAL ? lr_pairgen_lds_entry(i, H) : i].x;
    } else {
        for (int cc = 0; cc < n_act; ++cc) {
            double* g = reinterpret_cast<double*>(lr_chain_table(a, c0 + cc));
            const double* col = sm.t.pair.cols[cc][sm.final_p[cc]];
            for (int i = tid; i < COL; i += T) g[2 * i] = col[i];
        }
    }
}
