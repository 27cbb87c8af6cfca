// lr_packscan.hip - the launch-based engine's scan over the PACKED lineages (few chains, or very many lineages).
//
// Until late in round 5 the launch-based engine re-read ts / te - 16 bytes per lineage - in every iteration: HBM-bound, 36 us
// per iteration of 16 chains on 1e7 lineages with the scan at 0.97 of what a kernel that only reads the two arrays reaches.
// The persistent engines never did: they scan the lineages as packed groups (lr_pack.hip: 16 bytes for up to 14 lineages
// of one birth bin - header = birth entry + count, seven 16-bit byte offsets of death entries or pre-summed pairs of them;
// csrc/lr_scan.h "unit resolution: pair slots"), 1.14 bytes per lineage, every lineage still scored through its own
// (birth, death) entries.  This kernel gives the launch-based engine the same stream, ONCE per iteration for all the
// chains of a partition: a 1024-thread block stages the pair tables of up to four chain pairs in LDS (six planes each:
// S, E and the pair-sum planes every block derives from E - 52 KB at H = 136, two blocks per CU), decodes a group once and
// scores it against every pair - per group and pair 8 ds_read_b128 + 17 fp64 operations, the operations and their order
// per (group, pair) those of the persistent scan (lr_persist_scan_pair_slice).  The loop is bound by the LDS gathers, not
// by memory (11.4 MB per pass at 1e7 lineages): 0.8 of the gather peak at 1e8 lineages, where the four-chain kernel's scan
// - one pair per decode, twelve of sixteen waves - reaches 0.5 inside its block.
// 16 chains x 1e7 / 3e7 / 1e8 lineages: 16.5 / 27 / 54-59 us per iteration against 35.5 / 95 / 285 with the scan of ts / te
// (profiles/r05_packed_scan.txt); general lineage times in the pair-general form (GENERAL below).
// Tile partials as everywhere: partials[chain * lr_tile_stride(tiles) + tile], summed by lr_chain_step_kernel.
// The planner (lr_mcmc.hip: lr_packscan_planned, lr_packed_mode) takes it wherever the launches run with few chains or
// long passes, never pipelined; long scans run in two partitions of the chains on their own streams.
#include <cstdlib>

#include "lr_engine.h"

#define LR_PACKSCAN_THREADS 1024

// GENERAL lineage times: the pair-general form of the tables and groups (csrc/lr_scan.h, lr_persist_scan_pair_general): six
// planes S | E | 2 E | their slopes scaled by 2^-32, a slot = one lineage or two of the run that die in the same bin, beside
// the groups three arrays of 16 bytes per group - the slots' in-bin fractions as 32-bit fixed point and the sum of the
// group's birth fractions as one double: 64 bytes per group of 14 lineages instead of 224 bytes of ts / te.  Per group and
// pair 16 ds_read_b128 and the persistent scan's operations.
template <int PAIRS, int H, bool GENERAL>
__global__ __launch_bounds__(LR_PACKSCAN_THREADS) void lr_packscan_kernel(const uint4* __restrict__ idx8, const uint4* __restrict__ frac,
                                                                          long long fstride, long long n8,
                                                                          const double2* __restrict__ tables, int chain_base /* even */,
                                                                          int n_chains /* chains [chain_base, n_chains) are scored */,
                                                                          int n_bins, int tiles, double* __restrict__ partials) {
    extern __shared__ double2 tab[];                       // [PAIRS][LR_UNIT_PLANES * H]
    __shared__ double red[LR_PACKSCAN_THREADS / LR_WAVE][2 * PAIRS];
    constexpr int PLANES = LR_UNIT_PLANES * H;             // entries of one pair's table in LDS
    constexpr int ENTS = (GENERAL ? 4 : 2) * H;            // ... and in global memory: S | E (| slopes of S | slopes of E)
    const int tid = threadIdx.x, lane = tid & (LR_WAVE - 1), wave = tid / LR_WAVE;
    const int tile = blockIdx.x, pair0 = chain_base / 2 + blockIdx.y * PAIRS;
    const int n_pairs = (n_chains + 1) / 2;
    const int np = min(PAIRS, n_pairs - pair0);
    // the planes global memory holds, of the block's pairs ([pair][ENTS] double2 = (chain 2 p, chain 2 p + 1)); pair slots
    // beyond the last pair read as zeros
    for (int i = tid; i < PAIRS * ENTS; i += LR_PACKSCAN_THREADS) {
        const int p = i / ENTS, j = i - p * ENTS;
        tab[p * PLANES + (GENERAL ? lr_pairgen_lds_entry(j, H) : j)] = p < np ? tables[(size_t)(pair0 + p) * ENTS + j] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    {
        // the derived planes of every pair, one pair per LR_PACKSCAN_THREADS / PAIRS threads
        constexpr int TPP = LR_PACKSCAN_THREADS / PAIRS;
        const int p = tid / TPP;
        if (GENERAL) lr_pair_planes_block_general(tab + p * PLANES, H, n_bins, tid - p * TPP, TPP);
        else lr_pair_planes_block(tab + p * PLANES, H, n_bins, tid - p * TPP, TPP);
    }
    __syncthreads();

    const long long per = (n8 + tiles - 1) / tiles;
    const long long g0 = min((long long)tile * per, n8), g1 = min(g0 + per, n8);
    double acc[2 * PAIRS];
#pragma unroll
    for (int k = 0; k < 2 * PAIRS; ++k) acc[k] = 0.0;
    const char* lbase = reinterpret_cast<const char*>(tab);
    // The group of the NEXT trip is loaded - by a hand-placed load the compiler does not see (lr_gload16_async, as in the
    // persistent scans: written as a plain load it is folded into the loop top and waited for on the spot, a memory round
    // trip per trip) - right after this trip's group is decoded, into the same registers; a lane's last trip loads a group
    // it will not score (the next tile's, or the zeros behind the data: lr_groups_alloc keeps more spare groups than a
    // block's stride).  32-bit byte offsets from the tile's uniform base.
    const char* gbase = lr_uniform_ptr(idx8 + g0);
    const char* fb0 = GENERAL ? lr_uniform_ptr(frac + g0) : gbase;
    const char* fb1 = GENERAL ? lr_uniform_ptr(frac + fstride + g0) : gbase;
    const char* fb2 = GENERAL ? lr_uniform_ptr(frac + 2 * fstride + g0) : gbase;
    static_assert(LR_FRAC_ARRAYS == 3, "three fraction arrays");
    constexpr unsigned int stride_b = LR_PACKSCAN_THREADS * 16u;
    constexpr int SLOPES = 3 * H * 16;                      // (general) bytes from a value entry to its slope entry
    const unsigned int end_b = (unsigned int)(g1 - g0) * 16u;
    unsigned int off = (unsigned int)tid * 16u;
    bool has = off < end_b;
    lr_u32x4 w = {0u, 0u, 0u, 0u}, f0 = {0u, 0u, 0u, 0u}, f1 = {0u, 0u, 0u, 0u}, f2 = {0u, 0u, 0u, 0u};
    lr_gload16_async(w, gbase, off);
    if (GENERAL) lr_gload16_async(f0, fb0, off), lr_gload16_async(f1, fb1, off), lr_gload16_async(f2, fb2, off);
    while (has) {
        if (GENERAL) lr_gload_wait<0>(w), lr_gload_wait<0>(f0, f1, f2);
        else lr_gload_wait<0>(w);
        unsigned int oS = w.x & 0xfff0u, c4 = w.x & 0xfu;
        unsigned int o0 = lr_word_off16(w.x, 1), o1 = lr_word_off16(w.y, 0), o2 = lr_word_off16(w.y, 1), o3 = lr_word_off16(w.z, 0),
                     o4 = lr_word_off16(w.z, 1), o5 = lr_word_off16(w.w, 0), o6 = lr_word_off16(w.w, 1);
        // every field is OUT of the group's registers before the next group is loaded into them (the compiler would fold the
        // word selects into the gathers' address arithmetic behind the load, keep the old registers alive and copy the new
        // ones while the load is in flight: literate_amd/check_async_loads.py)
        asm volatile("" : "+v"(oS), "+v"(c4), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6));
        double fe[LR_SLOTS], sfs = 0.0;
        if (GENERAL) {
            // the fractions came with the group: into doubles before their registers are refilled
            fe[0] = (double)f0.x, fe[1] = (double)f0.y, fe[2] = (double)f0.z, fe[3] = (double)f0.w, fe[4] = (double)f1.x, fe[5] = (double)f1.y,
            fe[6] = (double)f1.z;
            asm volatile("" : "+v"(fe[0]), "+v"(fe[1]), "+v"(fe[2]), "+v"(fe[3]), "+v"(fe[4]), "+v"(fe[5]), "+v"(fe[6]));
            // (the sum is used as it is: a move the compiler cannot postpone - read as plain registers it would be copied out
            // of f2 whenever convenient, after the refill below has been issued for one; as lr_persist_scan_pair_general)
            unsigned int s_lo, s_hi;
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(s_lo), "=&v"(s_hi) : "v"(f2.x), "v"(f2.y));
            sfs = __hiloint2double((int)s_hi, (int)s_lo);
        }
        const double cnt = (double)c4;
        off += stride_b, has = off < end_b;
        lr_gload16_async(w, gbase, off);
        if (GENERAL) lr_gload16_async(f0, fb0, off), lr_gload16_async(f1, fb1, off), lr_gload16_async(f2, fb2, off);
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            const char* b = lbase + (size_t)p * PLANES * sizeof(double2);
            if (GENERAL) {
                const unsigned int o[LR_SLOTS] = {o0, o1, o2, o3, o4, o5, o6};
                const double2 Sv = *reinterpret_cast<const double2*>(b + oS), Ss = *reinterpret_cast<const double2*>(b + oS + SLOPES);
                double p0[LR_SLOTS], p1[LR_SLOTS];
#pragma unroll
                for (int k = 0; k < LR_SLOTS; ++k) {
                    const double2 Ev = *reinterpret_cast<const double2*>(b + o[k]), Es = *reinterpret_cast<const double2*>(b + o[k] + SLOPES);
                    p0[k] = fma(fe[k], Es.x, Ev.x);
                    p1[k] = fma(fe[k], Es.y, Ev.y);
                }
                // the same fixed tree over the slots as at unit resolution, then the birth side of the whole group
                // (as lr_persist_scan_pair_general)
                const double u0 = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p0[4] + p0[5]) + p0[6]);
                const double u1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + p1[6]);
                acc[2 * p] += fma(sfs, Ss.x, fma(cnt, Sv.x, u0));
                acc[2 * p + 1] += fma(sfs, Ss.y, fma(cnt, Sv.y, u1));
            } else {
                const double2 S = *reinterpret_cast<const double2*>(b + oS);
                const double2 E0 = *reinterpret_cast<const double2*>(b + o0), E1 = *reinterpret_cast<const double2*>(b + o1);
                const double2 E2 = *reinterpret_cast<const double2*>(b + o2), E3 = *reinterpret_cast<const double2*>(b + o3);
                const double2 E4 = *reinterpret_cast<const double2*>(b + o4), E5 = *reinterpret_cast<const double2*>(b + o5);
                const double2 E6 = *reinterpret_cast<const double2*>(b + o6);
                // fixed pairwise tree over the slots, then the birth entry `count` times (as lr_persist_scan_pair_slice)
                const double u0 = ((E0.x + E1.x) + (E2.x + E3.x)) + ((E4.x + E5.x) + E6.x);
                const double u1 = ((E0.y + E1.y) + (E2.y + E3.y)) + ((E4.y + E5.y) + E6.y);
                acc[2 * p] += fma(cnt, S.x, u0);
                acc[2 * p + 1] += fma(cnt, S.y, u1);
            }
        }
    }
    // the idle loads of the last trip
    lr_gload_wait<0>(w);
    if (GENERAL) lr_gload_wait<0>(f0, f1, f2);
    // lanes -> wave (a fixed tree), waves in order
#pragma unroll
    for (int k = 0; k < 2 * PAIRS; ++k) {
        const double s = lr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (tid < 2 * PAIRS && 2 * pair0 + tid < n_chains) {
        double t = 0.0;
        for (int v = 0; v < LR_PACKSCAN_THREADS / LR_WAVE; ++v) t += red[v][tid];
        partials[(size_t)(2 * pair0 + tid) * lr_tile_stride(tiles) + tile] = t;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------

// chain pairs a block takes: the most of 8, 4, 2, 1 whose tables fit ~3/4 of a CU's LDS and that the chains can fill
int lr_packscan_pairs(const lr_scan_plan& p, int n_chains) {
    const int n_pairs = (n_chains + 1) / 2;
    // (four pairs per block = two blocks per CU measured best: 16 chains x 1e7 / 3e7 / 1e8 lineages 17.7 / 27.4 / 64.0 us per
    // iteration against 24.2 / 33.1 / 69.5 with eight pairs and one block per CU; two pairs: 16.2 / 26.6 / 65.9)
    static const int cap = getenv("LR_PACKSCAN_PAIRS") ? atoi(getenv("LR_PACKSCAN_PAIRS")) : 4;      // (A/B runs)
    int pairs = (cap == 1 || cap == 2 || cap == 8) ? cap : 4;
    if (p.unit == LR_TAB_PAIRGEN && pairs > 4) pairs = 4;      // (general times: sixteen gathers and seven fractions per group and pair)
    while (pairs > 1 && ((size_t)pairs * LR_UNIT_PLANES * p.H * sizeof(double2) > 120 * 1024 || pairs / 2 >= n_pairs)) pairs >>= 1;
    return pairs;
}

// engine_mode 0 (unless LR_PACKED_SCAN=0) and 7; engine_mode 1 keeps the scan of ts / te
bool lr_packscan_eligible(const lr_mcmc_config* cfg, const lr_scan_plan& p) {
    static const int env = getenv("LR_PACKED_SCAN") ? atoi(getenv("LR_PACKED_SCAN")) : 1;
    if (cfg->engine_mode != 7 && (!env || cfg->engine_mode != 0)) return false;     // (7 = asked for; auto unless switched off)
    if ((p.unit != LR_TAB_UNIT && p.unit != LR_TAB_PAIRGEN) || !p.fast || p.cb < 2) return false;    // pair tables
    if (p.H != 40 && p.H != 72 && p.H != 136 && p.H != 264 && p.H != LR_H_WIDE) return false;
    // the packing holds lineage indices as int32 and the groups are addressed by 32-bit offsets (as lr_persist_eligible)
    if (cfg->n_lineages >= (1ll << 31) - 1 || lr_groups_alloc(cfg->n_lineages) * 16 >= (1ll << 32)) return false;
    return true;
}

// tiles of the packed scan (every pair group scans every tile)
void lr_packscan_plan(const lr_mcmc_config* cfg, lr_scan_plan* p, int cus) {
    // chains per partition boundary: eight at unit resolution (the pair tables lie pair by pair whatever the grouping), so
    // that sixteen chains already make two partitions - each on its own stream, one's step kernel under the other's scan
    if (p->unit == LR_TAB_UNIT && p->cb > 8) p->cb = 8, p->groups = (cfg->n_chains + 7) / 8;
    const int pairs = lr_packscan_pairs(*p, cfg->n_chains);
    const int pair_groups = ((cfg->n_chains + 1) / 2 + pairs - 1) / pairs;
    // two blocks per CU and pair group where that leaves a block at least four trips of its 1024 threads, else one
    // (1e7 lineages: 16.5 us per iteration with 128 tiles per pair group, 17.7 with 256, 20.3 with 349)
    static const int rounds_env = getenv("LR_PACKSCAN_ROUNDS") ? atoi(getenv("LR_PACKSCAN_ROUNDS")) : 0;
    const long long n8_est = (cfg->n_lineages + LR_GRP - 1) / LR_GRP;
    int rounds = rounds_env > 0 ? rounds_env : 2;                 // blocks per CU, all pair groups together
    if (rounds_env <= 0 && n8_est * pair_groups < 4ll * LR_PACKSCAN_THREADS * cus * rounds) rounds = 1;
    long long tiles = (long long)cus * rounds / pair_groups;
    const long long max_tiles = (n8_est + 2 * LR_PACKSCAN_THREADS - 1) / (2 * LR_PACKSCAN_THREADS);
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    p->tiles = (int)tiles;
    // (chunk keeps its meaning for the raw-stream scan, which runs if the packing refuses the input: unsorted beyond LR_MAX_RUNS)
    const long long unit = 2 * LR_SCAN_THREADS;
    p->chunk = lr_align_up64((cfg->n_lineages + tiles - 1) / tiles, unit);
    p->tiles = (int)((cfg->n_lineages + p->chunk - 1) / p->chunk);
}

template <int PAIRS, int H, bool GENERAL>
static int lr_packscan_launch(const lr_engine* e, int base, int count, hipStream_t stream) {
    const size_t lds = (size_t)PAIRS * LR_UNIT_PLANES * H * sizeof(double2);
    const void* fn = reinterpret_cast<const void*>(&lr_packscan_kernel<PAIRS, H, GENERAL>);
    if (lds > 64 * 1024) {
        // (per call: the attribute belongs to the function on the CURRENT device)
        const hipError_t he = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (he != hipSuccess) return (int)he;
    }
    const int n_pairs = (base + count + 1) / 2 - base / 2;
    const dim3 grid(e->plan.tiles, (n_pairs + PAIRS - 1) / PAIRS);
    hipLaunchKernelGGL((lr_packscan_kernel<PAIRS, H, GENERAL>), grid, dim3(LR_PACKSCAN_THREADS), lds, stream,
                       (const uint4*)(e->ws + e->lay.lineage_idx), (const uint4*)(e->ws + e->lay.lineage_frac), (long long)e->n8_alloc, e->n8,
                       (const double2*)(e->ws + e->lay.tables), base, base + count, e->cfg.n_bins, e->plan.tiles,
                       (double*)(e->ws + e->lay.partials));
    return (int)hipGetLastError();
}

template <int H>
static int lr_packscan_launch_h(const lr_engine* e, int base, int count, hipStream_t stream) {
    if (e->plan.unit == LR_TAB_PAIRGEN) {
        switch (lr_packscan_pairs(e->plan, e->cfg.n_chains)) {
            case 4: return lr_packscan_launch<4, H, true>(e, base, count, stream);
            case 2: return lr_packscan_launch<2, H, true>(e, base, count, stream);
            default: return lr_packscan_launch<1, H, true>(e, base, count, stream);
        }
    }
    switch (lr_packscan_pairs(e->plan, e->cfg.n_chains)) {
        case 8: return lr_packscan_launch<8, H, false>(e, base, count, stream);
        case 4: return lr_packscan_launch<4, H, false>(e, base, count, stream);
        case 2: return lr_packscan_launch<2, H, false>(e, base, count, stream);
        default: return lr_packscan_launch<1, H, false>(e, base, count, stream);
    }
}

// one pass of the packed lineages for the chains [base, base + count) (base even: a partition of the engine)
int lr_launch_packscan(const lr_engine* e, int base, int count, hipStream_t stream) {
    if (e->n8 <= 0 || (base & 1) || count < 1) return LR_ERR_STATE;
    switch (e->plan.H) {
        case 40: return lr_packscan_launch_h<40>(e, base, count, stream);
        case 72: return lr_packscan_launch_h<72>(e, base, count, stream);
        case 136: return lr_packscan_launch_h<136>(e, base, count, stream);
        case 264: return lr_packscan_launch_h<264>(e, base, count, stream);
        case LR_H_WIDE: return lr_packscan_launch_h<LR_H_WIDE>(e, base, count, stream);      // (255 .. 512 bins: two pairs per block)
        default: return LR_ERR_SIZE;
    }
}
