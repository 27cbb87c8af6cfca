// lr_spec.hip - host side of the speculative team engine (kernels: lr_spec.h)
#include <cstdlib>

#include "lr_engine.h"
#include "lr_spec.h"

int lr_launch_spec(lr_engine* e, const lr_step_args& a, const lr_packed_lineages& pk, int64_t n_iters, hipStream_t stream) {
    const int cpb = e->lay.spec_chains_per_team == 1 ? 1 : 2;
    const int mode = lr_spec_mode(e);
    const int blocks = (e->cfg.n_chains + cpb - 1) / cpb;          // teams
    const bool general = e->plan.unit == LR_TAB_PAIRGEN;
            lr_spec_args x;
            x.xchg = (unsigned long long*)(e->ws + e->lay.xchg);
            x.status = (unsigned int*)(e->ws + e->lay.status);
            x.team_blocks = e->lay.team_blocks, x.n_teams = blocks, x.cpb = cpb;
            {
                // A team per chain under a PARAMETRIC sampler: the tables are the helper waves', so the two candidate waves
                // idle after ~0.9 us of an iteration in which a scanner wave makes a trip per ~0.3 us.  On a long scan they take
                // a share: with W wave-trips per block and d = 3 trips they are late by, all ten waves finish together when the
                // eight scanners make t = (W + 2 d) / 10 trips and the two candidates t - d: share = 2 (t - d) / W, nothing when
                // the scan is shorter than their build.  (The RJ sampler's candidate waves are busy for ~1.9 us and a wave that
                // arrives last also decides: every share measured slower there.)  LR_SPEC_CAND_PCT overrides it (A/B runs).
                const double W = (double)((e->n8 + x.team_blocks - 1) / x.team_blocks) / 64.0;
                const double d = 3.0;
                const double t = (W + 2.0 * d) / 10.0;
                // (not in a team: the wave that arrives last also runs the exchange - measured 10-18 % slower with a share)
                double share = (e->cfg.sampler != 0 && x.team_blocks == 1 && t > d && W > 0.0) ? 2.0 * (t - d) / W : 0.0;
                const char* env = getenv("LR_SPEC_CAND_PCT");
                if (env) share = atof(env) / 100.0;
                if (share > 0.5) share = 0.5;
                if (share < 0.0) share = 0.0;
                x.cand_share_q16 = cpb == 1 ? (int)(share * 65536.0) : 0;
            }
            const size_t xbytes = (size_t)2 * blocks * LR_TEAM_MAX * LR_SPEC_GRANULES * 8;
            for (int64_t done = 0; done < n_iters;) {
                const int64_t n = (n_iters - done > 4096) ? 4096 : n_iters - done;   // keep single launches short
                // epochs count from 1 inside every launch: all granules start at zero (Guideline 16, "Re-initialise every call")
                if (x.team_blocks > 1) {
                    const hipError_t he = hipMemsetAsync(x.xchg, 0, xbytes, stream);
                    if (he != hipSuccess) return (int)he;
                }
                const dim3 grid((unsigned)(blocks * x.team_blocks)), blk(cpb == 1 ? LR_SPEC_THREADS_SINGLE : LR_SPEC_THREADS);
#define LR_SPEC_LAUNCH_M(HH, GG, RR, MM) \
    hipLaunchKernelGGL((lr_spec_kernel<HH, (MM) != 0 ? LR_SPEC_THREADS_SINGLE : LR_SPEC_THREADS, RR, GG, MM>), grid, blk, 0, stream, a, pk, e->n8, x, (long long)n)
#define LR_SPEC_LAUNCH(HH, GG)                                                         \
    if (e->cfg.sampler == 0) {                                                         \
        if (mode == 3) LR_SPEC_LAUNCH_M(HH, GG, true, 3);                              \
        else if (mode == 2) LR_SPEC_LAUNCH_M(HH, GG, true, 2);                         \
        else if (mode == 1) LR_SPEC_LAUNCH_M(HH, GG, true, 1);                         \
        else LR_SPEC_LAUNCH_M(HH, GG, true, 0);                                        \
    } else {                                                                           \
        if (mode == 3) LR_SPEC_LAUNCH_M(HH, GG, false, 3);                             \
        else if (mode == 2) LR_SPEC_LAUNCH_M(HH, GG, false, 2);                        \
        else if (mode == 1) LR_SPEC_LAUNCH_M(HH, GG, false, 1);                        \
        else LR_SPEC_LAUNCH_M(HH, GG, false, 0);                                       \
    }
                // (general times: H = 264 is never planned for this kernel, see lr_persist_variant)
                if (general) {
                    switch (e->plan.H) {
                        case 40: LR_SPEC_LAUNCH(40, true); break;
                        case 72: LR_SPEC_LAUNCH(72, true); break;
                        case 136: LR_SPEC_LAUNCH(136, true); break;
                        default: return LR_ERR_SIZE;
                    }
                } else {
                    switch (e->plan.H) {
                        case 40: LR_SPEC_LAUNCH(40, false); break;
                        case 72: LR_SPEC_LAUNCH(72, false); break;
                        case 136: LR_SPEC_LAUNCH(136, false); break;
                        case 264: LR_SPEC_LAUNCH(264, false); break;
                        default: return LR_ERR_SIZE;
                    }
                }
#undef LR_SPEC_LAUNCH
#undef LR_SPEC_LAUNCH_M
                const int rc = (int)hipGetLastError();
                if (rc) return rc;
                done += n;
            }
            return LR_OK;
}

#ifdef LR_DIAG
// diagnostic builds: the stamp buffers are per translation unit (static __device__ in lr_step.h)
extern "C" int lr_diag_dump_step_spec(unsigned long long* host_out, int n_words) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_step), (size_t)n_words * 8);
}
extern "C" int lr_diag_dump_seg_spec(unsigned long long* host_out, int n_words, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lr_diag_seg), (size_t)n_words * 8);
    if (reset) {
        static unsigned long long zeros[64 * 16];
        rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(lr_diag_seg), zeros, sizeof(zeros));
    }
    return rc;
}
#endif
