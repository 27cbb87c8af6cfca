// lr_spec.hip - host side of the speculative team engine (kernels: lr_spec.h)
#include <cstdlib>

#include "lr_engine.h"
#include "lr_spec.h"

int lr_launch_spec(lr_engine* e, const lr_step_args& a, const lr_packed_lineages& pk, int64_t n_iters, hipStream_t stream) {
    const int cpb = e->lay.spec_chains_per_team == 1 ? 1 : 2;
    const int blocks = (e->cfg.n_chains + cpb - 1) / cpb;          // teams
    const bool general = e->plan.unit == LR_TAB_PAIRGEN;
            lr_spec_args x;
            x.xchg = (unsigned long long*)(e->ws + e->lay.xchg);
            x.status = (unsigned int*)(e->ws + e->lay.status);
            x.team_blocks = e->lay.team_blocks, x.n_teams = blocks, x.cpb = cpb;
            const size_t xbytes = (size_t)2 * blocks * LR_TEAM_MAX * LR_SPEC_GRANULES * 8;
            for (int64_t done = 0; done < n_iters;) {
                const int64_t n = (n_iters - done > 4096) ? 4096 : n_iters - done;   // keep single launches short
                // epochs count from 1 inside every launch: all granules start at zero (Guideline 16, "Re-initialise every call")
                if (x.team_blocks > 1) {
                    const hipError_t he = hipMemsetAsync(x.xchg, 0, xbytes, stream);
                    if (he != hipSuccess) return (int)he;
                }
                const dim3 grid((unsigned)(blocks * x.team_blocks)), blk(cpb == 1 ? LR_SPEC_THREADS_SINGLE : LR_SPEC_THREADS);
#define LR_SPEC_LAUNCH(HH, GG)                                                                                                \
    if (e->cfg.sampler == 0 && cpb == 1)                                                                                      \
        hipLaunchKernelGGL((lr_spec_kernel<HH, LR_SPEC_THREADS_SINGLE, true, GG, true>), grid, blk, 0, stream, a, pk, e->n8, x, (long long)n); \
    else if (e->cfg.sampler == 0)                                                                                             \
        hipLaunchKernelGGL((lr_spec_kernel<HH, LR_SPEC_THREADS, true, GG, false>), grid, blk, 0, stream, a, pk, e->n8, x, (long long)n); \
    else if (cpb == 1)                                                                                                        \
        hipLaunchKernelGGL((lr_spec_kernel<HH, LR_SPEC_THREADS_SINGLE, false, GG, true>), grid, blk, 0, stream, a, pk, e->n8, x, (long long)n); \
    else                                                                                                                      \
        hipLaunchKernelGGL((lr_spec_kernel<HH, LR_SPEC_THREADS, false, GG, false>), grid, blk, 0, stream, a, pk, e->n8, x, (long long)n)
                // (the eight candidate tables + the scan table: 352 H bytes at unit resolution, 608 H on general times, where
                // H = 264 does not fit the LDS and is never planned for this kernel, see lr_persist_variant)
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
