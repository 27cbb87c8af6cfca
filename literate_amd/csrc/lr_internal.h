// lr_internal.h - host-side declarations shared between the translation units of libliterate_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/literate_hip.h"

#ifndef LR_SCAN_THREADS
#define LR_SCAN_THREADS 256
#endif
#define LR_SCAN_LDS_BUDGET (48 * 1024) /* table bytes per block we aim for (3 blocks / CU) */
#define LR_SCAN_LDS_MAX (150 * 1024)
#define LR_SCAN_LDS_WIDE (72 * 1024)   /* tables of the 16-chain general scan (lr_scan_wide_kernel): two blocks per CU */

struct lr_scan_plan {
    int cb;            // chains per block
    int groups;        // ceil(n_chains / cb)
    int tiles;         // lineage tiles
    long long chunk;   // lineages per tile (multiple of 2*LR_SCAN_THREADS)
    int tab_stride;    // double2 entries per chain = n_cls * 2 * H
    int H;             // entries reserved for each of the S and E tables (>= n_bins + 2)
    int fast;          // 1: templated immediate-offset kernel (n_cls == 1, H in {40,72,136,264})
    int unit;          // 1: unit-resolution tables (8-byte entries, fractions folded in); implies fast
    int n_cls;
    int threads;       // threads per block: LR_SCAN_THREADS, or LR_SCAN_WIDE_THREADS for the 16-chain general scan
    size_t lds_bytes;
};

// choose the launch shape of the lineage scan for (n lineages, n_chains, n_bins, model)
// `wide`: allow 16 chains of GENERAL tables per pass (lr_bd_loglik_batch; the engines keep their instantiated shapes)
int lr_plan_scan(long long n, int n_chains, int n_bins, int model, int unit, lr_scan_plan* plan, int wide = 0);

// The tile partials are laid out CHAIN-major: partials[chain * lr_tile_stride(tiles) + tile] - whoever adds a chain's tiles
// up reads one contiguous row (16 tiles to a 128-byte line), not one line per tile.  (Tile-major, a chain's ~2000 partials
// were ~2000 lines: the chain-step kernel of the launch-based engine waited 6 us for them.)
static inline __host__ __device__ int lr_tile_stride(int tiles) { return (tiles + 15) & ~15; }

// enqueue the scan of `n_chains` chains whose tables start at `tables` and whose rows of partials start at `partials`:
// partials[chain * partial_stride + tile] = sum over the tile's lineages, partial_stride = lr_tile_stride(plan.tiles)
int lr_launch_scan(const lr_scan_plan& plan, const double* ts, const double* te, long long n, double t0, int n_bins,
                   double end_time, const double2* tables, int n_chains, double* partials, int partial_stride,
                   hipStream_t stream);
// can the engine use the fused scan|step kernel for this plan?  (instantiated for a subset of shapes)
static inline bool lr_fused_supported(const lr_scan_plan& p) {
    if (!p.fast) return false;
    return p.unit ? (p.cb == 16 || p.cb == 8) : (p.cb == 8 || p.cb == 4);
}

static inline int lr_align_up(long long x, long long a) { return (int)((x + a - 1) / a * a); }
static inline long long lr_align_up64(long long x, long long a) { return (x + a - 1) / a * a; }
