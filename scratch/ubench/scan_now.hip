// Scan-loop lab, round 5: the CURRENT unit-resolution scan loop (lr_persist_scan_pair, hand-placed loads, wave-uniform trip
// count) of lr_scan.h on the real packed groups of cfg4 (scratch/ubench/dump_idx_now.py), alone on a CU: W waves, no
// steppers, no barriers - what does one pass cost, and what do its parts cost?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -I../../literate_amd/csrc -o scan_now scan_now.hip
#include "lr_scan.h"
#include <cstdio>
#include <cstdlib>
#define HH 136
#define REPS 50
// VAR 0: the engine's loop.  1: gathers only (the values xor-ed).  2: arithmetic only (values made from the offsets, no LDS).
// 3: the engine's loop on ONE group per lane (no group loads inside the loop).
template <int VAR>
__device__ __forceinline__ void pass(const char* lbase, const uint4* idx8, int n, int sid, int n_scan, double* a0, double* a1) {
    if (VAR == 0) {
        lr_scan_tail tail;
        lr_persist_scan_pair<HH, 1, true>(lbase, idx8, n, sid, n_scan, a0, a1, nullptr, &tail);
        lr_scan_drain(tail);
        return;
    }
    double acc0 = *a0, acc1 = *a1;
    const int nu = __builtin_amdgcn_readfirstlane(n);
    int i0 = __builtin_amdgcn_readfirstlane(sid);
    uint4 w = idx8[sid];
    while (i0 < nu) {
        const uint4 cur = w;
        i0 += n_scan;
        if (VAR != 3) w = idx8[min(sid + (i0 - __builtin_amdgcn_readfirstlane(sid)), n + 1023)];
        const unsigned int oS = cur.x & 0xfff0u;
        const double cnt = (double)(cur.x & 0xfu);
        const unsigned int o[7] = {lr_word_off16(cur.x, 1), lr_word_off16(cur.y, 0), lr_word_off16(cur.y, 1), lr_word_off16(cur.z, 0),
                                   lr_word_off16(cur.z, 1), lr_word_off16(cur.w, 0), lr_word_off16(cur.w, 1)};
        double2 S, E[7];
        if (VAR == 2) {
            S = make_double2(__hiloint2double(0x3ff00000, (int)oS), 1.0);
#pragma unroll
            for (int k = 0; k < 7; ++k) E[k] = make_double2(__hiloint2double(0x3ff00000, (int)o[k]), __hiloint2double(0x3ff00000, (int)o[k] + 1));
        } else {
            S = *reinterpret_cast<const double2*>(lbase + oS);
#pragma unroll
            for (int k = 0; k < 7; ++k) E[k] = *reinterpret_cast<const double2*>(lbase + o[k]);
        }
        if (VAR == 1) {
            unsigned long long x = (unsigned long long)__double_as_longlong(S.x);
#pragma unroll
            for (int k = 0; k < 7; ++k) x ^= (unsigned long long)__double_as_longlong(E[k].x);
            acc0 += __longlong_as_double((long long)(x & 0xffff));
        } else {
            const double u0 = ((E[0].x + E[1].x) + (E[2].x + E[3].x)) + ((E[4].x + E[5].x) + E[6].x);
            const double u1 = ((E[0].y + E[1].y) + (E[2].y + E[3].y)) + ((E[4].y + E[5].y) + E[6].y);
            acc0 += fma(cnt, S.x, u0);
            acc1 += fma(cnt, S.y, u1);
        }
    }
    *a0 = acc0, *a1 = acc1;
}

template <int VAR, bool BARRIER>
__global__ __launch_bounds__(1024) void k(const uint4* __restrict__ idx8, int n, long long* cyc, double* sink) {
    __shared__ double2 tab[LR_UNIT_PLANES * HH];
    for (int i = threadIdx.x; i < LR_UNIT_PLANES * HH; i += blockDim.x) tab[i] = make_double2(1.0 + i, 2.0 + i);
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < REPS; ++r) {
        pass<VAR>(reinterpret_cast<const char*>(tab), idx8, __builtin_amdgcn_readfirstlane(n), threadIdx.x, __builtin_amdgcn_readfirstlane((int)blockDim.x), &a0, &a1);
        if (BARRIER) __syncthreads();
    }
    const long long t1 = wall_clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}

template <int VAR, bool BARRIER>
static void run(const char* name, const uint4* d, int n, int threads, int blocks) {
    long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 8 * 1024 * 256);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<VAR, BARRIER>), dim3(blocks), dim3(threads), 0, 0, d, n, cyc, sink);
    long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
    long long lo = hw[0], hi = hw[1];
    for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
    const double us = (hi - lo) / 100.0 / REPS;              // wall clock: 100 MHz
    const double trips = (n + 63) / 64;
    printf("%-46s waves=%2d blocks=%3d: one pass %.2f us = %.0f cycles (2.4 GHz) per wave-trip CU-wide; LDS data path needs %.2f us\n", name, threads / 64, blocks,
           us, us * 2400.0 / trips, trips * 8 * 4.45 / 2400.0);
    hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
    FILE* f = fopen(argc > 1 ? argv[1] : "gpurun_out/idx8_now.bin", "rb");
    if (!f) { printf("no index file\n"); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* h = (unsigned char*)calloc(sz + 16 * 4096, 1); if (fread(h, 1, sz, f) != (size_t)sz) return 1; fclose(f);
    const int n = (int)(sz / 16);
    uint4* d; hipMalloc(&d, sz + 16 * 4096); hipMemcpy(d, h, sz + 16 * 4096, hipMemcpyHostToDevice);   // zero groups behind the data
    printf("%d groups\n", n);
    for (int blocks : {1, 256})
        for (int threads : {256, 512, 768, 896, 1024}) {
            run<0, false>("engine loop", d, n, threads, blocks);
            run<0, true>("engine loop + a barrier per pass", d, n, threads, blocks);
            run<1, false>("gathers only", d, n, threads, blocks);
            run<2, false>("arithmetic only (no LDS)", d, n, threads, blocks);
            run<3, false>("plain loop, one group per lane (no loads)", d, n, threads, blocks);
        }
    return 0;
}
