// Scan-loop lab: the unit-resolution scan of lr_scan.h on ONE CU against the real packed groups of cfg4, in variants,
// to see what bounds it.  hipcc --offload-arch=gfx950 -O3 -I../../literate_amd/csrc -o scan_lab scan_lab.hip
#include "lr_scan.h"
#include <cstdio>
#include <cstdlib>
#define H 136
#define REPS 40
// variant 0: the engine's loop; 1: gathers only (values xor-ed, no fp64 adds); 2: arithmetic only (no LDS: values from
// registers); 3: engine loop, unroll 2
// variant 4: all 15 gathers of a group issued back to back from one asm block, consumed in two halves
typedef __attribute__((ext_vector_type(2))) double d2;
__device__ __forceinline__ void scan_asm(const char* lbase, const uint4* idx8, int n, int sid, int n_scan, double* a0, double* a1) {
    double acc0 = *a0, acc1 = *a1;
    int i = sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) w = idx8[i];
    const unsigned int lb = (unsigned int)(size_t)lbase;     // LDS byte address of the table
    while (i < n) {
        const uint4 cur = w;
        const int nx = i + n_scan;
        if (nx < n) w = idx8[nx];
        unsigned int a[15];
        a[0] = lb + lr_grp_off<4>(cur, 0);
#pragma unroll
        for (int k = 0; k < LR_GRP; ++k) a[k + 1] = lb + lr_grp_off<4>(cur, k + 2);
        d2 S, E0, E1, E2, E3, E4, E5, E6, E7, E8, E9, E10, E11, E12, E13;
        asm volatile("ds_read_b128 %0, %15\n ds_read_b128 %1, %16 offset:2176\n ds_read_b128 %2, %17 offset:2176\n ds_read_b128 %3, %18 offset:2176\n"
                     "ds_read_b128 %4, %19 offset:2176\n ds_read_b128 %5, %20 offset:2176\n ds_read_b128 %6, %21 offset:2176\n ds_read_b128 %7, %22 offset:2176\n"
                     "ds_read_b128 %8, %23 offset:2176\n ds_read_b128 %9, %24 offset:2176\n ds_read_b128 %10, %25 offset:2176\n ds_read_b128 %11, %26 offset:2176\n"
                     "ds_read_b128 %12, %27 offset:2176\n ds_read_b128 %13, %28 offset:2176\n ds_read_b128 %14, %29 offset:2176\n s_waitcnt lgkmcnt(6)"
                     : "=&v"(S), "=&v"(E0), "=&v"(E1), "=&v"(E2), "=&v"(E3), "=&v"(E4), "=&v"(E5), "=&v"(E6), "=&v"(E7), "=&v"(E8), "=&v"(E9),
                       "=&v"(E10), "=&v"(E11), "=&v"(E12), "=&v"(E13)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
                       "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]));
        // S and E0..E7 have arrived
        const double cnt = (double)((cur.x >> 8) & 0xffu);
        d2 t0 = E0 + E1, t1 = E2 + E3, t2 = E4 + E5, t3 = E6 + E7;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(E8), "+v"(E9), "+v"(E10), "+v"(E11), "+v"(E12), "+v"(E13));
        d2 t4 = E8 + E9, t5 = E10 + E11, t6 = E12 + E13;
        const d2 u = ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + t6);
        acc0 += fma(cnt, S.x, u.x);
        acc1 += fma(cnt, S.y, u.y);
        i = nx;
    }
    *a0 = acc0, *a1 = acc1;
}

// variant 5: pair-slot format (7 slots of up to two lineages per group, 16-bit entry indices into a table that also
// holds the pair sums E[j] + E[j + d], d = 0..3): 8 gathers and 17 fp64 operations per 14 lineages x 2 chains
__device__ __forceinline__ unsigned int word_off(unsigned int v, int hi) {
    unsigned int r;
    const unsigned int sh = 4;
    if (hi) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(sh), "v"(v));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(sh), "v"(v));
    return r;
}
__device__ __forceinline__ void scan_pairs(const char* lbase, const uint4* idx, int n, int sid, int n_scan, double* a0, double* a1) {
    double acc0 = *a0, acc1 = *a1;
    int i = sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) w = idx[i];
    while (i < n) {
        const uint4 cur = w;
        const int nx = i + n_scan;
        if (nx < n) w = idx[nx];
        unsigned int sb;
        { const unsigned int sh = 4; asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(sb) : "v"(sh), "v"(cur.x)); }
        const double2 S = *reinterpret_cast<const double2*>(lbase + sb);
        const double cnt = (double)((cur.x >> 8) & 0xffu);
        double2 E[7];
        E[0] = *reinterpret_cast<const double2*>(lbase + word_off(cur.x, 1));
        E[1] = *reinterpret_cast<const double2*>(lbase + word_off(cur.y, 0));
        E[2] = *reinterpret_cast<const double2*>(lbase + word_off(cur.y, 1));
        E[3] = *reinterpret_cast<const double2*>(lbase + word_off(cur.z, 0));
        E[4] = *reinterpret_cast<const double2*>(lbase + word_off(cur.z, 1));
        E[5] = *reinterpret_cast<const double2*>(lbase + word_off(cur.w, 0));
        E[6] = *reinterpret_cast<const double2*>(lbase + word_off(cur.w, 1));
        const double u0 = ((E[0].x + E[1].x) + (E[2].x + E[3].x)) + ((E[4].x + E[5].x) + E[6].x);
        const double u1 = ((E[0].y + E[1].y) + (E[2].y + E[3].y)) + ((E[4].y + E[5].y) + E[6].y);
        acc0 += fma(cnt, S.x, u0);
        acc1 += fma(cnt, S.y, u1);
        i = nx;
    }
    *a0 = acc0, *a1 = acc1;
}

template <int VAR>
__device__ __forceinline__ void scan_once(const char* lbase, const uint4* idx8, int n, int sid, int n_scan, double* a0, double* a1) {
    if (VAR == 0) { lr_persist_scan_pair<H, 1>(lbase, idx8, n, sid, n_scan, a0, a1); return; }
    if (VAR == 3) { lr_persist_scan_pair<H, 2>(lbase, idx8, n, sid, n_scan, a0, a1); return; }
    if (VAR == 4) { scan_asm(lbase, idx8, n, sid, n_scan, a0, a1); return; }
    double acc0 = *a0, acc1 = *a1;
    int i = sid;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) w = idx8[i];
    const char* ebase = lbase + H * 16;
    while (i < n) {
        const uint4 cur = w;
        const int nx = i + n_scan;
        if (nx < n) w = idx8[nx];
        if (VAR == 1) {
            unsigned long long x = 0;
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const double2 v = *reinterpret_cast<const double2*>((k == 0 ? lbase : ebase) + lr_grp_off<4>(cur, k == 0 ? 0 : k + 1));
                x ^= (unsigned long long)__double_as_longlong(v.x);
            }
            acc0 += __longlong_as_double((long long)(x & 0xffff));
        } else {
            // same arithmetic on values made from the index words (no LDS traffic)
            double2 E[LR_GRP];
#pragma unroll
            for (int k = 0; k < LR_GRP; ++k) {
                const unsigned int o = lr_grp_off<4>(cur, k + 2);
                E[k] = make_double2(__hiloint2double(0x3ff00000, (int)o), __hiloint2double(0x3ff00000, (int)(o + 1)));
            }
            const unsigned int o = lr_grp_off<4>(cur, 0);
            const double2 S = make_double2(__hiloint2double(0x3ff00000, (int)o), 1.0);
            const double cnt = (double)((cur.x >> 8) & 0xffu);
            double t0[LR_GRP / 2], t1[LR_GRP / 2];
#pragma unroll
            for (int k = 0; k < LR_GRP / 2; ++k) t0[k] = E[2 * k].x + E[2 * k + 1].x, t1[k] = E[2 * k].y + E[2 * k + 1].y;
            const double u0 = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t0[4] + t0[5]) + t0[6]);
            const double u1 = ((t1[0] + t1[1]) + (t1[2] + t1[3])) + ((t1[4] + t1[5]) + t1[6]);
            acc0 += fma(cnt, S.x, u0);
            acc1 += fma(cnt, S.y, u1);
        }
        i = nx;
    }
    *a0 = acc0, *a1 = acc1;
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(const uint4* __restrict__ idx8, int n, long long* cyc, double* sink) {
    __shared__ double2 tab[2 * H];
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) tab[i] = make_double2(1.0 + i, 2.0 + i);
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) scan_once<VAR>(reinterpret_cast<const char*>(tab), idx8, n, threadIdx.x, blockDim.x, &a0, &a1);
    const long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}

__global__ __launch_bounds__(1024) void kp(const uint4* __restrict__ idx, int n, long long* cyc, double* sink) {
    __shared__ double2 tab[6 * H];
    for (int i = threadIdx.x; i < 6 * H; i += blockDim.x) tab[i] = make_double2(1.0 + i, 2.0 + i);
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) scan_pairs(reinterpret_cast<const char*>(tab), idx, n, threadIdx.x, blockDim.x, &a0, &a1);
    const long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}
static void run_pairs(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("no %s\n", path); return; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* h = (unsigned char*)malloc(sz); if (fread(h, 1, sz, f) != (size_t)sz) return; fclose(f);
    const int n = (int)(sz / 16);
    uint4* d; hipMalloc(&d, sz); hipMemcpy(d, h, sz, hipMemcpyHostToDevice);
    long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 8 * 1024);
    for (int threads : {512, 640, 768, 896, 1024}) {
        hipLaunchKernelGGL(kp, dim3(1), dim3(threads), 0, 0, d, n, cyc, sink);
        hipLaunchKernelGGL(kp, dim3(1), dim3(threads), 0, 0, d, n, cyc, sink);
        long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
        long long lo = hw[0], hi = hw[1];
        for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
        printf("%-34s waves=%2d blocks=  1: one pass %.2f us (%.1f cycles per wave-trip of 64 groups, CU-wide)\n", "pair-slot format", threads / 64,
               (hi - lo) / (double)REPS / 2400.0, (hi - lo) / (double)REPS / ((n + 63) / 64));
    }
}

template <int VAR>
static void run(const char* name, const uint4* d, int n, int threads, int blocks) {
    long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 8 * 1024 * 256);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, d, n, cyc, sink);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, d, n, cyc, sink);
    long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
    long long lo = hw[0], hi = hw[1];
    for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
    printf("%-34s waves=%2d blocks=%3d: one pass %.2f us (%.1f cycles per wave-trip of 64 groups, CU-wide)\n", name, threads / 64, blocks,
           (hi - lo) / (double)REPS / 2400.0, (hi - lo) / (double)REPS / ((n + 63) / 64));
    hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
    if (argc > 2) { run_pairs(argv[2]); return 0; }
    FILE* f = fopen(argc > 1 ? argv[1] : "scratch/ubench/idx8_cfg4.bin", "rb");
    if (!f) { printf("no index file\n"); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* h = (unsigned char*)malloc(sz); if (fread(h, 1, sz, f) != (size_t)sz) return 1; fclose(f);
    const int n = (int)(sz / 16);
    uint4* d; hipMalloc(&d, sz); hipMemcpy(d, h, sz, hipMemcpyHostToDevice);
    for (int blocks : {1})
        for (int threads : {512, 640, 768, 896, 1024}) {
            run<0>("engine loop", d, n, threads, blocks);
            run<3>("engine loop, unroll 2", d, n, threads, blocks);
            run<4>("asm gathers: 15 back to back", d, n, threads, blocks);
            run<1>("gathers only", d, n, threads, blocks);
            run<2>("arithmetic only", d, n, threads, blocks);
        }
    return 0;
}
