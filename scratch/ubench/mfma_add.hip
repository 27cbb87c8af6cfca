// mfma_add.hip - can the (idle) matrix cores take some of the scan loop's fp64 adds off the vector ALUs?
// v_mfma_f64_4x4x4_4b_f64 with A = a 4x4 identity per block is a lane-wise D = B + C (one f64 per lane in B, C, D).
// Measured here: (1) the lane layout (is it lane-wise with that A?), (2) cycles per instruction of the MFMA alone,
// (3) a loop of 14 independent fp64 accumulations per iteration (the scan loop's 14 adds per trip) with M of them done by
// the MFMA, at 1 / 2 / 4 waves per SIMD, one CU and all 256.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_add mfma_add.hip && ./mfma_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_IT 2048

__global__ void k_layout(double* out, int mode) {
    const int lane = threadIdx.x & 63;
    double a = mode == 0 ? (((lane & 3) == ((lane >> 2) & 3)) ? 1.0 : 0.0) : 1.0;
    double b = (double)lane, c = 1000.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    out[threadIdx.x] = d;
}

template <int M>
__global__ __launch_bounds__(1024) void k_mix(unsigned long long* out, double seed, double* sink) {
    const int lane = threadIdx.x & 63;
    const double a = ((lane & 3) == ((lane >> 2) & 3)) ? 1.0 : 0.0;
    double acc[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) acc[j] = seed + j + threadIdx.x;
    double g = seed * 0.5 + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("" : "+v"(g));
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            // spread the MFMAs evenly over the 14 operations
            const bool use_mfma = M > 0 && ((j * M) / 14 != ((j + 1) * M) / 14);
            if (use_mfma) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, g, acc[j], 0, 0, 0);
            else asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(g));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 14; ++j) s += acc[j];
    if (s == 12345.678) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M>
static void run(int grid, int threads, unsigned long long* d_out, double* d_sink) {
    unsigned long long h[256 * 16];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_mix<M>, dim3(grid), dim3(threads), 0, 0, d_out, 1.5, d_sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d_out, sizeof(unsigned long long) * grid * 16, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double mx = 0, sum = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < waves; ++w) {
            double c = (double)h[b * 16 + w] / N_IT;
            sum += c;
            if (c > mx) mx = c;
        }
    // cycles per iteration per SIMD = wave cycles per iteration / ... each wave runs its own 14 ops: SIMD cost per wave-iteration
    // = wave cycles / waves-per-SIMD
    const double per_simd = (sum / (grid * waves)) / (waves / 4.0 < 1 ? 1 : waves / 4.0);
    printf("  M=%2d grid=%3d waves/SIMD=%d : %.1f cycles per wave-iteration (max %.1f) -> %.1f SIMD cycles per 14 ops\n", M, grid, waves / 4 ? waves / 4 : 1,
           sum / (grid * waves), mx, per_simd);
}

int main() {
    double* d;
    hipMalloc(&d, 64 * sizeof(double));
    double h[64];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("layout, A = %s, B = lane, C = 1000:\n", mode == 0 ? "identity by (lane&3)==((lane>>2)&3)" : "ones");
        for (int l = 0; l < 64; ++l) printf("%6.0f%s", h[l], (l & 15) == 15 ? "\n" : " ");
    }
    unsigned long long* d_out;
    double* d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 16);
    hipMalloc(&d_sink, 8);
    for (int grid : {1, 256})
        for (int threads : {256, 512, 1024}) {
            run<0>(grid, threads, d_out, d_sink);
            run<1>(grid, threads, d_out, d_sink);
            run<2>(grid, threads, d_out, d_sink);
            run<3>(grid, threads, d_out, d_sink);
            run<4>(grid, threads, d_out, d_sink);
            run<5>(grid, threads, d_out, d_sink);
            run<7>(grid, threads, d_out, d_sink);
            run<14>(grid, threads, d_out, d_sink);
        }
    return 0;
}
