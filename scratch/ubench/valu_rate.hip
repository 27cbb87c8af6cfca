// Micro-benchmark: issue cycles per wave64 instruction for fp64 add / fma / mul, v_pk_add_f32, 64-bit integer add.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int OP>
__global__ void k(double* out, long long* cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = seed * 0.5, c = 1.0;
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        if (OP == 0) {
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == 1) {
            asm volatile("v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n v_fma_f64 %3, %3, %9, %8\n"
                         "v_fma_f64 %4, %4, %9, %8\n v_fma_f64 %5, %5, %9, %8\n v_fma_f64 %6, %6, %9, %8\n v_fma_f64 %7, %7, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (OP == 2) {
            asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                         "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == 3) {
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == 4) {
            asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                         "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == 5) {
            asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n"
                         "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));
        } else if (OP == 7) {
            // 32 x 32 + 64 -> 64 multiply-add (the Philox rounds): vcc is the carry-out operand
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i), "v"(i + 7) : "vcc");
        } else if (OP == 8) {
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 9) {
            // the same fma as a dependent chain: the latency a lone wave sees
            asm volatile("v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n"
                         "v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n v_fma_f64 %0, %0, %2, %1\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if (OP == 6) {
            float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, fb = (float)b;
            asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                         "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb));
            a0 = f0, a1 = f1, a2 = f2, a3 = f3;
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
    (void)w0; (void)w1;
}
template <int OP>
void run(const char* name, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * 8); hipMalloc(&cyc, 8 * 64);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0);
    long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost); long long lo = hw[0], hi = hw[1]; for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; } long long h = hi - lo; long long hh[2] = {h, h * 10 / 24};
    // waves per SIMD = threads / 256
    printf("%-16s threads=%4d: %.2f clock64 ticks per instruction per wave, %.2f per instruction per SIMD; %.2f ns per instruction per wave (clock64 at %.0f MHz)\n", name, threads,
           (double)h / (N_IT * 8.0), (double)h / (N_IT * 8.0) / (threads / 256.0 < 1 ? 1 : threads / 256.0), hh[1] * 10.0 / (N_IT * 8.0), (double)h / (hh[1] * 10.0) * 1e3);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int t : {64}) {
        run<0>("v_add_f64", t); run<9>("v_fma_f64 dependent", t); run<7>("v_mad_u64_u32", t); run<8>("v_rcp_f64", t); run<5>("v_cvt_f64_u32", t);
    }
    for (int t : {1024}) { run<7>("v_mad_u64_u32", t); run<8>("v_rcp_f64", t); }
    for (int t : {256, 1024}) {
        run<0>("v_add_f64", t); run<1>("v_fma_f64", t); run<2>("v_mul_f64", t); run<3>("v_pk_add_f32", t);
        run<4>("v_lshl_add_u64", t); run<5>("v_cvt_f64_u32", t); run<6>("v_add_f32", t);
    }
    return 0;
}
