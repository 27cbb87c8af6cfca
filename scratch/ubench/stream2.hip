// stream2.hip - what does a plain streaming read reach on this box?  One array / two arrays (ts, te), 16-byte loads,
// contiguous chunk per block or grid-stride, 1 / 2 / 4 loads per array in flight per thread.  The yardstick for the
// hbm_frac of lr_bin_unit_events and lr_bd_loglik_batch (bench.py abi).
//   hipcc --offload-arch=gfx950 -O3 -o stream2 stream2.hip && ./stream2
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ARRAYS, int DEPTH, bool STRIDE, int T>
__global__ __launch_bounds__(T) void k(const double2* __restrict__ a, const double2* __restrict__ b, long long n2, long long chunk2, double* out) {
    const long long step = STRIDE ? (long long)gridDim.x * T : T;
    long long i = STRIDE ? (long long)blockIdx.x * T + threadIdx.x : (long long)blockIdx.x * chunk2 + threadIdx.x;
    const long long end = STRIDE ? n2 : min(n2, (long long)(blockIdx.x + 1) * chunk2);
    double acc = 0.0;
    for (; i + (DEPTH - 1) * step < end; i += DEPTH * step) {
        double2 x[DEPTH], y[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            x[d] = a[i + d * step];
            if (ARRAYS == 2) y[d] = b[i + d * step];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += x[d].x + x[d].y;
            if (ARRAYS == 2) acc += y[d].x * y[d].y;
        }
    }
    for (; i < end; i += step) acc += a[i].x + (ARRAYS == 2 ? b[i].y : 0.0);
    if (acc == 123.456) out[0] = acc;
}
template <int ARRAYS, int DEPTH, bool STRIDE, int T>
void run(const double2* a, const double2* b, long long n, int blocks, double* out) {
    const long long n2 = n / 2, chunk2 = (n2 + blocks - 1) / blocks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ARRAYS, DEPTH, STRIDE, T>), dim3(blocks), dim3(T), 0, 0, a, b, n2, chunk2, out);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<ARRAYS, DEPTH, STRIDE, T>), dim3(blocks), dim3(T), 0, 0, a, b, n2, chunk2, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%.0e arrays=%d depth=%d %-7s T=%4d blocks=%5d: %7.1f us  %6.0f GB/s\n", (double)n, ARRAYS, DEPTH, STRIDE ? "stride" : "chunk", T, blocks, ms * 1e3,
           8.0 * n * ARRAYS / (ms * 1e-3) / 1e9);
}
int main() {
    for (long long n : {10000000ll, 30000000ll}) {
        double2 *a, *b; double* out;
        hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&out, 8);
        hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
        run<1, 1, false, 256>(a, b, n, 2048, out); run<1, 2, false, 256>(a, b, n, 2048, out); run<1, 4, true, 256>(a, b, n, 2048, out);
        run<2, 1, false, 256>(a, b, n, 2048, out); run<2, 2, false, 256>(a, b, n, 2048, out); run<2, 4, false, 256>(a, b, n, 2048, out);
        run<2, 1, true, 256>(a, b, n, 2048, out); run<2, 2, true, 256>(a, b, n, 2048, out); run<2, 4, true, 256>(a, b, n, 2048, out);
        run<2, 2, false, 1024>(a, b, n, 256, out); run<2, 2, true, 1024>(a, b, n, 256, out); run<2, 4, true, 1024>(a, b, n, 512, out);
        run<2, 2, true, 256>(a, b, n, 4096, out); run<2, 2, true, 512>(a, b, n, 1024, out);
        hipFree(a); hipFree(b); hipFree(out);
    }
    return 0;
}
