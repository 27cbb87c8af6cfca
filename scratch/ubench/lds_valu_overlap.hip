// Micro-benchmark: do LDS gathers and vector-ALU work of DIFFERENT waves of a CU overlap?  One block per CU, W waves; per
// loop trip a wave issues G ds_read_b128 (conflict-free or conflicting pattern), waits for them, then V fp64 instructions that
// consume the gathered values (the shape of the scan loops: gathers, wait, adds).  If the trip time of (G, V) is the larger
// of (G, 0) and (0, V) the two pipes overlap across waves; if it is their sum, a CU pays for both.
//   hipcc --offload-arch=gfx950 -O3 -o lds_valu_overlap lds_valu_overlap.hip && ./lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4000
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(2))) double d2;

template <int G, int V>
__global__ __launch_bounds__(1024) void k(int pattern, long long* cyc, double* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned int*>(lds)[i] = 0x3ff00000u * (i & 1);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned int a[8];
    for (int j = 0; j < 8; ++j) {
        int e = (lane + 64 * j) % 816;                       // conflict-free: consecutive entries
        if (pattern == 1) e = (lane * 37 + j * 11) % 136;    // scattered among 136 entries (conflicts as random death bins)
        a[j] = (unsigned int)e * 16u;
    }
    d2 r[8];
    for (int j = 0; j < 8; ++j) r[j] = d2{1.0, 1.0};
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        if (G > 0) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                         "ds_read_b128 %4, %12\n ds_read_b128 %5, %13\n ds_read_b128 %6, %14\n ds_read_b128 %7, %15\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            // 8 independent accumulation chains, inputs from the gathered registers
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[v & 7]) : "v"(r[(v + 3) & 7].x));
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j] + r[j].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}

template <int G, int V>
static double run(int waves, int pattern, int blocks) {
    long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, (size_t)blocks * 1024 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<G, V>), dim3(blocks), dim3(waves * 64), 65536, 0, pattern, cyc, sink);
    long long h[64]; hipMemcpy(h, cyc, 8 * 64, hipMemcpyDeviceToHost);
    long long lo = h[0], hi = h[1];
    for (int w = 0; w < waves; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    hipFree(cyc); hipFree(sink);
    return (double)(hi - lo) / N_IT;       // shader-clock cycles per trip of the whole block
}

int main() {
    for (int blocks : {1, 256})
        for (int pattern : {0, 1})
            for (int waves : {4, 8, 16}) {
                const double g = run<8, 0>(waves, pattern, blocks), v17 = run<0, 17>(waves, pattern, blocks), gv17 = run<8, 17>(waves, pattern, blocks);
                const double v40 = run<0, 40>(waves, pattern, blocks), gv40 = run<8, 40>(waves, pattern, blocks);
                printf("blocks=%3d %-12s waves=%2d: cycles per trip of the CU:  8 gathers %6.1f | 17 fp64 %6.1f | both %6.1f (max %6.1f, sum %6.1f) || "
                       "40 fp64 %6.1f | 8 gathers + 40 fp64 %6.1f (max %6.1f, sum %6.1f)\n",
                       blocks, pattern ? "scattered" : "conflict-free", waves, g, v17, gv17, g > v17 ? g : v17, g + v17, v40, gv40, g > v40 ? g : v40, g + v40);
            }
    return 0;
}
