// Micro-benchmark: ds_read_b128 throughput of one CU by access pattern and number of waves.
// hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip && ./lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_IT 2000
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
__global__ void k(const int* __restrict__ idx, int stride_bytes, long long* cyc, unsigned int* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned int*>(lds)[i] = i;
    __syncthreads();
    unsigned int a[8];
    for (int j = 0; j < 8; ++j) a[j] = (unsigned int)(idx[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + j] * stride_bytes);
    u4 r0, r1, r2, r3, r4, r5, r6, r7;
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                     "ds_read_b128 %4, %12\n ds_read_b128 %5, %13\n ds_read_b128 %6, %14\n ds_read_b128 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    }
    long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r0.x ^ r1.y ^ r2.z ^ r3.w ^ r4.x ^ r5.y ^ r6.z ^ r7.w;
    if ((threadIdx.x & 63) == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}
// the scan's real access pattern: groups from a file (16 bytes each: birth index, count, 14 death indices), lane l of
// wave w takes groups (trip * n_waves + w) * 64 + l; per trip 15 ds_read_b128 (S at idx * 16, E at (H + idx) * 16)
__global__ void kreal(const unsigned char* __restrict__ groups, int n_groups, int H, int swz, long long* cyc, unsigned int* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned int*>(lds)[i] = i;
    __syncthreads();
    const int n_scan = blockDim.x;
    u4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14;
    unsigned int acc = 0;
    long long t0 = clock64();
    const uint4* g16 = reinterpret_cast<const uint4*>(groups);
    for (int rep = 0; rep < 20; ++rep) {
        int g = threadIdx.x;
        uint4 w = make_uint4(0, 0, 0, 0);
        if (g < n_groups) w = g16[g];
        while (g < n_groups) {
            const uint4 cur = w;
            const int nx = g + n_scan;
            if (nx < n_groups) w = g16[nx];
            const unsigned int ww[4] = {cur.x, cur.y, cur.z, cur.w};
            unsigned int a[15];
#pragma unroll
            for (int j = 0; j < 15; ++j) {
                const int b = j == 0 ? 0 : j + 1;                         // byte 0 birth index, bytes 2..15 death indices
                unsigned int e = (ww[b >> 2] >> (8 * (b & 3))) & 0xffu;
                if (j) e += (unsigned int)H;
                if (swz) e ^= (e >> 4) & 15u;
                a[j] = e * 16u;
            }
            asm volatile("ds_read_b128 %0, %15\n ds_read_b128 %1, %16\n ds_read_b128 %2, %17\n ds_read_b128 %3, %18\n ds_read_b128 %4, %19\n"
                         "ds_read_b128 %5, %20\n ds_read_b128 %6, %21\n ds_read_b128 %7, %22\n ds_read_b128 %8, %23\n ds_read_b128 %9, %24\n"
                         "ds_read_b128 %10, %25\n ds_read_b128 %11, %26\n ds_read_b128 %12, %27\n ds_read_b128 %13, %28\n ds_read_b128 %14, %29\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(r10),
                           "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14)
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
                           "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]));
            acc ^= r0.x ^ r1.x ^ r2.x ^ r3.x ^ r4.x ^ r5.x ^ r6.x ^ r7.x ^ r8.x ^ r9.x ^ r10.x ^ r11.x ^ r12.x ^ r13.x ^ r14.x;
            g = nx;
        }
    }
    long long t1 = clock64();
    sink[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
}

static void run_real(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("no %s\n", path); return; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* h = (unsigned char*)malloc(sz); if (fread(h, 1, sz, f) != (size_t)sz) return; fclose(f);
    const int n_groups = (int)(sz / 16);
    unsigned char* d; long long* cyc; unsigned int* sink;
    hipMalloc(&d, sz); hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 4096);
    hipMemcpy(d, h, sz, hipMemcpyHostToDevice);
    for (int swz = 0; swz < 2; ++swz)
        for (int threads : {256, 512, 896, 1024}) {
            hipLaunchKernelGGL(kreal, dim3(1), dim3(threads), 65536, 0, d, n_groups, 136, swz, cyc, sink);
            hipLaunchKernelGGL(kreal, dim3(1), dim3(threads), 65536, 0, d, n_groups, 136, swz, cyc, sink);
            long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
            long long lo = hw[0], hi = hw[1];
            for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
            const double reads = 20.0 * 15.0 * ((n_groups + 63) / 64);
            printf("%s (%d groups)%s waves=%2d: %.2f cycles per wave-read, one pass of the file %.2f us at 2.4 GHz\n", path, n_groups,
                   swz ? " xor-swizzled" : "", threads / 64, (hi - lo) / reads, (hi - lo) / 20.0 / 2400.0);
        }
}

// structure of the conflict rules: every lane reads the same entry in all 8 reads of the loop
static void run_probe() {
    int* idx; long long* cyc; unsigned int* sink;
    hipMalloc(&idx, 1024 * 8 * 4); hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 1024 * 4);
    int h[1024 * 8];
    struct { const char* name; int kind; int s; } P[] = {
        {"same bank class, entry = (l / s) * 16", 0, 1}, {"same bank class, entry = (l / s) * 16", 0, 2}, {"same bank class, entry = (l / s) * 16", 0, 4},
        {"same bank class, entry = (l / s) * 16", 0, 8}, {"same bank class, entry = (l / s) * 16", 0, 16}, {"same bank class, entry = (l / s) * 16", 0, 32},
        {"entry = l / s", 1, 1}, {"entry = l / s", 1, 2}, {"entry = l / s", 1, 4}, {"entry = l / s", 1, 8}, {"entry = l / s", 1, 16},
        {"entry = (l % s) * 16", 2, 2}, {"entry = (l % s) * 16", 2, 4}, {"entry = (l % s) * 16", 2, 8},
        {"entry = l % s", 3, 2}, {"entry = l % s", 3, 4}, {"entry = l % s", 3, 8}, {"entry = l % s", 3, 16}, {"entry = l % s", 3, 32},
        {"entry = 3 * l (stride 3)", 4, 3}, {"entry = 5 * l mod 128", 4, 5}, {"entry = 2 * l", 4, 2}, {"entry = 4 * l mod 128", 4, 4}, {"entry = 8 * l mod 128", 4, 8},
        {"entry = l + (l / 16)  (16 consecutive then skip)", 5, 0}, {"entry = (l / 2) * 3", 6, 0}};
    for (auto& p : P) {
        for (int t = 0; t < 1024; ++t)
            for (int j = 0; j < 8; ++j) {
                const int l = t % 64;
                int e = 0;
                if (p.kind == 0) e = (l / p.s) * 16;
                if (p.kind == 1) e = l / p.s;
                if (p.kind == 2) e = (l % p.s) * 16;
                if (p.kind == 3) e = l % p.s;
                if (p.kind == 4) e = (p.s * l) % 128;
                if (p.kind == 5) e = l + l / 16;
                if (p.kind == 6) e = (l / 2) * 3;
                h[t * 8 + j] = e;
            }
        hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
        const int threads = 512;
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, idx, 16, cyc, sink);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, idx, 16, cyc, sink);
        long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
        long long lo = hw[0], hi = hw[1];
        for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
        const double reads = (double)N_IT * 8 * (threads / 64);
        printf("%-52s s=%2d: %.2f cycles per wave-read\n", p.name, p.s, (hi - lo) / reads);
    }
}

// which lanes share a pass with lane i0: everybody reads entry 1, lane i0 entry 16, lane j entry 32 (same banks as 16)
static void run_mates() {
    int* idx; long long* cyc; unsigned int* sink;
    hipMalloc(&idx, 1024 * 8 * 4); hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 1024 * 4);
    int h[1024 * 8];
    for (int i0 : {0, 5, 20, 37}) {
        printf("lanes sharing a ds_read_b128 pass with lane %d:", i0);
        for (int j = 0; j < 64; ++j) {
            if (j == i0) continue;
            for (int t = 0; t < 1024; ++t)
                for (int q = 0; q < 8; ++q) h[t * 8 + q] = (t % 64 == i0) ? 16 : ((t % 64 == j) ? 32 : 1);
            hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, idx, 16, cyc, sink);
            long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
            long long lo = hw[0], hi = hw[1];
            for (int w = 0; w < 4; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
            const double c = (hi - lo) / ((double)N_IT * 8 * 4);
            if (c > 4.5) printf(" %d", j);
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == '-' && argv[1][1] == 'm') { run_mates(); return 0; }
    if (argc > 1 && argv[1][0] == '-') { run_probe(); return 0; }
    if (argc > 1) { run_real(argv[1]); return 0; }
    int* idx; long long* cyc; unsigned int* sink;
    hipMalloc(&idx, 1024 * 8 * 4); hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 1024 * 4);
    int h[1024 * 8];
    const char* names[] = {"linear (lane l -> entry l + 64 j mod 1024)", "random among 136 entries", "one entry (broadcast)",
                           "sorted-ish: lane l -> entry (l / 4 + j) mod 136", "random among 136, 32-byte stride", "random among 1024 entries"};
    for (int pat = 0; pat < 6; ++pat) {
        srand(7);
        for (int t = 0; t < 1024; ++t)
            for (int j = 0; j < 8; ++j) {
                int e = 0;
                if (pat == 0) e = (t % 64 + 64 * j) % 1024;
                if (pat == 1 || pat == 4) e = rand() % 136;
                if (pat == 2) e = 5;
                if (pat == 3) e = ((t % 64) / 4 + j) % 136;
                if (pat == 5) e = rand() % 1024;
                h[t * 8 + j] = e;
            }
        hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
        for (int threads : {256, 512, 768, 1024}) {
            const int stride = pat == 4 ? 32 : 16;
            hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, idx, stride, cyc, sink);
            hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, idx, stride, cyc, sink);
            long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
            long long lo = hw[0], hi = hw[1];
            for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
            const double reads = (double)N_IT * 8 * (threads / 64);
            printf("%-48s waves=%2d: %.2f cycles per wave-read (CU), %.0f B/clk/CU\n", names[pat], threads / 64, (hi - lo) / reads, 1024.0 * reads / (hi - lo));
        }
    }
    return 0;
}
