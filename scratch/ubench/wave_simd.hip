// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8],
// sh [12], se [15:13] on gfx9.)  Prints, for block sizes 1024 / 768 / 512, the SIMD of every wave of a few blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    const int wave = threadIdx.x / 64;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = hw;
    // stay resident for a while so that blocks spread over the CUs as in a persistent launch
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000ull) {}
}
int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 16 * 4);
    for (int T : {1024, 768, 512}) {
        hipMemset(d, 0xff, 256 * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(256), dim3(T), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(256 * 16);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("T = %d: simd of waves 0..%d (slot in brackets), blocks 0, 1, 2, 100, 255\n", T, T / 64 - 1);
        int hist[16][4] = {};
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < T / 64; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
        for (int b : {0, 1, 2, 100, 255}) {
            printf("  block %3d (cu %2u se %u):", b, (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
            for (int w = 0; w < T / 64; ++w) printf(" %u[%u]", (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15);
            printf("\n");
        }
        printf("  histogram over 256 blocks, wave: simd0 simd1 simd2 simd3\n");
        for (int w = 0; w < T / 64; ++w) printf("   w%-2d: %3d %3d %3d %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    }
    return 0;
}
