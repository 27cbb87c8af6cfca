// Does v_add_f64 pay for source operands that sit in the same VGPR banks?  (register number mod 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int VAR>
__global__ void k(long long* cyc, double* out) {
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        if (VAR == 0)   // sources 4 apart: same banks
            asm volatile("v_add_f64 v[40:41], v[20:21], v[24:25]\n v_add_f64 v[42:43], v[28:29], v[32:33]\n v_add_f64 v[44:45], v[20:21], v[32:33]\n v_add_f64 v[46:47], v[24:25], v[28:29]\n"
                         "v_add_f64 v[48:49], v[20:21], v[24:25]\n v_add_f64 v[50:51], v[28:29], v[32:33]\n v_add_f64 v[52:53], v[20:21], v[32:33]\n v_add_f64 v[54:55], v[24:25], v[28:29]\n"
                         ::: "v20","v21","v24","v25","v28","v29","v32","v33","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        if (VAR == 1)   // sources 2 apart: different banks
            asm volatile("v_add_f64 v[40:41], v[20:21], v[22:23]\n v_add_f64 v[42:43], v[28:29], v[30:31]\n v_add_f64 v[44:45], v[20:21], v[30:31]\n v_add_f64 v[46:47], v[22:23], v[28:29]\n"
                         "v_add_f64 v[48:49], v[20:21], v[22:23]\n v_add_f64 v[50:51], v[28:29], v[30:31]\n v_add_f64 v[52:53], v[20:21], v[30:31]\n v_add_f64 v[54:55], v[22:23], v[28:29]\n"
                         ::: "v20","v21","v22","v23","v28","v29","v30","v31","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        if (VAR == 2)   // sdwa shift
            asm volatile("v_lshlrev_b32_sdwa v40, v20, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v41, v20, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                         "v_lshlrev_b32_sdwa v42, v20, v25 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v43, v20, v25 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                         "v_lshlrev_b32_sdwa v44, v20, v26 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v45, v20, v26 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                         "v_lshlrev_b32_sdwa v46, v20, v27 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v47, v20, v27 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                         ::: "v20","v24","v25","v26","v27","v40","v41","v42","v43","v44","v45","v46","v47");
        if (VAR == 3)   // plain 32-bit ops: v_and / v_lshlrev
            asm volatile("v_and_b32 v40, v20, v24\n v_lshlrev_b32 v41, 4, v24\n v_and_b32 v42, v20, v25\n v_lshlrev_b32 v43, 4, v25\n v_and_b32 v44, v20, v26\n v_lshlrev_b32 v45, 4, v26\n v_and_b32 v46, v20, v27\n v_lshlrev_b32 v47, 4, v27\n"
                         ::: "v20","v24","v25","v26","v27","v40","v41","v42","v43","v44","v45","v46","v47");
        if (VAR == 4)   // v_bfe_u32
            asm volatile("v_bfe_u32 v40, v24, 8, 8\n v_bfe_u32 v41, v24, 16, 8\n v_bfe_u32 v42, v25, 8, 8\n v_bfe_u32 v43, v25, 16, 8\n v_bfe_u32 v44, v26, 8, 8\n v_bfe_u32 v45, v26, 16, 8\n v_bfe_u32 v46, v27, 8, 8\n v_bfe_u32 v47, v27, 16, 8\n"
                         ::: "v24","v25","v26","v27","v40","v41","v42","v43","v44","v45","v46","v47");
        if (VAR == 5)   // dependent chain of fp64 adds (latency)
            asm volatile("v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n"
                         "v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n v_add_f64 v[40:41], v[40:41], v[22:23]\n"
                         ::: "v22","v23","v40","v41");
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[2 * (threadIdx.x >> 6)] = t0, cyc[2 * (threadIdx.x >> 6) + 1] = t1;
    if (out) out[threadIdx.x] = 0.0;
}
template <int VAR>
void run(const char* name) {
    long long* cyc; hipMalloc(&cyc, 8 * 64);
    for (int threads : {256, 1024}) {
        hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(threads), 0, 0, cyc, (double*)nullptr);
        hipLaunchKernelGGL(k<VAR>, dim3(1), dim3(threads), 0, 0, cyc, (double*)nullptr);
        long long hw[64]; hipMemcpy(hw, cyc, 8 * 64, hipMemcpyDeviceToHost);
        long long lo = hw[0], hi = hw[1];
        for (int w = 0; w < threads / 64; ++w) { if (hw[2 * w] < lo) lo = hw[2 * w]; if (hw[2 * w + 1] > hi) hi = hw[2 * w + 1]; }
        printf("%-44s waves/SIMD=%d: %.2f cycles per instruction per SIMD\n", name, threads / 256, (hi - lo) / (N_IT * 8.0) / (threads / 256));
    }
}
int main() {
    run<0>("v_add_f64, sources in the same banks");
    run<1>("v_add_f64, sources in different banks");
    run<2>("v_lshlrev_b32_sdwa (byte select)");
    run<3>("v_and_b32 / v_lshlrev_b32");
    run<4>("v_bfe_u32");
    run<5>("v_add_f64 dependent chain");
    return 0;
}
