// issue_rate.hip - ONE table of vector-instruction issue costs on gfx950, every class the scan loop and the chain step use,
// each as 8 independent streams (no dependent chain shorter than 8 instructions), at 1 / 2 / 4 waves per SIMD on one CU,
// with v_add_f32 and v_add_f64 as controls.  Cycles are shader cycles (s_memtime; MI355X_MICROARCH.md: tick = shader
// cycle), the clock itself is reported from s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define N_IT 4096

#define S8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define OPS8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define OPD8 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)

enum { ADD_F32, ADD_F64, FMA_F64, AND_B32, LSHL_B32, BFE_U32, AND_OR_B32, PERM_B32, LSHL_SDWA, AND_SDWA, CVT_F64_U32, ADD_U32, MOV_B32,
       LSHL_ADD_U32, MAD_U32_U24, DS_READ_B128, DS_READ_B64, MUL_F64, CNDMASK, ADD_F64_DEP,
       LSHR_B32, AND_LIT, AND_SGPR, CNDMASK_E64, MOV_DPP, CMP_F64, MAX_F64, MED3_I32, XOR_B32, MAD_U64_U32, FMA_F32, MUL_LO_U32, CMP_CND_VCC, CND_VCC_SET, CMP_U32_VCC, READLANE, CMP_2CND_VCC, CMP_2CND_SGPR, CMP_4CND_VCC, CND_VCC_NOP, N_OPS };
static const char* NAMES[N_OPS] = {"v_add_f32", "v_add_f64", "v_fma_f64", "v_and_b32", "v_lshlrev_b32", "v_bfe_u32", "v_and_or_b32", "v_perm_b32",
                                   "v_lshlrev_b32_sdwa WORD_1", "v_and_b32_sdwa WORD_0", "v_cvt_f64_u32", "v_add_u32", "v_mov_b32",
                                   "v_lshl_add_u32", "v_mad_u32_u24", "ds_read_b128 (conflict-free)", "ds_read_b64 (conflict-free)", "v_mul_f64",
                                   "v_cndmask_b32 vcc", "v_add_f64 one dependent chain", "v_lshrrev_b32", "v_and_b32 literal", "v_and_b32 sgpr mask",
                                   "v_cndmask_b32 e64 sgpr mask", "v_mov_b32_dpp row_shr:1", "v_cmp_lt_f64 -> sgpr pair", "v_max_f64", "v_med3_i32", "v_xor_b32",
                                   "v_mad_u64_u32", "v_fma_f32", "v_mul_lo_u32", "v_cmp_lt_u32 vcc + v_cndmask vcc (per PAIR)", "v_cndmask_b32 vcc (vcc set before the loop)",
                                   "v_cmp_lt_u32 -> vcc", "v_readlane_b32", "v_cmp vcc + 2 v_cndmask vcc (per TRIPLE)", "v_cmp s[20:21] + 2 v_cndmask e64 (per TRIPLE)",
                                   "v_cmp vcc + 4 v_cndmask vcc (per FIVE)", "v_cndmask vcc + v_add_u32 alternating (per PAIR)"};

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, double seed) {
    __shared__ uint4 lds[1024];
    lds[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    unsigned a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const unsigned b = (unsigned)seed + 3u, c = 0x00ff00ffu;
    const double db = seed * 0.5, dc = 1.0;
    const unsigned lane_off = (threadIdx.x & 63) * 16;     // ds reads: lane l -> entry l (conflict-free)
    uint4 r0, r1, r2, r3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < N_IT; ++i) {
        if (OP == ADD_F32) {
#define F(n) "v_add_f32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b));
#undef F
        } else if (OP == ADD_F64) {
#define F(n) "v_add_f64 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPD8 : "v"(db));
#undef F
        } else if (OP == MUL_F64) {
#define F(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPD8 : "v"(dc));
#undef F
        } else if (OP == FMA_F64) {
#define F(n) "v_fma_f64 %" #n ", %" #n ", %9, %8\n"
            asm volatile(S8(F) : OPD8 : "v"(db), "v"(dc));
#undef F
        } else if (OP == ADD_F64_DEP) {
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                         "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(d0) : "v"(db));
        } else if (OP == AND_B32) {
#define F(n) "v_and_b32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(c));
#undef F
        } else if (OP == LSHL_B32) {
#define F(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
            asm volatile(S8(F) : OPS8);
#undef F
        } else if (OP == BFE_U32) {
#define F(n) "v_bfe_u32 %" #n ", %" #n ", 3, 13\n"
            asm volatile(S8(F) : OPS8);
#undef F
        } else if (OP == AND_OR_B32) {
#define F(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(c), "v"(b));
#undef F
        } else if (OP == PERM_B32) {
#define F(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(b), "v"(c));
#undef F
        } else if (OP == LSHL_SDWA) {
#define F(n) "v_lshlrev_b32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
            asm volatile(S8(F) : OPS8 : "v"(4u));
#undef F
        } else if (OP == AND_SDWA) {
#define F(n) "v_and_b32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            asm volatile(S8(F) : OPS8 : "v"(c));
#undef F
        } else if (OP == CVT_F64_U32) {
#define F(n) "v_cvt_f64_u32 %" #n ", %8\n"
            asm volatile(S8(F) : OPD8 : "v"(b));
#undef F
        } else if (OP == ADD_U32) {
#define F(n) "v_add_u32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b));
#undef F
        } else if (OP == MOV_B32) {
#define F(n) "v_mov_b32 %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b));
#undef F
        } else if (OP == LSHL_ADD_U32) {
#define F(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b));
#undef F
        } else if (OP == MAD_U32_U24) {
#define F(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(b), "v"(c));
#undef F
        } else if (OP == CNDMASK) {
#define F(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
            asm volatile(S8(F) : OPS8 : "v"(b) : "vcc");
#undef F
        } else if (OP == LSHR_B32) {
#define F(n) "v_lshrrev_b32 %" #n ", 16, %" #n "\n"
            asm volatile(S8(F) : OPS8);
#undef F
        } else if (OP == AND_LIT) {
#define F(n) "v_and_b32 %" #n ", 0xfff0, %" #n "\n"
            asm volatile(S8(F) : OPS8);
#undef F
        } else if (OP == AND_SGPR) {
#define F(n) "v_and_b32 %" #n ", %8, %" #n "\n"
            asm volatile(S8(F) : OPS8 : "s"(c));
#undef F
        } else if (OP == CNDMASK_E64) {
#define F(n) "v_cndmask_b32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(b), "s"(0x5555555555555555ull));
#undef F
        } else if (OP == MOV_DPP) {
#define F(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
            asm volatile(S8(F) : OPS8);
#undef F
        } else if (OP == CMP_F64) {
            asm volatile("v_cmp_lt_f64 s[20:21], %0, %8\n v_cmp_lt_f64 s[22:23], %1, %8\n v_cmp_lt_f64 s[24:25], %2, %8\n v_cmp_lt_f64 s[26:27], %3, %8\n"
                         "v_cmp_lt_f64 s[28:29], %4, %8\n v_cmp_lt_f64 s[30:31], %5, %8\n v_cmp_lt_f64 s[32:33], %6, %8\n v_cmp_lt_f64 s[34:35], %7, %8\n"
                         : OPD8 : "v"(db) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if (OP == MAX_F64) {
#define F(n) "v_max_f64 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPD8 : "v"(db));
#undef F
        } else if (OP == MED3_I32) {
#define F(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(b), "v"(c));
#undef F
        } else if (OP == XOR_B32) {
#define F(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(c));
#undef F
        } else if (OP == MAD_U64_U32) {
#define F(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n"
            asm volatile(S8(F) : OPD8 : "v"(b), "v"(c) : "vcc");
#undef F
        } else if (OP == FMA_F32) {
#define F(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
            asm volatile(S8(F) : OPS8 : "v"(b), "v"(c));
#undef F
        } else if (OP == MUL_LO_U32) {
#define F(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b));
#undef F
        } else if (OP == CMP_CND_VCC) {
#define F(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
            asm volatile(S8(F) : OPS8 : "v"(b) : "vcc");
#undef F
        } else if (OP == CND_VCC_SET) {
#define F(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
            asm volatile("s_mov_b64 vcc, 0x5555\n" S8(F) : OPS8 : "v"(b) : "vcc");
#undef F
        } else if (OP == CMP_U32_VCC) {
#define F(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b) : "vcc");
#undef F
        } else if (OP == READLANE) {
            asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                         "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3\n"
                         : OPS8 : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (OP == CMP_2CND_VCC) {
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : OPS8 : "v"(b) : "vcc");
        } else if (OP == CMP_2CND_SGPR) {
            asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n"
                         "v_cmp_lt_u32 s[22:23], %2, %8\n v_cndmask_b32 %2, %2, %8, s[22:23]\n v_cndmask_b32 %3, %3, %8, s[22:23]\n"
                         "v_cmp_lt_u32 s[24:25], %4, %8\n v_cndmask_b32 %4, %4, %8, s[24:25]\n v_cndmask_b32 %5, %5, %8, s[24:25]\n"
                         "v_cmp_lt_u32 s[26:27], %6, %8\n v_cndmask_b32 %6, %6, %8, s[26:27]\n v_cndmask_b32 %7, %7, %8, s[26:27]\n"
                         "v_cmp_lt_u32 s[20:21], %0, %8\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n"
                         "v_cmp_lt_u32 s[22:23], %2, %8\n v_cndmask_b32 %2, %2, %8, s[22:23]\n v_cndmask_b32 %3, %3, %8, s[22:23]\n"
                         "v_cmp_lt_u32 s[24:25], %4, %8\n v_cndmask_b32 %4, %4, %8, s[24:25]\n v_cndmask_b32 %5, %5, %8, s[24:25]\n"
                         "v_cmp_lt_u32 s[26:27], %6, %8\n v_cndmask_b32 %6, %6, %8, s[26:27]\n v_cndmask_b32 %7, %7, %8, s[26:27]\n"
                         : OPS8 : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (OP == CMP_4CND_VCC) {
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : OPS8 : "v"(b) : "vcc");
        } else if (OP == CND_VCC_NOP) {
#define F(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_add_u32 %" #n ", %" #n ", %8\n"
            asm volatile(S8(F) : OPS8 : "v"(b) : "vcc");
#undef F
        } else if (OP == DS_READ_B128) {
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                         "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                         "s_waitcnt lgkmcnt(0)\n" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(lane_off) : "memory");
        } else if (OP == DS_READ_B64) {
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                         "ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:5120\n ds_read_b64 %2, %4 offset:6144\n ds_read_b64 %3, %4 offset:7168\n"
                         "s_waitcnt lgkmcnt(0)\n" : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(lane_off / 2) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) out[4 * wave] = t0, out[4 * wave + 1] = t1, out[4 * wave + 2] = w0, out[4 * wave + 3] = w1;
    // keep every stream alive
    if (seed == -1.0) out[0] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned long long)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + r0.x + r1.x + r2.x + r3.x;
}

template <int OP>
void run(int waves_per_simd, int blocks, FILE* f) {
    const int threads = 256 * waves_per_simd;
    unsigned long long* out;
    const int n_waves = blocks * threads / 64;
    hipMalloc(&out, n_waves * 32);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0);
    unsigned long long* h = new unsigned long long[n_waves * 4];
    hipMemcpy(h, out, n_waves * 32, hipMemcpyDeviceToHost);
    // per block: the span from its first wave's start to its last wave's end; the median block
    double worst_cyc = 0, worst_ns = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned long long lo = ~0ull, hi = 0, wlo = ~0ull, whi = 0;
        for (int w = 0; w < threads / 64; ++w) {
            const unsigned long long* p = h + 4 * (b * (threads / 64) + w);
            if (p[0] < lo) lo = p[0];
            if (p[1] > hi) hi = p[1];
            if (p[2] < wlo) wlo = p[2];
            if (p[3] > whi) whi = p[3];
        }
        if ((double)(hi - lo) > worst_cyc) worst_cyc = (double)(hi - lo), worst_ns = (double)(whi - wlo) * 10.0;
    }
    const double n_inst = (double)N_IT * 8.0;
    fprintf(f, "%-30s waves/SIMD=%d CUs=%3d: %6.2f cycles per wave-instruction per SIMD  (%.2f per wave; clock %.2f GHz)\n", NAMES[OP],
            waves_per_simd, blocks, worst_cyc / n_inst / waves_per_simd, worst_cyc / n_inst, worst_cyc / worst_ns);
    delete[] h;
    hipFree(out);
}

template <int OP>
void all(FILE* f) {
    for (int w : {1, 2, 4}) run<OP>(w, 1, f);
    run<OP>(4, 256, f);
}

int main() {
    FILE* f = stdout;
    all<ADD_F32>(f); all<ADD_F64>(f); all<MUL_F64>(f); all<FMA_F64>(f); all<ADD_F64_DEP>(f); all<CVT_F64_U32>(f);
    all<AND_B32>(f); all<LSHL_B32>(f); all<BFE_U32>(f); all<AND_OR_B32>(f); all<PERM_B32>(f); all<LSHL_SDWA>(f); all<AND_SDWA>(f);
    all<ADD_U32>(f); all<MOV_B32>(f); all<LSHL_ADD_U32>(f); all<MAD_U32_U24>(f); all<CNDMASK>(f);
    all<LSHR_B32>(f); all<AND_LIT>(f); all<AND_SGPR>(f); all<XOR_B32>(f); all<MED3_I32>(f); all<MUL_LO_U32>(f); all<MAD_U64_U32>(f); all<FMA_F32>(f);
    all<CMP_2CND_VCC>(f); all<CMP_2CND_SGPR>(f); all<CMP_4CND_VCC>(f); all<CND_VCC_NOP>(f);
    all<CMP_CND_VCC>(f); all<CND_VCC_SET>(f); all<CMP_U32_VCC>(f); all<READLANE>(f);
    all<CNDMASK_E64>(f); all<MOV_DPP>(f); all<CMP_F64>(f); all<MAX_F64>(f);
    all<DS_READ_B128>(f); all<DS_READ_B64>(f);
    return 0;
}
