// lr_math.h - the natural logarithm of the chain step.
//
// A chain step is one wave issuing one instruction every ~5 cycles (Kernels): its time is its instruction
// count.  The device library's fp64 log carries every intermediate as a head / tail pair to stay under one ulp and
// comes to ~100 instructions; a step takes four to six logarithms (the accept draw, the rates of the proposal, the
// priors, the split / merge moves).  lr_log is the plain-precision form, ~35 instructions:
//
//   x = 2^k m,  m in [sqrt(1/2), sqrt 2),  f = m - 1 (exact),  s = f / (2 + f),  z = s^2
//   log m = 2 s + s z P(z) = f - s (f - z P(z))          (2 s = f - s f: the rounding of s only meets f^2 / 2)
//   P(z) = 2/3 + 2/5 z + 2/7 z^2 + ... : seven coefficients interpolating it at the Chebyshev nodes of
//          [0, (3 - 2 sqrt 2)^2] (scratch/ubench/log_fit.py: |error| < 4e-16, i.e. 5e-18 of log m)
//   log x = k ln2_hi + (f + (k ln2_lo - s (f - z P)))    (ln2_hi has 32 significant bits: k ln2_hi is exact)
//
// Measured against the host's long-double logarithm over 2^26 arguments (scratch/ubench/log_check.hip): at most
// 1.1 ulp (the device library: 0.64), 0 -> -inf, negative / NaN -> NaN, +inf -> +inf, subnormals as the hardware's frexp takes them.
// The parity bar of the path is 1e-9 relative (tests/); every engine calls this one function, so they stay
// bit-identical with each other.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double lr_log(double x) {
    const double SQRT_HALF = 0x1.6a09e667f3bcdp-1;
    const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    const double P0 = 0x1.5555555555558p-1, P1 = 0x1.9999999995273p-2, P2 = 0x1.2492492dfd922p-2, P3 = 0x1.c71c62d5d7104p-3,
                 P4 = 0x1.7462b91f3eb6bp-3, P5 = 0x1.39fdcc7eb44fcp-3, P6 = 0x1.2b5f6d341e1a9p-3;
    double m = __builtin_amdgcn_frexp_mant(x);                 // [1/2, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < SQRT_HALF;
    m = __builtin_amdgcn_ldexp(m, low ? 1 : 0);                // [sqrt 1/2, sqrt 2)
    k -= low ? 1 : 0;
    const double f = m - 1.0, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double s = f * r;
    const double z = s * s;
    double p = __builtin_fma(P6, z, P5);
    p = __builtin_fma(p, z, P4);
    p = __builtin_fma(p, z, P3);
    p = __builtin_fma(p, z, P2);
    p = __builtin_fma(p, z, P1);
    p = __builtin_fma(p, z, P0);
    const double kd = (double)k;
    const double t = __builtin_fma(-z, p, f);                  // f - z P(z)
    const double tail = __builtin_fma(kd, LN2_LO, -(s * t));
    double res = __builtin_fma(kd, LN2_HI, f + tail);
    if (!(x > 0.0 && x < __builtin_inf())) res = (x == 0.0) ? -__builtin_inf() : (x > 0.0 ? x : __builtin_nan(""));
    return res;
}
