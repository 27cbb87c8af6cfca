// log_check: lr_log (csrc/lr_math.h) against the host's long-double logarithm, and its cost beside the device
// library's log.  hipcc --offload-arch=gfx950 -O3 -I literate_amd/csrc scratch/ubench/log_check.hip -o log_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "lr_math.h"

__global__ void eval_kernel(const double* x, double* mine, double* lib, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mine[i] = lr_log(x[i]), lib[i] = log(x[i]);
}

template <int WHICH>
__global__ void chain_kernel(double* out, double x0, int reps) {
    double x = x0 + threadIdx.x * 1e-3, acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        const double l = WHICH ? lr_log(x) : log(x);
        acc += l;
        x = x * 1.0000001 + l * 1e-9;       // dependent chain: one wave, nothing to overlap with
    }
    out[threadIdx.x] = acc;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17; return rng_state; }

int main() {
    const int n = 1 << 26;
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) {
        const uint64_t r = rnd();
        double v;
        switch (i & 7) {
            case 0: { uint64_t b = r & 0x7fffffffffffffffull; memcpy(&v, &b, 8); if (!(v == v) || std::isinf(v)) v = 1.5; } break;   // any exponent
            case 1: v = (double)(r >> 11) / 9007199254740992.0; break;                        // a uniform draw
            case 2: v = 1.0 + ((double)(r >> 11) / 9007199254740992.0 - 0.5) * 1e-3; break;   // around 1
            case 3: v = 0.70710678118654752 + ((double)(r >> 11) / 9007199254740992.0 - 0.5) * 1e-6; break;   // the split point
            case 4: v = ((double)(r >> 11) / 9007199254740992.0) * 4.0; break;                // rates
            case 5: v = 1.4142135623730951 + ((double)(r >> 11) / 9007199254740992.0 - 0.5) * 1e-6; break;
            case 6: v = ldexp(1.0 + (double)(r >> 12) / 4503599627370496.0, -1074 + (int)(r & 63)); break;   // subnormals and up
            default: v = (double)((r >> 40) + 1); break;                                      // integers (log k)
        }
        x[i] = v;
    }
    x[0] = 0.0, x[1] = -1.0, x[2] = INFINITY, x[3] = NAN, x[4] = 1.0, x[5] = 4.9406564584124654e-324, x[6] = 1.7976931348623157e308, x[7] = -0.0;
    double *dx, *dm, *dl;
    hipMalloc(&dx, n * 8ll), hipMalloc(&dm, n * 8ll), hipMalloc(&dl, n * 8ll);
    hipMemcpy(dx, x.data(), n * 8ll, hipMemcpyHostToDevice);
    eval_kernel<<<n / 256, 256>>>(dx, dm, dl, n);
    std::vector<double> mine(n), lib(n);
    hipMemcpy(mine.data(), dm, n * 8ll, hipMemcpyDeviceToHost), hipMemcpy(lib.data(), dl, n * 8ll, hipMemcpyDeviceToHost);
    printf("specials: log(0)=%g log(-1)=%g log(inf)=%g log(nan)=%g log(1)=%g log(denorm_min)=%.17g (%.17g) log(max)=%.17g (%.17g) log(-0)=%g\n",
           mine[0], mine[1], mine[2], mine[3], mine[4], mine[5], (double)logl(x[5]), mine[6], (double)logl(x[6]), mine[7]);
    double worst_mine = 0, worst_lib = 0; int wi = 0; long long over_half = 0;
    for (int i = 8; i < n; ++i) {
        const long double ref = logl((long double)x[i]);
        const double rd = (double)ref;
        const double ulp = std::fabs(std::nextafter(rd, INFINITY) - rd);
        const double em = (double)(fabsl((long double)mine[i] - ref) / ulp), el = (double)(fabsl((long double)lib[i] - ref) / ulp);
        if (em > worst_mine) worst_mine = em, wi = i;
        if (el > worst_lib) worst_lib = el;
        over_half += em > 0.5;
    }
    printf("n=%d  lr_log: worst %.3f ulp (x=%.17g), rounded differently from the exact value in %.3f %% of cases;  device library log: worst %.3f ulp\n",
           n, worst_mine, x[wi], 100.0 * over_half / n, worst_lib);
    double* out; hipMalloc(&out, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0), hipEventCreate(&e1);
    const int reps = 200000;
    for (int which = 0; which < 2; ++which) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (which) chain_kernel<1><<<1, 64>>>(out, 0.3, reps); else chain_kernel<0><<<1, 64>>>(out, 0.3, reps);
            hipEventRecord(e1), hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f ns per dependent call on one wave\n", which ? "lr_log" : "device library log", ms * 1e6 / reps);
    }
    return 0;
}
