// EXHAUSTIVE error of the f32 hardware primitives the filter kernels rest on (tools/microbench; not part of libf1p.so):
//   v_sin_f32 / v_cos_f32 (argument in revolutions) over EVERY f32 with |x| <= 8 revolutions (50 rad: the filter stops at 20),
//   v_rcp_f32 and v_sqrt_f32 / v_rsq_f32 over every positive normal f32,
// against the correctly rounded fp64 device-library results.  One thread per bit pattern: 2^31 patterns per function, seconds on
// the GPU.  The maxima are the epsilon_trig / epsilon_rcp of DESIGN.md's error budget -- by exhaustion, not by sampling.
// build: hipcc --offload-arch=gfx950 -O3 -o hw_f32_errors tools/microbench/hw_f32_errors.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Acc { double max_abs, max_rel; unsigned arg_abs, arg_rel; unsigned long long n; };

__device__ void atomic_max_d(double* addr, double v, unsigned* arg, unsigned bits) {
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (__longlong_as_double((long long)old) < v) {
        const unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
        if (prev == old) { *arg = bits; break; }
        old = prev;
    }
}

// which: 0 sin, 1 cos, 2 rcp, 3 sqrt, 4 rsq
__global__ void k_scan(int which, unsigned lo, unsigned hi, Acc* acc) {
    const unsigned long long span = (unsigned long long)hi - lo + 1;
    double mabs = 0.0, mrel = 0.0; unsigned aabs = 0, arel = 0; unsigned long long n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (unsigned long long)gridDim.x * blockDim.x) {
        for (int sign = 0; sign < (which < 2 ? 2 : 1); ++sign) {
            const unsigned bits = (unsigned)(lo + i) | (sign ? 0x80000000u : 0u);
            const float x = __uint_as_float(bits);
            float got; double want;
            if (which == 0) { got = __builtin_amdgcn_sinf(x); want = sin(6.283185307179586476925 * (double)x); }
            else if (which == 1) { got = __builtin_amdgcn_cosf(x); want = cos(6.283185307179586476925 * (double)x); }
            else if (which == 2) { got = __builtin_amdgcn_rcpf(x); want = 1.0 / (double)x; }
            else if (which == 3) { got = __builtin_sqrtf(x); want = sqrt((double)x); }
            else { got = __builtin_amdgcn_rsqf(x); want = 1.0 / sqrt((double)x); }
            const double e = fabs((double)got - want);
            const double r = want != 0.0 ? e / fabs(want) : 0.0;
            if (e > mabs) { mabs = e; aabs = bits; }
            if (r > mrel && fabs(want) > 1e-30) { mrel = r; arel = bits; }
            ++n;
        }
    }
    atomic_max_d(&acc->max_abs, mabs, &acc->arg_abs, aabs);
    atomic_max_d(&acc->max_rel, mrel, &acc->arg_rel, arel);
    atomicAdd(&acc->n, n);
}

int main() {
    Acc* d; CHK(hipMalloc(&d, sizeof(Acc)));
    const char* names[5] = {"v_sin_f32 (revolutions)", "v_cos_f32 (revolutions)", "v_rcp_f32", "v_sqrt_f32 (as compiled: __builtin_sqrtf)", "v_rsq_f32"};
    auto f2u = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
    printf("exhaustive error of the f32 primitives against fp64 (every bit pattern in the stated range)\n");
    for (int which = 0; which < 5; ++which) {
        // trig: |x| in [0, 8] revolutions incl. subnormals and 0; others: every positive normal float
        const unsigned lo = which < 2 ? 0u : 0x00800000u, hi = which < 2 ? f2u(8.0f) : 0x7f7fffffu;
        CHK(hipMemset(d, 0, sizeof(Acc)));
        hipLaunchKernelGGL(k_scan, dim3(256 * 32), dim3(256), 0, 0, which, lo, hi, d);
        CHK(hipDeviceSynchronize());
        Acc h; CHK(hipMemcpy(&h, d, sizeof(Acc), hipMemcpyDeviceToHost));
        float xa, xr; memcpy(&xa, &h.arg_abs, 4); memcpy(&xr, &h.arg_rel, 4);
        printf("%-46s %11llu arguments: max abs error %.4e (at x = %.9g), max rel error %.4e = %.2f ulp(1) (at x = %.9g)\n", names[which], h.n, h.max_abs, (double)xa,
               h.max_rel, h.max_rel / 5.9604644775390625e-08, (double)xr);
    }
    return 0;
}
