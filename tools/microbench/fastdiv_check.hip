// fastdiv_check: the split fp64 division used by k_stmpc_refine_tp (denominator-only Newton refinement of v_rcp_f64 + a three-operation
// numerator half) against the compiler's own division, bit for bit, over random operands of the guarded exponent range [2^-300, 2^300]
// (both signs), operands clustered around powers of two, and quotients near rounding boundaries.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fastdiv_check fastdiv_check.hip && ./fastdiv_check [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ double rcp_refined(double den) {
    const double r0 = __builtin_amdgcn_rcp(den);
    const double f0 = __builtin_fma(-den, r0, 1.0);
    const double r1 = __builtin_fma(r0, f0, r0);
    const double f2 = __builtin_fma(-den, r1, 1.0);
    return __builtin_fma(r1, f2, r1);
}
__device__ __forceinline__ double div_by(double num, double den, double r) {
    const double m = num * r;
    const double e = __builtin_fma(-den, m, num);
    return __builtin_fma(e, r, m);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ __forceinline__ double make(uint64_t bits, int mode) {
    // sign | exponent in [723, 1323] | mantissa; mode 1: mantissa near 0 or all ones (powers of two), mode 2: short mantissas (exact quotients)
    uint64_t man = bits & 0xfffffffffffffull;
    if (mode == 1) man = (bits >> 60 & 1) ? (0xfffffffffffffull - (man & 0xff)) : (man & 0xff);
    if (mode == 2) man &= 0xfffff00000000ull;
    const uint64_t ex = 723 + (bits >> 52) % 601;
    return __longlong_as_double((long long)((bits & 0x8000000000000000ull) | (ex << 52) | man));
}
__global__ void k(uint64_t seed, unsigned long long* bad, double* ex_num, double* ex_den) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (int it = 0; it < 256; ++it) {
        const uint64_t a = mix(seed + i * 256 + it), b = mix(a ^ 0x9e3779b97f4a7c15ull);
        const int mode = (int)((a >> 61) % 3);
        const double num = make(a, mode), den = make(b, (int)((b >> 61) % 3));
        const double q0 = num / den, q1 = div_by(num, den, rcp_refined(den));
        if (__double_as_longlong(q0) != __double_as_longlong(q1)) { if (!nb) { ex_num[0] = num; ex_den[0] = den; } ++nb; }
    }
    if (nb) atomicAdd(bad, nb);
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 64;
    unsigned long long* bad; double *en, *ed;
    hipMalloc(&bad, 8); hipMalloc(&en, 8); hipMalloc(&ed, 8); hipMemset(bad, 0, 8);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, 0x1234567ull + (uint64_t)r * 0x100000000ull, bad, en, ed);
    unsigned long long h = 0; double hn = 0, hd = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hn, en, 8, hipMemcpyDeviceToHost); hipMemcpy(&hd, ed, 8, hipMemcpyDeviceToHost);
    printf("pairs %.3e  mismatches %llu", (double)rounds * 65536.0 * 256.0 * 256.0, h);
    if (h) printf("  e.g. %.17g / %.17g", hn, hd);
    printf("\n");
    return h ? 1 : 0;
}
