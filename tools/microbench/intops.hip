// Issue rate of the integer / conversion / transcendental VALU instructions the in-kernel control sampler is built from
// (tools/microbench; not part of libf1p.so).  8 independent chains per lane, 4 waves per SIMD, all CUs; the rate is printed
// relative to v_add_u32 so "quarter rate" instructions show up as 0.25.
// build: hipcc --offload-arch=gfx950 -O3 -o intops tools/microbench/intops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define KERNEL(name, body)                                                                                   \
    __global__ __launch_bounds__(256) void name(unsigned* out, int iters, unsigned a, unsigned b) {          \
        unsigned x[8];                                                                                       \
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;                                    \
        for (int it = 0; it < iters; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { body; }                                          \
        }                                                                                                    \
        unsigned s = 0;                                                                                      \
        for (int i = 0; i < 8; ++i) s ^= x[i];                                                               \
        if (s == 0x12345678u) out[0] = s;                                                                    \
    }

KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x[i])))
KERNEL(k_sad_u8, asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x[i])))
KERNEL(k_cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i])))
KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_sin_f32, asm volatile("v_sin_f32 %0, %0" : "+v"(x[i])))
KERNEL(k_log_f32, asm volatile("v_log_f32 %0, %0" : "+v"(x[i])))
KERNEL(k_sqrt_f32, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i])))
KERNEL(k_rcp_f32, asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])))
KERNEL(k_mad_u64_u32, { unsigned long long t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x[i]), "v"(a) : "vcc"); x[i] = (unsigned)(t >> 32) ^ (unsigned)t; })
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))

typedef void (*kern_t)(unsigned*, int, unsigned, unsigned);

int main() {
    unsigned* o;
    CHK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4, iters = 100000;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u32_u24", k_mad24},
        {"v_alignbit_b32", k_alignbit}, {"v_sad_u8", k_sad_u8}, {"v_bfe_u32", k_bfe}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_fma_f32", k_fma_f32}, {"v_sin_f32", k_sin_f32}, {"v_log_f32", k_log_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32},
        {"v_mad_u64_u32(+xor)", k_mad_u64_u32}, {"v_perm_b32", k_perm}, {"v_lshl_add_u32", k_lshl_add}, {"v_xad_u32", k_xad}, {"v_add3_u32", k_add3}};
    double base = 0;
    for (auto& kk : ks) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, o, iters, 0x9E3779B9u, 0x85EBCA6Bu);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double rate = (double)blocks * 256 * iters * 8 / (best * 1e-3) / 1e12;
        if (base == 0) base = rate;
        printf("%-22s %8.3f ms  %6.2f T lane-instr/s  rel %.2f\n", kk.name, best, rate, rate / base);
    }
    return 0;
}
