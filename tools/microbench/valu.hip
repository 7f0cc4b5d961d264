// Sustained fp64 / packed-f32 VALU issue rate of the chip (tools/microbench; not part of libf1p.so): the denominators the
// VALU-bound kernels are held against in DESIGN.md.  8 independent FMA chains per lane, 4 waves per SIMD, all CUs.
// build: hipcc --offload-arch=gfx950 -O3 -o valu tools/microbench/valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.6789) out[0] = s;
}

__global__ __launch_bounds__(256) void k_pk32(float* out, int iters, float a, float b) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 1e-6f + i; x[i].y = x[i].x + 0.5f; }
    const f2 va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], va, vb);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    double* o64; float* o32;
    CHK(hipMalloc(&o64, 8)); CHK(hipMalloc(&o32, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4, iters = 200000;      // ~14 ms per launch: long enough for the clocks to settle      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.999999, 1e-7);
            else hipLaunchKernelGGL(k_pk32, dim3(blocks), dim3(256), 0, 0, o32, iters, 0.999999f, 1e-7f);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double lane_instr = (double)blocks * 256 * iters * 8;
        if (which == 0) printf("v_fma_f64     : %8.3f ms  %6.2f T lane-instr/s  = %6.1f TFLOP/s\n", best, lane_instr / (best * 1e-3) / 1e12, 2 * lane_instr / (best * 1e-3) / 1e12);
        else printf("v_pk_fma_f32  : %8.3f ms  %6.2f T lane-instr/s  = %6.1f TFLOP/s (2 floats per lane-instr)\n", best, lane_instr / (best * 1e-3) / 1e12, 4 * lane_instr / (best * 1e-3) / 1e12);
    }
    return 0;
}
