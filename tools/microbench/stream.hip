// HBM streaming patterns of the shooting-MPC control tensor f32 [E][T][2][R] (tools/microbench; not part of libf1p.so).
//   A: one 256-thread workgroup per ego, lane reads rollouts r and r + 256 with dword loads (the k_kmpc_shoot_mixed pattern)
//   B: one 128-thread workgroup per ego, lane reads 4 consecutive rollouts with one dwordx4 load per (t, control)
//   C: linear float4 copy-less read of the whole buffer (grid-stride), the achievable-bandwidth reference
// build: hipcc --offload-arch=gfx950 -O3 -o stream tools/microbench/stream.hip ; run: ./stream [E]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void kA(const float* __restrict__ c, int T, int R, float* out) {
    const float* ce = c + (size_t)blockIdx.x * T * 2 * R;
    float s = 0.f;
    for (int r = threadIdx.x; r < R; r += 512) {
        const int r1 = r + 256 < R ? r + 256 : r;
        for (int t0 = 0; t0 < T; t0 += 10) {
            float a[10], d[10], a1[10], d1[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int t = t0 + j < T ? t0 + j : T - 1;
                a[j] = ce[((size_t)t * 2) * R + r]; d[j] = ce[((size_t)t * 2 + 1) * R + r];
                a1[j] = ce[((size_t)t * 2) * R + r1]; d1[j] = ce[((size_t)t * 2 + 1) * R + r1];
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) s += a[j] * d[j] + a1[j] * d1[j];
        }
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(128) void kB(const float* __restrict__ c, int T, int R, float* out) {
    const float* ce = c + (size_t)blockIdx.x * T * 2 * R;
    float s = 0.f;
    for (int r = threadIdx.x * 4; r < R; r += 512) {
        for (int t0 = 0; t0 < T; t0 += 10) {
            float4 a[10], d[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int t = t0 + j < T ? t0 + j : T - 1;
                a[j] = *reinterpret_cast<const float4*>(ce + ((size_t)t * 2) * R + r);
                d[j] = *reinterpret_cast<const float4*>(ce + ((size_t)t * 2 + 1) * R + r);
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) s += a[j].x * d[j].x + a[j].y * d[j].y + a[j].z * d[j].z + a[j].w * d[j].w;
        }
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void kC(const float4* __restrict__ c, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = c[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 8192, T = 30, R = 512;
    const size_t n = (size_t)E * T * 2 * R;
    float *c, *out;
    CHK(hipMalloc(&c, n * 4)); CHK(hipMalloc(&out, 4 * (size_t)E));
    CHK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int which = 0; which < 4; ++which) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            CHK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(kA, dim3(E), dim3(256), 0, 0, c, T, R, out);
            if (which == 1) hipLaunchKernelGGL(kB, dim3(E), dim3(128), 0, 0, c, T, R, out);
            if (which == 2) hipLaunchKernelGGL(kC, dim3(256 * 8), dim3(256), 0, 0, (const float4*)c, n / 4, out);
            if (which == 3) hipLaunchKernelGGL(kC, dim3(256 * 32), dim3(256), 0, 0, (const float4*)c, n / 4, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        const char* names[] = {"A dword, 256 thr/ego", "B dwordx4, 128 thr/ego", "C linear float4, 2048 wg", "C linear float4, 8192 wg"};
        printf("%-28s %8.4f ms  %7.0f GB/s\n", names[which], best, n * 4 / (best * 1e-3) / 1e9);
    }
    return 0;
}
