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

// D: linear float4 fill (write bandwidth reference).  E: the materialised-lattice store pattern: a wave writes, for 64 candidates,
// 128-byte chunks (8 lanes x 16 B) at a stride of S*32 bytes between candidates, 4 stations at a time.
__global__ __launch_bounds__(256) void kD(float4* __restrict__ c, size_t n4) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) c[i] = v;
}
__global__ __launch_bounds__(256) void kE(float4* __restrict__ c, int C, int S) {
    // block = one ego (C candidates); wave w covers candidates 64 w .. 64 w + 63; rows of 32 B = 2 float4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* ego = c + (size_t)blockIdx.x * C * S * 2;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int cb = wave * 64; cb < C; cb += 256)
        for (int i0 = 0; i0 < S; i0 += 4) {
            const int n_units = 2 * (S - i0 < 4 ? S - i0 : 4);
            const int unit = lane % 8, sub = lane / 8;
            for (int g = 0; g < 8; ++g) {
                const int cand = cb + g * 8 + sub;
                if (cand < C && unit < n_units) ego[(size_t)cand * S * 2 + (size_t)i0 * 2 + unit] = v;
            }
        }
}

typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kDnt(v4f* __restrict__ c, size_t n4) {
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &c[i]);
}
__global__ __launch_bounds__(256) void kEnt(v4f* __restrict__ c, int C, int S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4f* ego = c + (size_t)blockIdx.x * C * S * 2;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (int cb = wave * 64; cb < C; cb += 256)
        for (int i0 = 0; i0 < S; i0 += 4) {
            const int n_units = 2 * (S - i0 < 4 ? S - i0 : 4);
            const int unit = lane % 8, sub = lane / 8;
            for (int g = 0; g < 8; ++g) {
                const int cand = cb + g * 8 + sub;
                if (cand < C && unit < n_units) __builtin_nontemporal_store(v, &ego[(size_t)cand * S * 2 + (size_t)i0 * 2 + unit]);
            }
        }
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
    {   // write patterns on a 4096 x 256 x 50 x 32 B = 1.68 GB tensor
        const int Eg = 4096, C = 256, S = 50;
        const size_t n4 = (size_t)Eg * C * S * 2;
        float4* w;
        CHK(hipMalloc(&w, n4 * 16));
        for (int which = 0; which < 4; ++which) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                CHK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(kD, dim3(256 * 16), dim3(256), 0, 0, w, n4);
                if (which == 1) hipLaunchKernelGGL(kE, dim3(Eg), dim3(256), 0, 0, w, C, S);
                if (which == 2) hipLaunchKernelGGL(kDnt, dim3(256 * 16), dim3(256), 0, 0, (v4f*)w, n4);
                if (which == 3) hipLaunchKernelGGL(kEnt, dim3(Eg), dim3(256), 0, 0, (v4f*)w, C, S);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            const char* names[] = {"D linear float4 fill", "E 128-B chunks, stride S*32 B", "D nontemporal", "E nontemporal"};
            printf("%-28s %8.4f ms  %7.0f GB/s\n", names[which], best, n4 * 16 / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
