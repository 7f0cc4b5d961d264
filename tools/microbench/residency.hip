// How many workgroups of a trivial kernel are co-resident on the chip?  (tools/microbench; not part of libf1p.so)
// Every block bumps a counter, then polls it (bounded: 3 ms) and records the largest value it saw; blocks that start only after
// others have left see a count larger than the resident set, so the census is the MINIMUM over blocks of (largest value seen by
// a block that started in the first 50 us).
// build: hipcc --offload-arch=gfx950 -O3 -o residency tools/microbench/residency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NV>
__global__ void k_census(unsigned* counter, unsigned long long* start, unsigned* seen, unsigned* xcc, float* sink) {
    float keep[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) keep[i] = (float)threadIdx.x * 1e-3f + (float)i;
    const unsigned long long t0 = wall_clock64();
    unsigned mx = 0;
    if (threadIdx.x == 0) {
        start[blockIdx.x] = t0;
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id;
        atomicAdd(counter, 1u);
        while (wall_clock64() - t0 < 300000ull) {                // 3 ms at 100 MHz
            const unsigned v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mx = v > mx ? v : mx;
            __builtin_amdgcn_s_sleep(32);
        }
        seen[blockIdx.x] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(keep[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += keep[i];
    if (s == 1234.5f) sink[0] = s;
}

template <int NV>
void run(int threads, int blocks) {
    unsigned* c; unsigned long long* st; unsigned* seen; unsigned* xcc; float* sink;
    CHK(hipMalloc(&c, 4)); CHK(hipMalloc(&st, 8 * blocks)); CHK(hipMalloc(&seen, 4 * blocks)); CHK(hipMalloc(&xcc, 4 * blocks)); CHK(hipMalloc(&sink, 4));
    CHK(hipMemset(c, 0, 4));
    hipLaunchKernelGGL(k_census<NV>, dim3(blocks), dim3(threads), 0, 0, c, st, seen, xcc, sink);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(blocks); std::vector<unsigned> hv(blocks), hx(blocks);
    CHK(hipMemcpy(hs.data(), st, 8 * blocks, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hv.data(), seen, 4 * blocks, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hx.data(), xcc, 4 * blocks, hipMemcpyDeviceToHost));
    const unsigned long long s0 = *std::min_element(hs.begin(), hs.end());
    int first = 0; unsigned census = ~0u;
    int per_xcc[16] = {0};
    for (int i = 0; i < blocks; ++i) if (hs[i] - s0 < 5000ull) { ++first; census = std::min(census, hv[i]); per_xcc[hx[i] & 15]++; }
    printf("threads %4d  regs>=%3d  grid %5d: %5d blocks started within 50 us, census %5u  (%.2f blocks per CU, %.1f waves per SIMD)  per XCC:", threads, NV, blocks, first,
           census, first / 256.0, first * (threads / 64.0) / 1024.0);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    hipFree(c); hipFree(st); hipFree(seen); hipFree(xcc); hipFree(sink);
}

int main() {
    for (int blocks : {256, 512, 1024, 2048, 4096}) run<8>(256, blocks);
    for (int blocks : {1024, 4096, 8192, 16384}) run<8>(64, blocks);
    for (int blocks : {256, 512, 1024}) run<8>(1024, blocks);
    for (int blocks : {2048, 4096}) run<56>(256, blocks);
    for (int blocks : {2048, 4096}) run<100>(256, blocks);
    return 0;
}
