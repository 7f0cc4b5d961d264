// dispatch: time of an (almost) empty kernel against grid shape -- is launching 4096 workgroups of 256 threads dearer than 2048 of 512?
//   hipcc --offload-arch=gfx950 -O3 -o dispatch dispatch.hip && ./dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p, int lds_words) { extern __shared__ int s[]; if (lds_words < 0) { s[threadIdx.x] = 1; p[0] = s[0]; } }
int main() {
    int* d; hipMalloc(&d, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int shapes[][3] = {{4096, 256, 11776}, {2048, 512, 23552}, {4096, 256, 0}, {2048, 512, 0}, {1024, 1024, 47104}, {8192, 128, 5888}, {16384, 64, 2944}};
    for (auto& sh : shapes) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(sh[0]), dim3(sh[1]), sh[2], 0, d, 0);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(sh[0]), dim3(sh[1]), sh[2], 0, d, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("grid %5d x %4d threads, %5d B LDS: %.2f us per launch (back to back)\n", sh[0], sh[1], sh[2], ms * 1000.0f / 200.0f);
    }
    return 0;
}
