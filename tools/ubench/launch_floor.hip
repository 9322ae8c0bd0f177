// Floor of a dependent kernel launch on this stack: N empty (or tiny) kernels back to back on one stream, us per kernel;
// and the same with one HIP event recorded between every pair (what the per-launch profile of nww_set_profiling sees).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 1000) p[0] = 1.0f; }
__global__ void chain_kernel(float* p, int n) {      // n dependent global round trips by one thread
    if (threadIdx.x == 0 && blockIdx.x == 0) { float v = p[0]; for (int i = 0; i < n; ++i) { p[(int)v & 1023] = v + 1.0f; __threadfence(); v = p[((int)v + 1) & 1023]; } p[0] = v; }
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int grid : {1, 64, 256}) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int N = 2000;
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d);
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        // one call = 4 kernels + sync, as the B = 1 forward
        t0 = std::chrono::steady_clock::now();
        for (int c = 0; c < 500; ++c) { for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d); hipStreamSynchronize(s); }
        const double call = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
        printf("grid %3d: %.2f us per empty kernel back to back; 4 empty kernels + stream sync: %.1f us per call\n", grid, us, call);
    }
    return 0;
}
