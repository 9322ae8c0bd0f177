// Micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate as a function of how many INDEPENDENT accumulators a wave
// rotates through (1 = every MFMA depends on the previous one) and of the waves per SIMD.  Also v_mfma_f32_16x16x4_f32.
// build: hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k_bf16(float* out, int iters, uint32_t seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 av = make_uint4(seed + threadIdx.x, seed * 3, seed * 5, seed * 7);
    bf16x8 A = __builtin_bit_cast(bf16x8, av), B = A;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32s(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float A = seed + threadIdx.x, B = seed * 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = 256 * wps;     // 256-thread workgroups: one wave per SIMD each
#define RUN(K, N, name)                                                                                        \
        { float ms = time_ms([&] { hipLaunchKernelGGL(K<N>, dim3(grid), dim3(256), 0, 0, out, iters, 3); });     \
          printf("%s NACC=%d waves/SIMD=%d: %.3f ms -> %.1f cycles(@2.4GHz)/MFMA/SIMD\n", name, N, wps, ms,          \
                 ms * 1e-3 * 2.4e9 / ((double)iters * 12 * wps)); }
        RUN(k_bf16, 1, "bf16 32x32x16") RUN(k_bf16, 2, "bf16 32x32x16") RUN(k_bf16, 3, "bf16 32x32x16") RUN(k_bf16, 4, "bf16 32x32x16") RUN(k_bf16, 6, "bf16 32x32x16")
        RUN(k_f32s, 1, "f32 16x16x4 ") RUN(k_f32s, 2, "f32 16x16x4 ") RUN(k_f32s, 4, "f32 16x16x4 ")
    }
    return 0;
}
