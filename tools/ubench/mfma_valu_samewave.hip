// Micro-benchmark: can ONE wave (alone on its SIMD) hide VALU instructions behind its own MFMAs on gfx950?
// 256-thread workgroup per CU = one wave per SIMD.  Each iteration issues 8 v_mfma_f32_32x32x16_bf16 (accumulators in
// AGPRs; CHAIN = 1: all on one accumulator, a dependent chain; CHAIN = 0: two accumulators alternating) with K
// independent v_fma_f32 behind every MFMA.  Prints ns per MFMA: flat in K = the VALU work is hidden.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_samewave.hip -o mfma_valu_samewave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int CHAIN, int TRANS, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float* out, int iters, uint32_t seed) {
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 av = make_uint4(seed + threadIdx.x, seed * 3, seed * 5, seed * 7);
    bf16x8 A = __builtin_bit_cast(bf16x8, av), B = A;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 0.001f + i;
    const float b = 0.999f, c = 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[CHAIN ? 0 : (m & 1)]) : "v"(A), "v"(B));
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (TRANS && j == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(c));
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int K, int CHAIN, int TRANS, int THREADS = 256> static void run(float* out) {
    const int iters = 20000;
    const float ms = time_ms([&] { hipLaunchKernelGGL((k<K, CHAIN, TRANS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, 3u); });
    // per SIMD: THREADS / 256 waves share the matrix pipe
    printf("%d wave(s)/SIMD %s K=%d%s: %.1f ns per MFMA of the SIMD (+%d VALU each)\n", THREADS / 256, CHAIN ? "chain     " : "two accums", K,
           TRANS ? " (1 v_exp)" : "", ms * 1e6 / (iters * 8.0 * (THREADS / 256)), K);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run<0, 1, 0>(out); run<2, 1, 0>(out); run<4, 1, 0>(out); run<6, 1, 0>(out); run<8, 1, 0>(out); run<12, 1, 0>(out);
    run<0, 0, 0>(out); run<2, 0, 0>(out); run<4, 0, 0>(out); run<6, 0, 0>(out); run<8, 0, 0>(out); run<12, 0, 0>(out);
    run<4, 1, 1>(out); run<6, 1, 1>(out);
    // two and three waves per SIMD, each interleaving its own MFMAs and VALU work
    run<0, 1, 0, 512>(out); run<4, 1, 0, 512>(out); run<6, 1, 0, 512>(out); run<8, 1, 0, 512>(out); run<10, 1, 0, 512>(out); run<12, 1, 0, 512>(out); run<16, 1, 0, 512>(out);
    run<8, 1, 0, 768>(out); run<12, 1, 0, 768>(out); run<16, 1, 0, 768>(out);
    return 0;
}
