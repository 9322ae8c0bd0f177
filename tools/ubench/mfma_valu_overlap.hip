// Micro-benchmark: do MFMA and VALU instructions of two DIFFERENT waves on the same SIMD overlap on gfx950?
// One 512-thread workgroup per CU = two waves per SIMD.  mode 0: both waves MFMA, 1: both VALU, 2: waves 0-3 MFMA and
// waves 4-7 VALU (one of each per SIMD), 3: waves 0-3 MFMA, others idle, 4: waves 4-7 VALU, others idle.
// Also the same with the f32 16x16x4 MFMA and with LDS reads as the second stream.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int F32, int AGPR>
__global__ void __launch_bounds__(512) k(float* out, int iters_m, int iters_v, int mode, uint32_t seed) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_v = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    float s = 0.f;
    if (do_m) {
        if (F32) {
            f32x4 acc[4];
            for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            float A = seed + threadIdx.x, B = 0.5f;
            if (AGPR) {
                for (int it = 0; it < iters_m; ++it)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[a]) : "v"(A), "v"(B));
            } else {
                for (int it = 0; it < iters_m; ++it)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, acc[a], 0, 0, 0);
            }
            for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][3];
        } else {
            f32x16 acc[4];
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
            uint4 av = make_uint4(seed + threadIdx.x, seed * 3, seed * 5, seed * 7);
            bf16x8 A = __builtin_bit_cast(bf16x8, av), B = A;
            if (AGPR) {
                for (int it = 0; it < iters_m; ++it)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[a]) : "v"(A), "v"(B));
            } else {
                for (int it = 0; it < iters_m; ++it)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
            }
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        }
    }
    if (do_v) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
        const float b = 0.999f, c = 1e-6f;
        for (int it = 0; it < iters_v; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        for (int i = 0; i < 16; ++i) s += a[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const char* names[5] = {"both waves MFMA", "both waves VALU", "one MFMA + one VALU wave per SIMD", "MFMA wave alone", "VALU wave alone"};
    for (int f32 = 0; f32 < 4; ++f32) {
        // equal stand-alone durations: 4 MFMA x 32 clk = 128 clk per iteration; 16 VALU x ~5 clk (single wave) ~ 80 clk
        const int im = 20000, iv = 20000;
        printf("--- %s\n", f32 == 1 ? "v_mfma_f32_16x16x4_f32" : f32 == 2 ? "v_mfma_f32_32x32x16_bf16, accumulators in AGPRs" : f32 == 3 ? "v_mfma_f32_16x16x4_f32, accumulators in AGPRs" : "v_mfma_f32_32x32x16_bf16");
        for (int mode = 0; mode < 5; ++mode) {
            float ms = f32 == 1 ? time_ms([&] { hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(512), 0, 0, out, im, iv, mode, 3u); })
                     : f32 == 2 ? time_ms([&] { hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, out, im, iv, mode, 3u); })
                     : f32 == 3 ? time_ms([&] { hipLaunchKernelGGL((k<1, 1>), dim3(256), dim3(512), 0, 0, out, im, iv, mode, 3u); })
                                : time_ms([&] { hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, out, im, iv, mode, 3u); });
            printf("%-36s %.3f ms\n", names[mode], ms);
        }
    }
    return 0;
}
