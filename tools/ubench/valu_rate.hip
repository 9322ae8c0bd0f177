// Micro-benchmark: VALU issue rate on gfx950 (cycles per wave64 instruction per SIMD) for the instruction kinds the
// frontend kernel is made of: v_fma_f32, v_pk_fma_f32, v_add_f32, v_cvt_f32_i32, v_log_f32, and a dependent fma chain.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(1024) k_valu(float* out, int iters, float seed) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    float b = seed * 0.999f, c = 1e-6f;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    f32x2 pb = {b, b}, pc = {c, c};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {          // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 1) {   // 8 independent v_pk_fma_f32 (= 16 fma lanes-worth)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (KIND == 2) {   // 16 v_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 3) {   // 16 v_log_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 4) {   // dependent chain of 16 fma
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (KIND == 5) {   // 16 v_cvt_f32_i32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 6) {   // 16 v_and_b32 (integer)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int instr_per_iter, int threads) {
    float* d;
    hipMalloc(&d, 4096 * 1024 * 4);
    const int iters = 20000, grid = 256 * 2;   // 2 workgroups per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(threads), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = grid/256 * threads/64 / 4
    const double waves_per_simd = (grid / 256.0) * (threads / 64.0) / 4.0;
    const double instr_per_simd = waves_per_simd * (double)iters * instr_per_iter;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s threads=%4d waves/SIMD=%.0f  %.3f ms  -> %.2f cycles(@2.4GHz)/wave-instr/SIMD\n", name, threads,
           waves_per_simd, ms, cyc / instr_per_simd);
    hipFree(d);
}

int main() {
    for (int threads : {128, 256, 512, 1024}) {
        run<0>("v_fma_f32", 16, threads);
        run<1>("v_pk_fma_f32", 8, threads);
        run<2>("v_add_f32", 16, threads);
        run<3>("v_log_f32", 16, threads);
        run<4>("fma dep chain", 16, threads);
        run<5>("v_cvt_f32_i32", 16, threads);
        run<6>("v_and_b32", 16, threads);
    }
    return 0;
}
