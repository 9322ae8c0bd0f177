// Power drawn by pure instruction streams on all 256 CUs (sample rocm-smi beside it: tools/power_mix.sh):
//   mode 0: v_mfma_f32_32x32x16_bf16 back to back (AGPR accumulators), 2 waves per SIMD
//   mode 1: v_fma_f32, 8 independent chains per lane, 4 waves per SIMD
//   mode 2: ds_read_b128 from conflict-free addresses, 4 waves per SIMD
//   mode 3: v_mfma_f32_32x32x2_f32 back to back
//   mode 4: as mode 0 with eight different pseudo-random operand pairs in rotation (operand toggling)
//   mode 6 / 7 / 8: mode 4 with A constant / A held for four consecutive instructions / a single accumulator
//   mode 5: mode 4 on waves 0-7 and the v_fma_f32 chains of mode 1 on waves 8-15 of the same workgroup
// Prints the sustained instruction rate; energy per wave-instruction = (power - idle power) / rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/power_mix.hip -o tools/ubench/power_mix ; run: power_mix <mode> <seconds>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) k_mfma(float* out, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    asm volatile("" : "+a"(acc0), "+a"(acc1));
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x + r); b[r] = (__bf16)(float)(r + 1); }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// ORD 0: A and B both change with every instruction; 1: A constant, B rotating; 2: A held for four consecutive
// instructions (both accumulators, two B's each); 3: as 0 on ONE accumulator (dependent chain)
template <bool MIX, int ORD = 0>
__global__ void __launch_bounds__(MIX ? 1024 : 512) k_mfma_rand(float* out, int iters) {
    if (MIX && threadIdx.x >= 512) {
        float x[8];
        for (int r = 0; r < 8; ++r) x[r] = (float)(threadIdx.x + r);
        const float m = 1.0000001f, c = 0.5f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(m), "v"(c));
        float s = 0.f;
        for (int r = 0; r < 8; ++r) s += x[r];
        if (s == 12345.678f) out[threadIdx.x] = s;
        return;
    }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    asm volatile("" : "+a"(acc0), "+a"(acc1));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 av[8], bv[8];
    for (int p = 0; p < 8; ++p)
        for (int r = 0; r < 4; ++r) {
            // random mantissas and signs, exponents of both halves kept near 1 so that nothing overflows
            av[p][r] = (hash32(threadIdx.x * 64 + p * 8 + r) & 0x807f807fu) | 0x3f003f00u;
            bv[p][r] = (hash32(threadIdx.x * 64 + p * 8 + r + 4) & 0x807f807fu) | 0x3a003a00u;
        }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a0 = ORD == 1 ? 0 : ORD == 2 ? (u & ~1) : u, a1 = ORD == 1 ? 0 : ORD == 2 ? (u & ~1) : (u + 3) & 7;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[a0]), __builtin_bit_cast(bf16x8, bv[u]), acc0, 0, 0, 0);
            if (ORD == 3) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[a1]), __builtin_bit_cast(bf16x8, bv[(u + 5) & 7]), acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[a1]), __builtin_bit_cast(bf16x8, bv[(u + 5) & 7]), acc1, 0, 0, 0);
        }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(512) k_mfma_f32(float* out, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = threadIdx.x, b = 1.5f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(1024) k_valu(float* out, int iters) {
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = (float)(threadIdx.x + r);
    const float m = 1.0000001f, c = 0.5f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(m), "v"(c));
    float s = 0.f;
    for (int r = 0; r < 8; ++r) s += x[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(1024) k_lds(float* out, int iters) {
    __shared__ float4 buf[1024];
    buf[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 16; ++u) { const float4 v = buf[(j + 64 * u) & 1023]; acc.x += v.x; acc.y += v.y; }
    if (acc.x + acc.y == 12345.678f) out[threadIdx.x] = acc.x;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    float* out; hipMalloc(&out, 1 << 20);
    const int iters = 20000;
    const double per_wave = mode == 0 || mode == 3 ? 16.0 * iters : mode == 1 ? 16.0 * iters : 16.0 * iters;      // instructions of interest per wave and launch
    const int threads = mode == 0 || mode >= 3 ? 512 : 1024;      // mode 5 counts its MFMA waves
    double launches = 0;
    auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 4; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k_mfma, dim3(256), dim3(512), 0, 0, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(k_valu, dim3(256), dim3(1024), 0, 0, out, iters);
            else if (mode == 2) hipLaunchKernelGGL(k_lds, dim3(256), dim3(1024), 0, 0, out, iters);
            else if (mode == 4) hipLaunchKernelGGL(k_mfma_rand<false>, dim3(256), dim3(512), 0, 0, out, iters);
            else if (mode == 6) hipLaunchKernelGGL((k_mfma_rand<false, 1>), dim3(256), dim3(512), 0, 0, out, iters);
            else if (mode == 7) hipLaunchKernelGGL((k_mfma_rand<false, 2>), dim3(256), dim3(512), 0, 0, out, iters);
            else if (mode == 8) hipLaunchKernelGGL((k_mfma_rand<false, 3>), dim3(256), dim3(512), 0, 0, out, iters);
            else if (mode == 5) hipLaunchKernelGGL(k_mfma_rand<true>, dim3(256), dim3(1024), 0, 0, out, iters);
            else hipLaunchKernelGGL(k_mfma_f32, dim3(256), dim3(512), 0, 0, out, iters);
        }
        hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double wave_instr = launches * 256.0 * (threads / 64) * per_wave;
    printf("mode %d: %.3e wave-instructions per second (%s)\n", mode, wave_instr / el,
           mode == 0 ? "v_mfma_f32_32x32x16_bf16" : mode == 4 ? "v_mfma_f32_32x32x16_bf16, operands toggling" :
           mode == 5 ? "v_mfma_f32_32x32x16_bf16 toggling, beside v_fma_f32 waves" :
           mode == 6 ? "bf16 MFMA, A constant, B toggling" : mode == 7 ? "bf16 MFMA, A held for four instructions" :
           mode == 8 ? "bf16 MFMA toggling, one accumulator" : mode == 1 ? "v_fma_f32" : mode == 2 ? "ds_read_b128" : "v_mfma_f32_32x32x2_f32");
    return 0;
}
