// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_32x32x2_f32 on gfx950, and of the bf16 MFMA
// fed by ds_read_b128 fragments (3 reads per 9 MFMAs, the conv2 "bf16x9" inner-loop mix).
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k_bf16(float* out, int iters, uint32_t seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 av = make_uint4(seed + threadIdx.x, seed * 3, seed * 5, seed * 7);
    bf16x8 A = __builtin_bit_cast(bf16x8, av), B = A;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float A = seed + threadIdx.x, B = seed * 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 3 LDS fragment reads (16 B per lane, pixel stride PS bytes) feeding 9 bf16 MFMAs against register-resident B
template <int PS>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int k = threadIdx.x; k < 64 * 1024 / 4; k += 256) reinterpret_cast<uint32_t*>(lds)[k] = seed + k;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int x = 2 * (i >> 2) + (i & 1), dy = (i >> 1) & 1;
    const unsigned char* base = lds + (dy * 40 + x) * PS + 16 * h;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    uint4 bv = make_uint4(seed, seed * 3, seed * 5, seed * 7);
    bf16x8 B0 = __builtin_bit_cast(bf16x8, bv), B1 = B0, B2 = B0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + (it & 15) * PS;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(p), a1 = *reinterpret_cast<const bf16x8*>(p + 32),
                     a2 = *reinterpret_cast<const bf16x8*>(p + 64);
        const unsigned char* q = p + 8 * PS;
        const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(q), c1 = *reinterpret_cast<const bf16x8*>(q + 32),
                     c2 = *reinterpret_cast<const bf16x8*>(q + 64);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, B2, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 6 products per tap (the bf16x6 ratio: 3 fragment reads per 6 MFMAs and tile), row stride WP pixels
template <int PS, int WP>
__global__ void __launch_bounds__(512) k_mix6(float* out, int iters, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int k = threadIdx.x; k < 100 * 1024 / 4; k += 512) reinterpret_cast<uint32_t*>(lds)[k] = seed + k;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int x = 2 * (i >> 2) + (i & 1), dy = (i >> 1) & 1;
    const unsigned char* base = lds + ((dy + 2 * (wave & 3)) * WP + x) * PS + 16 * h;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    uint4 bv = make_uint4(seed, seed * 3, seed * 5, seed * 7);
    bf16x8 B0 = __builtin_bit_cast(bf16x8, bv), B1 = B0, B2 = B0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + (it % 3) * PS + ((it / 3) % 3) * WP * PS;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(p), a1 = *reinterpret_cast<const bf16x8*>(p + 32),
                     a2 = *reinterpret_cast<const bf16x8*>(p + 64);
        const unsigned char* q = p + 16 * PS;
        const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(q), c1 = *reinterpret_cast<const bf16x8*>(q + 32),
                     c2 = *reinterpret_cast<const bf16x8*>(q + 64);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B0, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 6 products per tap with the three WEIGHT fragments also read from LDS (9 reads per 12 MFMAs), NWAVES waves per workgroup
template <int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES) k_mix6w(float* out, int iters, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int k = threadIdx.x; k < 130 * 1024 / 4; k += 64 * NWAVES) reinterpret_cast<uint32_t*>(lds)[k] = seed + k;
    __syncthreads();
    constexpr int PS = 96, WP = 34;
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int x = 2 * (i >> 2) + (i & 1), dy = (i >> 1) & 1;
    const unsigned char* base = lds + ((dy + 2 * (wave & 3)) * WP + x) * PS + 16 * h;
    const unsigned char* wbase = lds + 100 * 1024 + lane * 16;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const unsigned char* p = base + (it % 3) * PS + ((it / 3) % 3) * WP * PS;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(p), a1 = *reinterpret_cast<const bf16x8*>(p + 32),
                     a2 = *reinterpret_cast<const bf16x8*>(p + 64);
        const unsigned char* q = p + 16 * PS;
        const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(q), c1 = *reinterpret_cast<const bf16x8*>(q + 32),
                     c2 = *reinterpret_cast<const bf16x8*>(q + 64);
        const unsigned char* w = wbase + (it % 9) * 3 * 1024;
        const bf16x8 B0 = *reinterpret_cast<const bf16x8*>(w), B1 = *reinterpret_cast<const bf16x8*>(w + 1024),
                     B2 = *reinterpret_cast<const bf16x8*>(w + 2048);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, B0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, B0, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 64 * NWAVES + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000, grid = 512;                       // 2 workgroups of 4 waves per CU
    float ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("f32  32x32x2   : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 32 * 32 * 2 * 4 * iters * 4.0 * grid / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 7u); });
    printf("bf16 32x32x16  : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 32 * 32 * 16 * 4 * iters * 4.0 * grid / ms / 1e9);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix<112>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mix<112>, dim3(grid), dim3(256), 65536, 0, out, iters / 4, 7u); });
    printf("mix PS=112     : %.3f ms  %.1f TFLOP/s (bf16 flops)\n", ms, 2.0 * 32 * 32 * 16 * 18 * (iters / 4) * 4.0 * grid / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mix<96>, dim3(grid), dim3(256), 65536, 0, out, iters / 4, 7u); });
    printf("mix PS=96      : %.3f ms  %.1f TFLOP/s (bf16 flops)\n", ms, 2.0 * 32 * 32 * 16 * 18 * (iters / 4) * 4.0 * grid / ms / 1e9);
#define MIX6(PSV, WPV)                                                                                              \
    {                                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix6<PSV, WPV>), hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024); \
        ms = time_ms([&] { hipLaunchKernelGGL((k_mix6<PSV, WPV>), dim3(256), dim3(512), 110 * 1024, 0, out, iters / 4, 7u); });       \
        printf("mix6 PS=%d WP=%d (1 WG x 8 waves per CU): %.3f ms  %.1f TFLOP/s (bf16 flops)\n", PSV, WPV, ms,        \
               2.0 * 32 * 32 * 16 * 12 * (iters / 4) * 8.0 * 256 / ms / 1e9);                                       \
    }
    MIX6(96, 34) MIX6(112, 34) MIX6(112, 40) MIX6(96, 36) MIX6(128, 34)
#define MIX6W(NWV)                                                                                                  \
    {                                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix6w<NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 135 * 1024); \
        ms = time_ms([&] { hipLaunchKernelGGL((k_mix6w<NWV>), dim3(256), dim3(64 * NWV), 135 * 1024, 0, out, iters / 4, 7u); });   \
        printf("mix6 + weights from LDS, %d waves per CU: %.3f ms  %.1f TFLOP/s (bf16 flops)\n", NWV, ms,              \
               2.0 * 32 * 32 * 16 * 12 * (iters / 4) * (double)NWV * 256 / ms / 1e9);                                \
    }
    MIX6W(8) MIX6W(12) MIX6W(16)
    return 0;
}
