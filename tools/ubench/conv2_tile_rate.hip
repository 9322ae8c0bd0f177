// How fast does ONE wave run the conv2 tile loop of trunk_b.hip (54 dependent v_mfma_f32_32x32x16_bf16 per tile, three
// ds_read_b128 per tap one tap ahead), alone on its SIMD and with a second wave?  Matrix-pipe minimum: 54 x 32 = 1728 clocks.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTB_ABL=1] [-DTB_ACC2=1] -I nanowakeword_amd/csrc -I include tools/ubench/conv2_tile_rate.hip -o tools/ubench/conv2_tile_rate
#include "../../nanowakeword_amd/csrc/trunk_b.hip"
#include <stdio.h>

__global__ void __launch_bounds__(512, 2) tile_loop(const unsigned char* wpack, float* out, int tiles, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    for (int k = threadIdx.x; k < 40 * 1024 / 4; k += blockDim.x) reinterpret_cast<uint32_t*>(lds)[k] = 0x3f803f80u;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(wpack) + lane;
    bf16x8 bw[NWA];
#pragma unroll
    for (int j = 0; j < NWA; ++j) bw[j] = to_agpr(wp[j * 64]);
    if (threadIdx.x < 64)
        for (int j = 0; j < NWL; ++j) *reinterpret_cast<bf16x8*>(lds + 36 * 1024 + j * 1024 + lane * 16) = wp[(NWA + j) * 64];
    __syncthreads();
    const int rowB = 34 * 96 + 16;
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const unsigned char* pa = lds + dyi * rowB + xi * 96 + 16 * hi;
    const unsigned char* wl = lds + 36 * 1024 + lane * 16;
    float* dst = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t)
        conv2_tile<ACT_RELU, 6, false>(pa + (t & 3) * 2 * rowB, rowB, bw, wl, 0.1f, -0.1f, 1.0f, 0.0f, dst, 4);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    std::vector<float> w1(16 * 9, 0.1f), w2(32 * 16 * 9, 0.01f);
    float *dw1, *dw2, *dout; unsigned char* dpack; unsigned long long* dclk;
    hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4); hipMalloc(&dpack, trunk_b_packed_bytes());
    hipMalloc(&dout, 256 * 512 * 16); hipMalloc(&dclk, 256 * 8 * 8);
    hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
    launch_trunk_b_pack(dw1, dw2, dpack, 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    const int tiles = 400;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(tile_loop, dim3(256), dim3(64 * waves), 40 * 1024, 0, dpack, dout, tiles, dclk);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(256 * waves);
        hipMemcpy(c.data(), dclk, c.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += (double)v;
        printf("%d wave(s) per SIMD: %.0f clocks per tile and wave -> %.0f per tile on the SIMD (matrix-pipe minimum 1728)\n", waves / 4,
               s / c.size() / tiles, s / c.size() / tiles / (waves / 4));
    }
    return 0;
}
