// bc_chain.h - a BcResNet block's pointwise + shortcut products chained with the NEXT block's depthwise 3x3 (bc_chain.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct ChainArgs {
    const void* d;                   // [B][H*W][K] this block's depthwise output rows (float32, or binary16 x scale)
    const void* xs;                  // [B][H*W][K] the block input at the strided centres
    void* d_out;                     // [B][Ho*Wo][N] the next block's depthwise output
    void* xs_out;                    // [B][Ho*Wo][N] this block's output at the next block's strided centres
    const unsigned char* packed;     // launch_dual_x3_pack(terms = 2) of this block (N = 2 K outputs)
    const float* dw_wt;              // [9][N] the next block's depthwise weights, tap-major
    int B, H, W;                     // this block's output plane
    int sh, sw, Ho, Wo;              // the next block's depthwise stride and output plane
    // 0: float32 tensors, products on two binary16 terms with a per-pixel scale (DualArgs::h2); 1: bf16 tensors (one binary16 term after
    // the per-pixel scale); 2: binary16 tensors times their plan-time scales (DualArgs::act16 = 2), d_mul / xs_mul = the scales of
    // d_out / xs_out
    int act16 = 0;
    float d_mul = 1.0f, xs_mul = 1.0f;
};
// K = 32 or 64, N = 2 K; the plane of one clip (float32), all packed weights and the depthwise weights must fit the LDS of a CU
bool bc_chain_supported(int K, int H, int W);
hipError_t launch_bc_chain(const ChainArgs& a, int K, int act, int max_grid, hipStream_t s);
