// dual_x3.h - BcResNet block's pointwise + shortcut products on the bf16 matrix cores (dual_x3.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct DualArgs {
    const float* d;                  // [M][K] depthwise output rows
    const float* xs;                 // [M][K] block input at the strided centres
    float* out;                      // [M][N]
    const unsigned char* packed;     // launch_dual_x3_pack output
    int M, N;
    int nblk = 0;                    // filled by the launcher
    // x != nullptr: xs is not read; the shortcut's rows are gathered from the block input x [B][H][W][K] (channels last) at
    // (oy * sh, ox * sw) of pixel m = (b, oy, ox), M = B * Ho * Wo
    const float* x = nullptr; int H = 0, W = 0, Ho = 0, Wo = 0, sh = 1, sw = 1;
    // 16-bit activations (nww_config.act_dtype; ACT16_* of split_h2.h).  1 = bf16: d, xs / x and out are bf16 arrays; an activation is
    // then ONE bf16 term, so a float32 weight needs three products (hi, mid, lo) instead of six.  2 = binary16: the arrays hold
    // value x the tensor's plan-time power-of-two scale; the weights are packed as TWO binary16 terms of weight x scale
    // (launch_dual_x3_pack with terms = 2, which also folds 1 / (weight scale x input scale) into the BN factors): two products on
    // v_mfma_f32_32x32x16_f16; the result times out_mul (= out's scale) is rounded to binary16, saturating.
    int act16 = 0;
    float out_mul = 1.0f;
    // h2 = 1 (float32 activations only; NWW_ARITH_F16X3): the products on TWO binary16 terms per operand, three per float32 product on
    // v_mfma_f32_32x32x16_f16.  The weights are packed with terms = 2 (scale = f16_wscale, un = 1 / scale); an activation row is
    // scaled by ITS OWN power of two (largest element into [2^14, 2^15), found in registers - no plan-time bound on the tensor is
    // needed) and the accumulators are multiplied back per pixel.
    int h2 = 0;
    // mean_out != nullptr (the last block): out is NOT written; the block's output is averaged over the mean_P = Ho * Wo pixels of
    // every clip instead -> mean_out [M / mean_P][N] float32 (BcResNetModel's global average pool, architectures.py:677-678).  The
    // pixel tiling is then clip-aligned (a wave = 32 pixels of ONE clip), so a clip's sums do not depend on its slot in the batch.
    float* mean_out = nullptr; int mean_P = 0;
};
bool dual_x3_mean_supported(int pixels_per_clip);

// one packed 32-output block: 2 x K/16 x terms fragments of 1 KB + four 32-float folded-BN vectors, padded to whole 4 KB copy steps
__host__ __device__ inline size_t dual_x3_block_bytes(int K, int terms = 3) { return ((size_t)2 * (K / 16) * terms * 1024 + 512 + 4095) & ~(size_t)4095; }
bool dual_x3_supported(int K, int N);
size_t dual_x3_packed_bytes(int K, int N, int terms = 3);
// Wpw, Wsc [N][K]; (a1, b1) folded BN of the pointwise branch, (as, bs) of the shortcut (null = 1 / 0)
// terms = 3: three bf16 terms per weight.  terms = 2 (binary16 activations): two binary16 terms of Wpw x pw_ws / Wsc x sc_ws, and the
// packed BN factors are a1 x pw_un / as x sc_un (pw_un = 1 / (pw_ws x d's scale), sc_un = 1 / (sc_ws x xs's scale))
struct DualPackScales { float pw_ws = 1.0f, sc_ws = 1.0f, pw_un = 1.0f, sc_un = 1.0f; };
hipError_t launch_dual_x3_pack(const float* Wpw, const float* Wsc, const float* a1, const float* b1, const float* as,
                               const float* bs, void* out, int K, int N, hipStream_t s, int terms = 3, DualPackScales sc = DualPackScales{});
hipError_t launch_dual_x3(const DualArgs& a, int K, int act, hipStream_t s);
