// ffn_x3.h - fused Conformer feed-forward module (ffn_x3.hip): h <- h + rscale * (W2 . swish(W1 . LayerNorm(h) + b1) + b2)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct FfnArgs {
    float* h;                        // [M][D], updated in place (a workgroup reads and writes only its own 128 rows)
    const float* ln_w; const float* ln_b;      // LayerNorm weight / bias [D]
    const unsigned char* packed;     // launch_ffn_x3_pack output: per 32-unit hidden block W1 fragments, W2 fragments, b1
    const float* b2;                 // [D]
    int M;
    float rscale;
    // h2_x > 0: the two-term binary16 arithmetic (NWW_ARITH_F16X3; weights packed by launch_ffn_x3_pack with sw1 / sw2 > 0).
    // Powers of two from plan-time bounds: the LayerNorm-ed rows times h2_x, W1 times h2_w1, the hidden activations times
    // h2_h, W2 times h2_w2 all stay inside the binary16 range (|LayerNorm| <= sqrt(D) max|w| + max|b|, |swish(v)| <= |v|)
    float h2_x = 0.0f, h2_w1 = 1.0f, h2_h = 1.0f, h2_w2 = 1.0f;
    // ---- round 6: a row-local Linear in FRONT of the module (two-term instances only): h0 = [h +] (px . Wp^T + pb), the module runs on h0
    // and writes h = h0 + rscale (...) - ConformerModel.input_proj in front of ff1 (no residual), conv_module.conv2 + residual in front of ff2
    // (architectures.py:441-543).  `packed` must then be made with perm = 1: h0 lives in the accumulator layout, whose feature order the
    // LayerNorm-ed fragments keep.
    const float* px = nullptr;       // [M][KP] rows (pro_k = KP > 0 selects the instance)
    const unsigned char* ppacked = nullptr;      // launch_ffn_x3_pro_pack output
    const float* pb = nullptr;       // [D] bias of the prologue Linear
    int pro_k = 0, pro_res = 0;      // KP (64 or D); add the rows of h (conv2 + residual)
    float p_un = 1.0f;               // 1 / scale of the packed prologue weights
    // ---- round 6: LayerNorm + mean over time BEHIND the module (the block's final LayerNorm feeding only the time average of the last block):
    // rows are not written; per 32-row tile and clip segment the LayerNorm-ed rows are summed EXACTLY (as two planes of integers, 2^-36 of the
    // plan-time bound: order-independent, so a clip's mean does not depend on where the clip sits in the batch) into msum [tiles][2][2][D]
    const float* ln2_w = nullptr; const float* ln2_b = nullptr;
    float* msum = nullptr; int T = 0; float m_scale = 0.0f;
};

// One packed hidden block = a W1 part (D/16 x 3 fragments of 1 KB) and a W2 part (ceil(D/32) x 2 x 3 fragments, then the
// block's 32 biases), each padded to whole 4 KB copy steps (256 lanes x 16 B)
// (nt = terms per value: 3 bf16 or 2 binary16)
__host__ __device__ inline size_t ffn_x3_w1_bytes(int D, int nt = 3) { return ((size_t)(D / 16) * nt * 1024 + 4095) & ~(size_t)4095; }
__host__ __device__ inline size_t ffn_x3_w2_bytes(int D, int nt = 3) { return ((size_t)((D + 31) / 32) * 2 * nt * 1024 + 128 + 4095) & ~(size_t)4095; }
__host__ __device__ inline size_t ffn_x3_block_bytes(int D, int nt = 3) { return ffn_x3_w1_bytes(D, nt) + ffn_x3_w2_bytes(D, nt); }
size_t ffn_x3_packed_bytes(int D);
bool ffn_x3_supported(int D, bool h2 = true);      // D (= d_model; hidden = 4 D) for which an instance is compiled (192 / 256: two-term form only)
// W1 [4D][D], b1 [4D], W2 [D][4D] float32 -> packed
// sw1, sw2 > 0: two binary16 terms of W1 sw1 / W2 sw2 instead of three bf16 terms
// perm = 1: W1's k slots follow the accumulator layout (feature 32 ob + 8 g + 4 half + q at k-block 2 ob + g / 2), for the prologue instances
hipError_t launch_ffn_x3_pack(const float* W1, const float* b1, const float* W2, void* out, int D, hipStream_t s, float sw1 = 0.0f, float sw2 = 0.0f, int perm = 0);
// the prologue Linear's weights W [D][KP] -> ceil(D/32) tiles of KP/16 x 2 fragments (two binary16 terms of W x ws), ffn_x3_pro_tile_bytes(KP) apart
__host__ __device__ inline size_t ffn_x3_pro_tile_bytes(int KP) { return ((size_t)(KP / 16) * 2 * 1024 + 4095) & ~(size_t)4095; }
bool ffn_x3_pro_supported(int D, int KP);
hipError_t launch_ffn_x3_pro_pack(const float* W, void* out, int D, int KP, float ws, hipStream_t s);
// finish of the epilogue instances: out [B][D] = (sum over the clip's tile segments of msum) / (m_scale T)
size_t ffn_x3_msum_bytes(int M, int D);
hipError_t launch_ffn_x3_mean_finish(const float* msum, float* out, int B, int T, int D, float m_scale, hipStream_t s);
hipError_t launch_ffn_x3(const FfnArgs& a, int D, hipStream_t s);
