#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// fused Conv(1->C1)+[BN]+act+pool -> Conv(C1->C2)+[BN]+act+pool ; see trunk.hip
struct TrunkArgs {
    const float* in;                                   // [B][H][W]
    const float *w1, *b1, *al1, *be1;                  // conv1 [C1][1][3][3], bias (may be null), folded BN (may be null)
    const float *w2, *b2, *al2, *be2;                  // conv2 [C2][C1][3][3]
    float* out;                                        // [B][C2][H/4][W/4]
    int B, H, W, act;
    int dbg = 0;                                       // ablation only: bit0 skip conv1, bit1 skip conv2
    int bn_pos = 0;                                    // trunk_b, two-term form: every folded-BN factor al1 / al2 is >= 0 (plan-time check)
    int strips = 1;                                    // row strips per clip (set by launch_cnn_trunk)
    // > 0: write the output as the split-operand GEMM's A tiles instead of [B][C2][H/4][W/4]: 128-clip row blocks x
    // out_blocked k-tiles of 32 features, each (row block, k-tile) a contiguous [128][32] float tile (gemm_x3.hip)
    int out_blocked = 0;
    unsigned long long* trace = nullptr;               // tools/ubench/trunk_trace.hip only (-DNWW_TRACE): s_memtime stamps per wave and phase
    // trunk_b: weight fragments packed once at plan time (launch_trunk_b_pack); workgroups [wg_end[s-1], wg_end[s]) own row
    // strip s for life (wg_end[0] == 0: strip = blockIdx % strips instead)
    const unsigned char* wpack = nullptr;
    int wg_end[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // trunk_b, streaming hop (nww_stream.hip): `in` is a window inside a per-clip ring (in_clip_stride floats between clips, 0 = H * W);
    // only the pooled output rows [sub_a[s], sub_b[s]) of n_sub explicit strips are computed (n_sub = 0: all rows, `strips` strips);
    // with out_ring_rows > 0 output row R of channel c goes to out + b * out_clip_stride + c * out_ch_stride + ((out_row0 + R) %
    // out_ring_rows) * (W / 4) - a ring of pooled rows per clip and channel, read back by conv3_x3 with the same row map
    size_t in_clip_stride = 0, out_clip_stride = 0, out_ch_stride = 0;
    int out_ring_rows = 0, out_row0 = 0;
    int n_sub = 0, sub_a[4] = {0, 0, 0, 0}, sub_b[4] = {0, 0, 0, 0};
    // trunk_b, products = 3 (two binary16 terms per operand): powers of two fixed at plan time from bounds on the operands -
    // f16_in scales the input, the conv1 / conv2 accumulators are f16_k1 / f16_k2 times the true sums (operand scale x weight
    // scale), conv1's output is kept as f16_s1 times its value (conv2's operand), the output leaves f16_so times its value
    // the input is clamped to +-f16_clamp (the bound f16_in was derived from) before it is scaled
    float f16_in = 1.0f, f16_k1 = 1.0f, f16_s1 = 1.0f, f16_k2 = 1.0f, f16_so = 1.0f, f16_clamp = 65504.0f;
};
struct TrunkStrip {
    int R2a, R2b, a1_base, a1_lo, a1_hi, a1_rows, iy0, in_rows, y_lo, y_hi;
};
// the strip that produces pooled output rows [R2a, R2b): the A1 / input rows it needs (seam rows are recomputed by every strip
// that needs them, with the same arithmetic: results do not depend on how a clip is cut)
__host__ __device__ inline TrunkStrip trunk_strip_rows(int H, int R2a, int R2b) {
    const int H1 = H / 2;
    TrunkStrip g;
    g.R2a = R2a;
    g.R2b = R2b;
    g.a1_base = 2 * g.R2a - 1;                                  // A1 row kept at local row 0
    g.a1_lo = g.a1_base < 0 ? 0 : g.a1_base;
    g.a1_hi = 2 * g.R2b < H1 - 1 ? 2 * g.R2b : H1 - 1;
    g.a1_rows = 2 * (g.R2b - g.R2a) + 2;
    g.iy0 = 2 * g.a1_lo - 1;                                    // input row kept at local row 0
    const int iy1 = 2 * g.a1_hi + 2;
    g.in_rows = iy1 - g.iy0 + 1;
    g.y_lo = g.iy0 < 0 ? 0 : g.iy0;
    g.y_hi = iy1 < H - 1 ? iy1 : H - 1;
    return g;
}
__host__ __device__ inline TrunkStrip trunk_strip(int H, int S, int s) {
    const int H2 = H / 4;
    return trunk_strip_rows(H, (s * H2 + S - 1) / S, ((s + 1) * H2 + S - 1) / S);
}
size_t trunk_lds_bytes(int C1, int H, int W, int strips);
// strips needed for a workgroup to fit in LDS (0 = does not fit at all); *wgs_per_cu = 2 when two workgroups share a CU
int trunk_pick_strips(int C1, int H, int W, int* wgs_per_cu);
hipError_t launch_cnn_trunk(const TrunkArgs& a, int C1, int C2, int max_grid, hipStream_t s);

// standalone 3x3 conv (pad 1, stride 1) + bias/BN + act (+ MaxPool2) on MFMA f32 for C1 = 32 input channels and
// Cout a multiple of 32: in [B][32][H][W] -> out [B][Cout][H or H/2][W or W/2]; one workgroup per clip, input staged in LDS.
struct ConvMfmaArgs {
    const float* in; const float* w; const float *bias, *alpha, *beta; float* out;
    int B, H, W, Cout, act, pool;
    int Cin = 32;          // conv3_x3 only: 64 takes the wide instance (conv3_x3_wide_fits)
    // conv3_x3 only, k-split passes (more than 32 input channels on the 32-channel instances): the pass reads input channels
    // [w_cin_off, w_cin_off + 32) of w_cin - planes and weight columns alike - and starts its sums from acc_in [B][Cout][H][W]
    // (raw sums of the earlier passes; act = ACT_NONE writes such sums: no bias, no BN, un-pooled)
    int w_cin = 32, w_cin_off = 0;
    const float* acc_in = nullptr;
    int strip_h = 0;       // conv3_x3 only: > 0 - the plane goes through LDS in strips of this many rows (conv3_x3_strip_rows; pooled or raw output)
    // conv3_x3 only, streaming hop: the input planes live in per-clip rings of pooled rows (TrunkArgs::out_ring_rows): row y of
    // channel c of clip b at in + b * in_clip_stride + c * in_ch_stride + ((in_row0 + y) % in_ring_rows) * W
    size_t in_clip_stride = 0, in_ch_stride = 0;
    int in_ring_rows = 0, in_row0 = 0;
    // ... and (pooled + seq_out mode) only the pooled output rows outside [keep_lo, keep_hi] are computed; row j of that range is
    // row j + keep_shift of the previous hop's sequence buffer seq_prev (same layout as out), copied over
    const float* seq_prev = nullptr; int keep_lo = 0, keep_hi = -1, keep_shift = 0;
    int row_pitch = 0;     // conv3_x3: bytes between the rows of the LDS plane (filled by the launcher)
    // fused AvgPool2d(kernel (H, avg_kw), stride (H, avg_sw)) -> out [B][Cout][avg_ow] when avg_ow > 0 (pool must be 0)
    int avg_kw = 0, avg_sw = 0, avg_ow = 0;
    int avg_y = 0;         // conv3_x3 only: the windows run along y and cover all columns (the same pool on a transposed plane)
    // conv3_x3 only, pooled mode: write out [B][W/2][Cout * H/2] (feature = channel * H/2 + row), the recurrent layers' input
    int seq_out = 0;
    // conv3_x3, two binary16 terms per operand (NWW_ARITH_F16X3): h2_in > 0 - the input times h2_in and the weights times h2_w
    // (powers of two from plan-time bounds) stay inside the binary16 range
    float h2_in = 0.0f, h2_w = 1.0f;
};
size_t conv_mfma_lds_bytes(int C1, int H, int W);
hipError_t launch_conv3x3_mfma(const ConvMfmaArgs& a, int C1, int max_grid, hipStream_t s);
// split-operand bf16 instance of the same stage (conv3_x3.hip): 32 input channels, Cout % 32 == 0, six products
size_t conv3_x3_lds_bytes(int H, int W, int avg_ow);
bool conv3_x3_fits(int H, int W, int Cout, int avg_ow, int pool);
int conv3_x3_strip_rows(int H, int W, int Cout);               // planes of more than 512 pixels: rows per strip, 0 = none
size_t conv3_x3_strip_lds_bytes(int sh, int W);
bool conv3_x3_wide_fits(int Cin, int H, int W, int Cout);      // 64 input channels, pooled, two binary16 terms only
size_t conv3_x3_wide_lds_bytes(int Cin, int H, int W);
hipError_t launch_conv3_x3(const ConvMfmaArgs& a, int max_grid, hipStream_t s);

// Conv2d(1, 32, 3, p1) + bias/BN + act + MaxPool2 on MFMA, channels-last output [B][H/2][W/2][32] (BcResNet init conv)
struct Conv1NhwcArgs {
    const float* in; const float* w; const float* bias; const float* alpha; const float* beta; float* out;
    int B, H, W, act;
};
bool conv1_pool_nhwc_mfma_fits(int H, int W);
// the same fused with the first block's depthwise 3x3 (stride sh x sw, pad 1): d_out, xs_out [B][Ho][Wo][32]
struct Conv1DwArgs {
    const float* in; const float* w; const float* bias; const float* alpha; const float* beta;
    const float* dw_wt;              // [9][32] depthwise weights, tap-major
    float* d_out; float* xs_out;
    int B, H, W, act, sh, sw;
    int Ho = 0, Wo = 0, rows_dw = 0; // filled by the launcher
    int bf16_out = 0;                // nww_config.act_dtype: 1 = d_out / xs_out are bf16 arrays, 2 = binary16 arrays of value x scale
    float d_scale = 1.0f, xs_scale = 1.0f;   // bf16_out == 2: the tensors' plan-time power-of-two scales
    const unsigned char* wpack = nullptr;   // launch_bc_front_b only: conv weight fragments (launch_bc_front_b_pack)
    // launch_bc_front_b with products = 3 (two binary16 terms; weights from launch_bc_front_b_pack_f16 with scale sw): the input is
    // clamped to +-f16_clamp and multiplied by f16_in; f16_unscale = 1 / (f16_in * sw)
    float f16_in = 0.0f, f16_clamp = 0.0f, f16_unscale = 1.0f;
    int bn_pos = 0;                  // launch_bc_front_b: every folded-BN factor alpha is >= 0 (the pooling takes the window's maximum only)
};
// the same stage with the convolution from split operands on the bf16 matrix cores (trunk_b.hip); products = 6 / 9
size_t bc_front_b_packed_bytes();
hipError_t launch_bc_front_b_pack(const float* w1, unsigned char* packed, hipStream_t s);
hipError_t launch_bc_front_b_pack_f16(const float* w1, unsigned char* packed, float sw, hipStream_t s);
int bc_front_b_rows(int H, int W, int sh);         // depthwise rows per LDS strip, 0 = does not fit
hipError_t launch_bc_front_b(const Conv1DwArgs& a, int products, int max_grid, hipStream_t s);
int conv1_pool_dw_rows(int H, int W, int sh);      // depthwise rows per LDS strip, 0 = does not fit
hipError_t launch_conv1_pool_dw_nhwc(const Conv1DwArgs& a, int max_grid, hipStream_t s);
hipError_t launch_conv1_pool_nhwc_mfma(const Conv1NhwcArgs& a, int max_grid, hipStream_t s);

// The same trunk with both convolutions on the bf16 matrix cores by exact operand splitting (trunk_b.hip): every float32
// product is formed from `products` = 9 (all partial products, exact) or 6 (the terms below 2^-23 of the product dropped)
// v_mfma_f32_32x32x16_bf16; conv1 as a transposed product per pooled pixel, conv1's output kept in LDS as three bf16 terms
// per value, channels last.
// products = 3: two binary16 terms per operand, three partial products on v_mfma_f32_32x32x16_f16 (weights packed by
// launch_trunk_b_pack_f16 with their power-of-two scales; TrunkArgs::f16_*).
size_t trunk_b_lds_bytes(int H, int W, int strips, int products = 6);
int trunk_b_pick_strips(int H, int W);
// does an explicit strip of pooled rows [r2a, r2b) (streaming hop) fit in LDS?  Judged on the three-term form's need, like the strip count
bool trunk_b_rows_fit(int H, int W, int r2a, int r2b);
size_t trunk_b_packed_bytes();
hipError_t launch_trunk_b_pack(const float* w1, const float* w2, unsigned char* packed, hipStream_t s);   // a.wpack
hipError_t launch_trunk_b_pack_f16(const float* w1, const float* w2, unsigned char* packed, float sw1, float sw2, hipStream_t s);
hipError_t launch_cnn_trunk_b(const TrunkArgs& a, int products, int max_grid, hipStream_t s);
