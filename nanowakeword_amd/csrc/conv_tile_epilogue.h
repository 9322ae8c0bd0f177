// conv_tile_epilogue.h - what happens to one 32 pixel x 32 channel MFMA accumulator tile of the 3x3 convs (trunk.hip,
// conv3_x3.hip): bias / folded BatchNorm / activation in the C layout, then MaxPool2 in-lane (a lane's 4-register
// groups are 2x2 windows), the fused export-form AvgPool, or the plain store.  Shared so that the float32-MFMA and the
// split-operand kernels cannot drift apart.
#pragma once
#include <hip/hip_runtime.h>
#include "layers.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The activation is a template parameter: a run-time switch costs ~5 scalar branches per element, and with one
// wave per SIMD every taken branch is an exposed instruction-fetch bubble (measured: 2.7k cycles per 16 outputs).
template <int ACT>
__device__ __forceinline__ float trunk_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return nww_silu(v);
    return v;
}

struct AvgWin { int kw, sw, ow; int along_y = 0; };   // windows [j*sw, j*sw + kw) along x (along_y: along y, conv3_x3 only), j < ow <= 4, covering all rows (columns)

// acc = tile (R, X): conv rows 2R, 2R+1, columns 16X .. 16X+15; register 4k+q of lane (i, hi) is channel i, column
// 16X + 4k + 2hi + (q & 1), row 2R + (q >> 1).
// POOL: outb = [cout][H2][W2] pooled planes (H2, W2 pooled sizes).  !POOL: outb = [cout][H2][W2] with H2, W2 the conv
// output sizes.  AVG: nothing is stored; wsum[j] collects the lane's share of window j.  seq_ch != 0 (POOL only): see below.
template <int ACT, bool POOL, bool AVG>
__device__ __forceinline__ void conv_tile_epilogue(const f32x16& acc, int R, int X, float bias2, float al2, float be2,
                                                   bool has_bn, float* outb, int i, int hi, int H2, int W2, int r_off,
                                                   float* wsum, AvgWin aw, int seq_ch = 0) {
    if (POOL) {
        float own[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float m;
            if (ACT == ACT_RELU) {
                m = -INFINITY;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[4 * k + q] + bias2;
                    if (has_bn) v = v * al2 + be2;
                    m = fmaxf(m, trunk_act<ACT>(v));
                }
            } else {
                // GELU / SiLU: one minimum, monotone on either side, and bias + folded BN is monotone - the window's maximum of
                // act(bn(v)) sits at its largest or smallest v: two activations instead of four (trunk_b.hip pool_quad)
                float a = fmaxf(fmaxf(acc[4 * k], acc[4 * k + 1]), fmaxf(acc[4 * k + 2], acc[4 * k + 3])) + bias2;
                float b = fminf(fminf(acc[4 * k], acc[4 * k + 1]), fminf(acc[4 * k + 2], acc[4 * k + 3])) + bias2;
                if (has_bn) { a = a * al2 + be2; b = b * al2 + be2; }
                m = fmaxf(trunk_act<ACT>(a), trunk_act<ACT>(b));
            }
            own[k] = m;                              // pooled column 8X + 2k + hi
        }
        // half 0 keeps columns 0..3 of the 8-column segment, half 1 keeps 4..7
        const float s0 = hi ? own[0] : own[2], s1 = hi ? own[1] : own[3];
        const float r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
        float4 o;
        if (hi == 0) { o.x = own[0]; o.y = r0; o.z = own[1]; o.w = r1; }
        else         { o.x = r0; o.y = own[2]; o.z = r1; o.w = own[3]; }
        const int pcol = 8 * X + 4 * hi;
        if (seq_ch) {
            // sequence layout for the recurrent layers (CRNN: permute(0, 3, 1, 2) + flatten, architectures.py:270-272):
            // outb = the clip's [W2][seq_ch] rows, feature index = channel * H2 + row (channel = the caller's base + i)
            float* dq = outb + (size_t)pcol * seq_ch + (size_t)i * H2 + R + r_off;
            if (pcol + 0 < W2) dq[0] = o.x;
            if (pcol + 1 < W2) dq[seq_ch] = o.y;
            if (pcol + 2 < W2) dq[2 * (size_t)seq_ch] = o.z;
            if (pcol + 3 < W2) dq[3 * (size_t)seq_ch] = o.w;
            return;
        }
        float* dst = outb + ((size_t)i * H2 + R + r_off) * W2 + pcol;
        if ((W2 & 3) == 0 && pcol + 3 < W2) {
            *reinterpret_cast<float4*>(dst) = o;
        } else {
            if (pcol + 0 < W2) dst[0] = o.x;
            if (pcol + 1 < W2) dst[1] = o.y;
            if (pcol + 2 < W2) dst[2] = o.z;
            if (pcol + 3 < W2) dst[3] = o.w;
        }
    } else if (AVG) {
        // fused AvgPool over full-height windows along x (the export form of AdaptiveAvgPool2d((1, ow)),
        // _export/onnx.py:146-152): every lane adds its activated outputs to the windows they fall in; the
        // caller reduces the per-lane sums in a fixed order.  The conv output itself never reaches HBM.
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[4 * k + q] + bias2;
                if (has_bn) v = v * al2 + be2;
                v = trunk_act<ACT>(v);
                const int y = 2 * (R + r_off) + (q >> 1), x = 16 * X + 4 * k + 2 * hi + (q & 1);
                if (y < H2 && x < W2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < aw.ow && x >= j * aw.sw && x < j * aw.sw + aw.kw) wsum[j] += v;
                }
            }
    } else {
        // un-pooled: quad k of this lane = columns 16X + 4k + 2hi + {0,1} of rows 2R, 2R+1.  Exchange with the
        // partner half-wave so half 0 owns row 2R and half 1 row 2R+1, 4 consecutive columns per quad.
        const int y = 2 * (R + r_off) + hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t2 = acc[4 * k + q] + bias2;
                if (has_bn) t2 = t2 * al2 + be2;
                v[q] = trunk_act<ACT>(t2);               // q = 2*dy + dx
            }
            const float s0 = hi ? v[0] : v[2], s1 = hi ? v[1] : v[3];   // half 0 gives away row 2R+1, half 1 row 2R
            const float r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
            float4 o;
            if (hi == 0) { o.x = v[0]; o.y = v[1]; o.z = r0; o.w = r1; }      // row 2R  : cols 4k..4k+3
            else         { o.x = r0; o.y = r1; o.z = v[2]; o.w = v[3]; }      // row 2R+1: cols 4k..4k+3
            const int col = 16 * X + 4 * k;
            if (y < H2) {
                float* dst = outb + ((size_t)i * H2 + y) * W2 + col;
                if (col + 0 < W2) dst[0] = o.x;
                if (col + 1 < W2) dst[1] = o.y;
                if (col + 2 < W2) dst[2] = o.z;
                if (col + 3 < W2) dst[3] = o.w;
            }
        }
    }
}
