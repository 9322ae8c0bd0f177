// trunk.hip - fused conv trunk of the CNN-family heads for gfx950:
//     x[H][W] -> Conv2d(1,C1,3,p1) (+BN) + act + MaxPool2 -> Conv2d(C1,C2,3,p1) (+BN) + act + MaxPool2 -> [C2][H2][W2]
// (reference: CNNModel.conv1/conv2 architectures.py:54-59,74-75; the first two stages of CRNNModel.cnn
//  :217-225 and of E2E_MelSpectrogram_CNN.conv_block :840-849 have the same shape with BatchNorm folded in).
//
// One workgroup (4 waves) owns one clip at a time; nothing between the input and the pooled conv2 output
// touches HBM:
//   P0  input plane -> LDS with a zero halo (coalesced 16-byte global loads)
//   P1  conv1 on v_mfma_f32_16x16x4_f32: M = 16 pixels (2 rows x 8 columns), N = 16 channels, K = 9 taps padded
//       to 12; weights are 3 VGPRs; bias/BN/act and the 2x2 max happen in the C layout (a lane's 4 accumulator
//       registers are one pooling window); result -> LDS plane set A1[C1][H1+2][W1+2] (zero halo), 113 KB for
//       the (101,64) log-mel.
//   P2  conv2 as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32, fmaf chain): M = 32 conv2 output
//       pixels arranged as a 2-row x 16-column patch so that each lane's 4-register groups of the C layout
//       are exactly the 2x2 pooling quads; N = 32 output channels; K = C1*9 taken as (channel pair) x tap so
//       the two half-waves read the same tap of channels 2c and 2c+1.  B fragments (weights) stay in
//       C1*9/2 VGPRs for the whole kernel; A fragments are single ds_read_b32 per MFMA.  Two tiles are in
//       flight per wave (independent accumulators).  Epilogue: bias/BN/act, in-lane 2x2 max, one shuffle
//       pair with the partner half-wave, 16-byte stores.
// Per clip: 2*9*C1*C2*(2*H2)*(2*W2) conv2 flops + 2*9*C1*(2*H1)*(2*W1) conv1 flops; HBM bytes = 4*H*W in +
// 4*C2*H2*W2 out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "trunk.h"
#include "split_h2.h"

#include "conv_tile_epilogue.h"

// Row strips.  A clip is cut into S strips of pooled output rows; strip s = pooled rows [R2a, R2b) needs the A1 rows
// 2*R2a-1 .. 2*R2b (one halo row each side, recomputed by both neighbours at a seam) and for those the input rows
// 2*a1_lo-1 .. 2*a1_hi+2.  Rows outside the image are the zero halo.  With S = 2 a (101,64) strip takes 76 KB of LDS,
// so two workgroups share a CU and one's conv1 / epilogue gaps are filled by the other's conv2 MFMAs.
size_t trunk_lds_bytes(int C1, int H, int W, int S) {
    const int W1 = W / 2;
    if (S == 1) {                                   // whole-clip instance: planes of H+2 and H/2+2 rows
        const size_t in_f = ((size_t)(H + 2) * (W + 2) + 3) & ~(size_t)3;
        return (in_f + (size_t)C1 * (H / 2 + 2) * (W1 + 2) + 64) * sizeof(float);
    }
    size_t worst = 0;
    for (int s = 0; s < S; ++s) {
        const TrunkStrip g = trunk_strip(H, S, s);
        const size_t in_f = ((size_t)g.in_rows * (W + 2) + 3) & ~(size_t)3;
        const size_t a1_f = (size_t)C1 * g.a1_rows * (W1 + 2) + 64;
        if (in_f + a1_f > worst) worst = in_f + a1_f;
    }
    return worst * sizeof(float);
}
// strips per clip: the whole clip (one 8-wave workgroup per CU) whenever it fits in LDS, else the fewest strips that
// do.  Two 4-wave workgroups per CU on half-clip strips (NWW_TRUNK_STRIPS=2) measured no better: with equal work the
// two stay in lock-step, so one's conv1 does not fall under the other's conv2.
int trunk_pick_strips(int C1, int H, int W, int* wgs_per_cu) {
    const int H2 = H / 4;
    for (int S = 1; S <= H2; ++S)
        if (trunk_lds_bytes(C1, H, W, S) <= 160 * 1024) { *wgs_per_cu = 1; return S; }
    *wgs_per_cu = 0;
    return 0;
}

// conv2 for tile t (and t+1 when TWO): 32 pixels x 32 channels x K = C1*9 each, A operands prefetched one channel
// pair ahead of the MFMAs that consume them, then bias/BN/act, in-lane 2x2 max, half-wave exchange, 16-byte store.
template <int C1, int ACT, bool TWO, bool POOL = true, bool AVG = false>
__device__ __forceinline__ void conv2_tiles(const float* A1, int lane_off, int P1, int Wp1, int nX, int t,
                                            const float (&breg)[C1 * 9 / 2], float bias2, float al2, float be2,
                                            bool has_bn, float* outb, int i, int hi, int H2, int W2, int r_off = 0,
                                            float* wsum = nullptr, AvgWin aw = AvgWin{0, 0, 0}) {
    // POOL: outb = [cout][H2][W2] pooled planes (H2, W2 pooled sizes).  !POOL: outb = [cout][H2][W2] with H2, W2 the
    // conv output sizes; the tile still covers conv rows 2R, 2R+1 and 16 columns.
    const int R0 = t / nX, X0 = t - R0 * nX;
    const int t1 = TWO ? t + 1 : t;
    const int R1 = t1 / nX, X1 = t1 - R1 * nX;
    const float* pa = A1 + lane_off + (2 * R0) * Wp1 + 16 * X0;
    const float* pb = A1 + lane_off + (2 * R1) * Wp1 + 16 * X1;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float na[9], nb[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3) * Wp1 + (tap % 3);
        na[tap] = pa[off];
        if (TWO) nb[tap] = pb[off];
    }
#pragma unroll
    for (int c2 = 0; c2 < C1 / 2; ++c2) {
        float ca[9], cb[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) { ca[tap] = na[tap]; if (TWO) cb[tap] = nb[tap]; }
        if (c2 + 1 < C1 / 2) {
            const float* qa = pa + 2 * (c2 + 1) * P1;
            const float* qb = pb + 2 * (c2 + 1) * P1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = (tap / 3) * Wp1 + (tap % 3);
                na[tap] = qa[off];
                if (TWO) nb[tap] = qb[off];
            }
        }
        // keep the next pair's LDS reads ABOVE this pair's MFMAs (hipcc otherwise sinks each read to one MFMA before
        // its use and waits on it at once, exposing the LDS latency on every step)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[tap], breg[c2 * 9 + tap], acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[tap], breg[c2 * 9 + tap], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    conv_tile_epilogue<ACT, POOL, AVG>(acc0, R0, X0, bias2, al2, be2, has_bn, outb, i, hi, H2, W2, r_off, wsum, aw);
    if (TWO) conv_tile_epilogue<ACT, POOL, AVG>(acc1, R1, X1, bias2, al2, be2, has_bn, outb, i, hi, H2, W2, r_off, wsum, aw);
}

// STRIP = false is the whole-clip instance: its geometry is written as plain functions of H and W because the kernel
// sits at 256 VGPRs and hipcc's allocation is fragile there (the same numbers routed through the strip arithmetic
// cost 32 spilled VGPRs and 9 % of the kernel's time).
template <int C1, int C2, int ACT, int NW, bool STRIP>
__global__ void __launch_bounds__(64 * NW, 2) cnn_trunk_kernel(TrunkArgs a) {
    constexpr int NTHR = 64 * NW;
    static_assert(C1 == 16 && C2 == 32, "C1 == 16 (one 16-wide MFMA column block), C2 == 32");
    constexpr int KS = C1 * 9 / 2;                         // MFMA steps per tile (2 k per step)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
    const int S = a.strips;
    // a workgroup keeps ONE strip index for its whole life, so the zero halos written once below stay valid
    // Only a handful of derived numbers stay live (the strip struct itself would cost ~10 SGPRs the kernel lacks):
    // local row of input row y is y - iy0 with iy0 = 2*a1_lo - 1, so conv1 of local A1 row Rl reads local input rows
    // 2*Rl ..; A1 row a1_lo sits at local A1 row a1_shift (1 below the zero halo at the image top, 0 at a seam).
    int n_a1, a1_shift, nR2, n_in, row_shift, P1, in_f;
    size_t in_off, out_off;
    const int Wp0 = W + 2, Wp1 = W1 + 2;
    if (!STRIP) {
        n_a1 = H1; a1_shift = 1; nR2 = H2; n_in = H * W; row_shift = 1;
        P1 = (H1 + 2) * Wp1;
        in_f = ((H + 2) * Wp0 + 3) & ~3;
        in_off = 0; out_off = 0;
    } else {
        const TrunkStrip sg = trunk_strip(H, S, (int)blockIdx.x % S);
        n_a1 = sg.a1_hi - sg.a1_lo + 1;
        a1_shift = sg.a1_lo - sg.a1_base;
        nR2 = sg.R2b - sg.R2a;
        n_in = (sg.y_hi - sg.y_lo + 1) * W;
        row_shift = sg.y_lo - sg.iy0;
        P1 = sg.a1_rows * Wp1;
        in_f = (sg.in_rows * Wp0 + 3) & ~3;
        in_off = (size_t)sg.y_lo * W;
        out_off = (size_t)sg.R2a * W2;
    }
    float* In = lds;
    float* A1 = lds + in_f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;

    // zero both LDS regions once: halos stay zero, interiors are rewritten per clip
    for (int k = tid; k < in_f + C1 * P1 + 64; k += NTHR) lds[k] = 0.0f;

    // conv2 weights -> B fragments: step s = c2*9 + tap, lane (cout = i, channel = 2*c2 + hi)
    float breg[KS];
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        const int c2 = st / 9, tap = st - c2 * 9;
        breg[st] = a.w2[((size_t)i * C1 + 2 * c2 + hi) * 9 + tap];
    }
    const float bias2 = a.b2 ? a.b2[i] : 0.0f;
    const float al2 = a.al2 ? a.al2[i] : 1.0f, be2 = a.al2 ? a.be2[i] : 0.0f;

    // conv1 weights -> B fragments of the 16x16x4 MFMA: lane (g = l>>4, channel = l&15), step st: tap 4*st + g
    float w1reg[3];
    int tap_off1[3];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int tap = 4 * st + (lane >> 4);
        w1reg[st] = (tap < 9 && (lane & 15) < C1) ? a.w1[(size_t)(lane & 15) * 9 + tap] : 0.0f;
        tap_off1[st] = tap < 9 ? (tap / 3) * Wp0 + (tap % 3) : 0;
    }
    const float bias1 = a.b1 ? a.b1[lane & 15] : 0.0f;
    const float al1 = a.al1 ? a.al1[lane & 15] : 1.0f, be1 = a.al1 ? a.be1[lane & 15] : 0.0f;

    // conv2 tiling of this strip (tile rows are local pooled rows 0 .. R2b-R2a-1)
    const int nX = (W1 + 15) / 16, nT = nR2 * nX;
    // leftover tiles go to waves 0..rem-1, which sit on different SIMDs (waves w and w+4 share one)
    const int t_base = nT / NW, t_rem = nT - t_base * NW;
    const int t_begin = wave * t_base + min(wave, t_rem), t_end = t_begin + t_base + (wave < t_rem ? 1 : 0);
    // lane's pixel inside a tile: i = 4*quad + 2*dy + dx  (quad along x)
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const int lane_off = hi * P1 + dyi * Wp1 + xi;

    // input rows y_lo..y_hi of a clip -> LDS `In` (local row y - iy0, interior at column +1).  The first clip is loaded
    // synchronously; afterwards the NEXT clip's rows are fetched into registers at the start of conv2 (In is only read
    // by conv1) and written to LDS when conv2 is done, so their HBM latency hides under the MFMA phase.
    const bool vec_in = (W & 3) == 0 && n_in <= 16 * NTHR;
    auto store_plane_regs = [&](const float4 (&pre)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx4 = tid + q * NTHR;
            if (idx4 < n_in / 4) {
                const int idx = idx4 * 4, y = idx / W, x = idx - y * W;
                float* d = In + (y + row_shift) * Wp0 + x + 1;
                d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
            }
        }
    };
    auto load_plane_sync = [&](const float* xin) {
        for (int idx = tid; idx < n_in; idx += NTHR) {
            const int y = idx / W, x = idx - y * W;
            In[(y + row_shift) * Wp0 + x + 1] = xin[idx];
        }
    };
    const int b0 = STRIP ? (int)blockIdx.x / S : (int)blockIdx.x, bstep = STRIP ? (int)gridDim.x / S : (int)gridDim.x;
    __syncthreads();
    if (b0 < a.B) load_plane_sync(a.in + (size_t)b0 * H * W + in_off);
    __syncthreads();
    for (int b = b0; b < a.B; b += bstep) {
        const int bnext = b + bstep;
        // ---------------- P1: conv1 + act + pool -> A1 on v_mfma_f32_16x16x4_f32
        // tile = 16 conv1 pixels as 2 rows x 8 columns, pixel i = 4*quad + 2*dy + dx; K = 9 taps padded to 12
        // (3 steps of 4); lane (i = l&15, g = l>>4) feeds tap 4*step + g.  C layout: column = channel l&15,
        // rows 4g..4g+3 = the 2x2 quad g -> the four accumulator registers of a lane ARE one pooling window.
        if (!(a.dbg & 1)) {
            // groups of 4 horizontally adjacent tiles: one index division per group, the tiles of a group are at
            // constant +8 column offsets (immediate offsets on the LDS reads)
            const int nX1 = (2 * W1 + 7) / 8, ngx = (nX1 + 3) / 4, nG = n_a1 * ngx;
            const int i1 = lane & 15, g1 = lane >> 4;
            const int pix_off = ((i1 >> 1) & 1) * Wp0 + 2 * (i1 >> 2) + (i1 & 1);
            float* a1lane = A1 + i1 * P1 + a1_shift * Wp1 + g1 + 1;
            for (int g = wave; g < nG; g += NW) {
                const int R = g / ngx, X0 = 4 * (g - R * ngx);                          // R = A1 row - a1_lo
                const float* rowp = In + (2 * R) * Wp0 + 8 * X0 + pix_off;
                f32x4 acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    const float* q = rowp + tap_off1[st];
                    float av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = q[8 * u];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], w1reg[st], acc[u], 0, 0, 0);
                }
                float* wr = a1lane + R * Wp1 + 4 * X0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float m = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = acc[u][q] + bias1;
                        if (a.al1) v = v * al1 + be1;
                        m = fmaxf(m, trunk_act<ACT>(v));
                    }
                    if (X0 + u < nX1 && 4 * (X0 + u) + g1 < W1) wr[4 * u] = m;
                }
            }
        }
        __syncthreads();
        // ---------------- P2: conv2 on MFMA; tiles in pairs (two independent accumulators), a lone tile alone
        float* outb = a.out + (size_t)b * C2 * H2 * W2 + out_off;
        float4 pre[4];
        const bool fetch = bnext < a.B;
        if (fetch && vec_in) {
            const float4* xin4 = reinterpret_cast<const float4*>(a.in + (size_t)bnext * H * W + in_off);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx4 = tid + q * NTHR;
                if (idx4 < n_in / 4) pre[q] = xin4[idx4];
            }
        }
        if (!(a.dbg & 2)) {
            int t = t_begin;
            // Two waves of ONE workgroup on a SIMD (w and w + 4 when NW == 8) would otherwise run in lock-step: MFMA
            // phases overlapping (each at half rate) and epilogue/prologue gaps overlapping too (pipe idle).  The upper
            // half starts with one single tile, which puts it half an iteration out of phase so one wave's gaps fall
            // under the other's MFMAs.  (With NW == 4 the SIMD's second wave belongs to another workgroup.)
            if (NW > 4 && wave >= NW / 2 && t < t_end) {
                conv2_tiles<C1, ACT, false>(A1, lane_off, P1, Wp1, nX, t, breg, bias2, al2, be2, a.al2 != nullptr, outb, i, hi, H2, W2);
                t += 1;
            }
            for (; t + 1 < t_end; t += 2)
                conv2_tiles<C1, ACT, true>(A1, lane_off, P1, Wp1, nX, t, breg, bias2, al2, be2, a.al2 != nullptr, outb, i, hi, H2, W2);
            if (t < t_end)
                conv2_tiles<C1, ACT, false>(A1, lane_off, P1, Wp1, nX, t, breg, bias2, al2, be2, a.al2 != nullptr, outb, i, hi, H2, W2);
        }
        if (fetch) {
            if (vec_in) store_plane_regs(pre);
            else load_plane_sync(a.in + (size_t)bnext * H * W + in_off);
        }
        __syncthreads();                                     // A1 is free for the next clip's P1, In holds the next clip
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Standalone MFMA conv for the third conv stage of the E2E mel-CNN (32->64, un-pooled; architectures.py:851-853) and
// of CRNN (32->32, pooled; :217-225): same tile routine as conv2 above with K = 32*9 (144 B fragments per wave).
// One workgroup per clip: the [32][H][W] input is staged into a zero-haloed LDS plane set with coalesced loads; the
// 8 waves split into Cout/32 groups (one 32-channel N tile each) and share the tiles of their group.
size_t conv_mfma_lds_bytes(int C1, int H, int W) { return ((size_t)C1 * (H + 3) * (W + 2) + 64) * sizeof(float); }

template <int C1, int ACT, bool POOL, int NW, bool AVG = false>
__global__ void __launch_bounds__(64 * NW, NW / 4) conv3x3_mfma_kernel(ConvMfmaArgs a) {
    constexpr int NTHR = 64 * NW, KS = C1 * 9 / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, W = a.W, Wp = W + 2, P = (H + 3) * Wp;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;
    const int ngroups = a.Cout / 32, gsz = NW / ngroups;           // waves per 32-channel group
    const int grp = wave / gsz, wg = wave - grp * gsz;
    for (int k = tid; k < C1 * P + 64; k += NTHR) lds[k] = 0.0f;
    float breg[KS];
    const int cout = 32 * grp + i;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c2 = s / 9, tap = s - c2 * 9;
        breg[s] = a.w[((size_t)cout * C1 + 2 * c2 + hi) * 9 + tap];
    }
    const float bias = a.bias ? a.bias[cout] : 0.0f;
    const float al = a.alpha ? a.alpha[cout] : 1.0f, be = a.alpha ? a.beta[cout] : 0.0f;
    const bool bn = a.alpha != nullptr;
    const int nRp = POOL ? H / 2 : (H + 1) / 2, nX = (W + 15) / 16, nT = nRp * nX;
    const int t_base = nT / gsz, t_rem = nT - t_base * gsz;
    const int t_begin = wg * t_base + min(wg, t_rem), t_end = t_begin + t_base + (wg < t_rem ? 1 : 0);
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const int lane_off = hi * P + dyi * Wp + xi;
    __syncthreads();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float* xin = a.in + (size_t)b * C1 * H * W;
        for (int idx = tid; idx < C1 * H * W; idx += NTHR) {       // coalesced along x; interior at (+1,+1)
            const int c = idx / (H * W), r = idx - c * H * W, y = r / W, x = r - y * W;
            lds[c * P + (y + 1) * Wp + x + 1] = xin[idx];
        }
        __syncthreads();
        float* outb = a.out + ((size_t)b * a.Cout + 32 * grp) * Ho * Wo;
        float wsum[4] = {0.f, 0.f, 0.f, 0.f};
        const AvgWin aw{a.avg_kw, a.avg_sw, a.avg_ow};
        int t = t_begin;
        for (; t + 1 < t_end; t += 2)
            conv2_tiles<C1, ACT, true, POOL, AVG>(lds, lane_off, P, Wp, nX, t, breg, bias, al, be, bn, outb, i, hi, Ho, Wo, 0, wsum, aw);
        if (t < t_end)
            conv2_tiles<C1, ACT, false, POOL, AVG>(lds, lane_off, P, Wp, nX, t, breg, bias, al, be, bn, outb, i, hi, Ho, Wo, 0, wsum, aw);
        __syncthreads();                                       // every wave is done reading the input planes
        if (AVG) {
            // fixed-order reduction: part[wave][lane][4] in LDS (the input region is free now), then one lane per
            // (channel, window) adds the 2*gsz partials of its channel in ascending (wave, half) order
            float* part = lds;
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(wave * 64 + lane) * 4 + j] = wsum[j];
            __syncthreads();
            const float inv = 1.0f / (float)(H * a.avg_kw);
            for (int o = tid; o < a.Cout * a.avg_ow; o += NTHR) {
                const int co = o / a.avg_ow, j = o - co * a.avg_ow;
                const int g = co / 32, ci = co - 32 * g;
                float sum = 0.0f;
                for (int w2 = 0; w2 < gsz; ++w2)
                    for (int h2 = 0; h2 < 2; ++h2) sum += part[((g * gsz + w2) * 64 + h2 * 32 + ci) * 4 + j];
                a.out[((size_t)b * a.Cout + co) * a.avg_ow + j] = sum * inv;
            }
            __syncthreads();
            for (int k = tid; k < NW * 64 * 4; k += NTHR) part[k] = 0.0f;     // restore the zero halo cells we overwrote
            __syncthreads();
        }
    }
}

hipError_t launch_conv3x3_mfma(const ConvMfmaArgs& a, int C1, int max_grid, hipStream_t s) {
    if (C1 != 32 || a.Cout % 32 != 0 || (8 % (a.Cout / 32)) != 0) return hipErrorInvalidValue;
    const size_t lds = conv_mfma_lds_bytes(C1, a.H, a.W);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int grid = a.B < max_grid ? a.B : max_grid;
    if (grid < 1) grid = 1;
    if (a.avg_ow > 0 && (a.pool || a.avg_ow > 4 || (size_t)8 * 64 * 4 * sizeof(float) > lds)) return hipErrorInvalidValue;
#define CM_LAUNCH(ACTV, POOLV)                                                                                         \
    {                                                                                                                  \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<32, ACTV, POOLV, 8>), lds);      \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((conv3x3_mfma_kernel<32, ACTV, POOLV, 8>), dim3(grid), dim3(512), lds, s, a);               \
    }
#define CM_LAUNCH_AVG(ACTV)                                                                                            \
    {                                                                                                                  \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<32, ACTV, false, 8, true>), lds); \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((conv3x3_mfma_kernel<32, ACTV, false, 8, true>), dim3(grid), dim3(512), lds, s, a);         \
    }
#define CM_ACT(ACTV)                                         \
    if (a.avg_ow > 0) CM_LAUNCH_AVG(ACTV) else if (a.pool) CM_LAUNCH(ACTV, true) else CM_LAUNCH(ACTV, false)
    switch (a.act) {
        case ACT_RELU: CM_ACT(ACT_RELU) break;
        case ACT_GELU: CM_ACT(ACT_GELU) break;
        case ACT_SILU: CM_ACT(ACT_SILU) break;
        default: return hipErrorInvalidValue;
    }
#undef CM_ACT
#undef CM_LAUNCH
#undef CM_LAUNCH_AVG
    return hipGetLastError();
}

hipError_t launch_cnn_trunk(const TrunkArgs& a, int C1, int C2, int max_grid, hipStream_t s) {
    if (C1 != 16 || C2 != 32) return hipErrorInvalidValue;
#ifdef NWW_ABLATION      // stage-skipping builds for phase timing (results are garbage): never in the shipped library
    static const int dbg = [] { const char* e = getenv("NWW_TRUNK_DBG"); return e ? atoi(e) : 0; }();
#else
    const int dbg = 0;
#endif
    static const int force_strips = 0;
    TrunkArgs aa = a;
    aa.dbg = dbg;
    int per_cu = 0;
    int S = trunk_pick_strips(C1, a.H, a.W, &per_cu);
    if (force_strips > 0 && force_strips <= a.H / 4) {
        S = force_strips;
        const size_t l = trunk_lds_bytes(C1, a.H, a.W, S);
        per_cu = l <= 80 * 1024 ? 2 : (l <= 160 * 1024 ? 1 : 0);
    }
    if (S < 1 || per_cu < 1) return hipErrorInvalidValue;
    aa.strips = S;
    const size_t lds = trunk_lds_bytes(C1, a.H, a.W, S);
    // two workgroups per CU: 4 waves each (one per SIMD, 256 VGPRs); a lone workgroup: 8 waves (two per SIMD)
    int nw = per_cu == 2 ? 4 : 8;
    long want = (long)a.B * S, cap = (long)max_grid * per_cu;
    int grid = (int)(want < cap ? want : cap);
    grid -= grid % S;
    if (grid < S) grid = S;
#define TRUNK_LAUNCH2(ACTV, NWV, STRIPV)                                                                           \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_kernel<16, 32, ACTV, NWV, STRIPV>), lds);  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((cnn_trunk_kernel<16, 32, ACTV, NWV, STRIPV>), dim3(grid), dim3(64 * NWV), lds, s, aa); \
    }
#define TRUNK_LAUNCH1(ACTV, NWV)                                                                                   \
    if (S == 1) TRUNK_LAUNCH2(ACTV, NWV, false) else TRUNK_LAUNCH2(ACTV, NWV, true)
#define TRUNK_LAUNCH(ACTV)                                                                                         \
    if (nw == 4) TRUNK_LAUNCH1(ACTV, 4) else TRUNK_LAUNCH1(ACTV, 8)
    switch (a.act) {
        case ACT_RELU: TRUNK_LAUNCH(ACT_RELU) break;
        case ACT_GELU: TRUNK_LAUNCH(ACT_GELU) break;
        case ACT_SILU: TRUNK_LAUNCH(ACT_SILU) break;
        default: return hipErrorInvalidValue;
    }
#undef TRUNK_LAUNCH2
#undef TRUNK_LAUNCH1
#undef TRUNK_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// First conv of BcResNet (architectures.py:655-660): Conv2d(1, 32, 3, p1, bias=False) + BN + act + MaxPool2, written
// channels-last [B][H/2][W/2][32] for the depthwise / 1x1 stages.  Same MFMA scheme as the trunk's conv1 - tile = 16
// conv pixels (2 rows x 8 columns, a lane's 4 accumulator registers are one pooling window) x 16 channels on
// v_mfma_f32_16x16x4_f32, K = 9 taps padded to 12 - with two channel blocks per tile; the pooled values go straight to
// HBM (lanes = channels -> 64-byte runs).  One workgroup walks clips; the next clip's plane is prefetched into
// registers while the current one is convolved.  (The VALU version of this stage took 0.95 ms per 8192 clips.)
template <int ACT, int NW>
__global__ void __launch_bounds__(64 * NW, 2) conv1_pool_nhwc_mfma_kernel(Conv1NhwcArgs a) {
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2, Wp0 = W + 2;
    const int in_f = ((H + 2) * Wp0 + 3) & ~3;
    float* In = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < in_f; k += NTHR) In[k] = 0.0f;
    const int i1 = lane & 15, g1 = lane >> 4;
    float wreg[2][3];
    int tap_off[3];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int tap = 4 * st + g1;
        wreg[0][st] = tap < 9 ? a.w[(size_t)i1 * 9 + tap] : 0.0f;
        wreg[1][st] = tap < 9 ? a.w[(size_t)(16 + i1) * 9 + tap] : 0.0f;
        tap_off[st] = tap < 9 ? (tap / 3) * Wp0 + (tap % 3) : 0;
    }
    float bias[2], al[2], be[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        bias[cb] = a.bias ? a.bias[16 * cb + i1] : 0.0f;
        al[cb] = a.alpha ? a.alpha[16 * cb + i1] : 1.0f;
        be[cb] = a.alpha ? a.beta[16 * cb + i1] : 0.0f;
    }
    const bool bn = a.alpha != nullptr;
    const int nX1 = (2 * W1 + 7) / 8, ngx = (nX1 + 3) / 4, nG = H1 * ngx;
    const int pix_off = ((i1 >> 1) & 1) * Wp0 + 2 * (i1 >> 2) + (i1 & 1);
    const int dR = NW / ngx, dX = NW - dR * ngx;
    const int R_first = wave / ngx, X_first = wave - R_first * ngx;
    const bool vec_in = (W & 3) == 0 && H * W <= 16 * NTHR;
    auto load_sync = [&](const float* xin) {
        for (int idx = tid; idx < H * W; idx += NTHR) {
            const int y = idx / W, x = idx - y * W;
            In[(y + 1) * Wp0 + x + 1] = xin[idx];
        }
    };
    __syncthreads();
    if ((int)blockIdx.x < a.B) load_sync(a.in + (size_t)blockIdx.x * H * W);
    __syncthreads();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int bnext = b + gridDim.x;
        float4 pre[4];
        const bool fetch = bnext < a.B;
        if (fetch && vec_in) {
            const float4* xin4 = reinterpret_cast<const float4*>(a.in + (size_t)bnext * H * W);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx4 = tid + q * NTHR;
                if (idx4 < H * W / 4) pre[q] = xin4[idx4];
            }
        }
        float* outb = a.out + (size_t)b * H1 * W1 * 32;
        int R = R_first, X = X_first;
        for (int g = wave; g < nG; g += NW) {
            const int X0 = 4 * X;
            const float* rowp = In + (2 * R) * Wp0 + 8 * X0 + pix_off;
            f32x4 acc[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[cb][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const float* q = rowp + tap_off[st];
                float av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = q[8 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], wreg[0][st], acc[0][u], 0, 0, 0);
                    acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], wreg[1][st], acc[1][u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = 4 * (X0 + u) + g1;
                if (X0 + u < nX1 && col < W1) {
                    float* dst = outb + ((size_t)R * W1 + col) * 32 + i1;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        float m = -INFINITY;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v = acc[cb][u][q] + bias[cb];
                            if (bn) v = v * al[cb] + be[cb];
                            m = fmaxf(m, trunk_act<ACT>(v));
                        }
                        dst[16 * cb] = m;
                    }
                }
            }
            R += dR; X += dX;
            if (X >= ngx) { X -= ngx; ++R; }
        }
        __syncthreads();                                       // everyone is done with this clip's plane
        if (fetch) {
            if (vec_in) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx4 = tid + q * NTHR;
                    if (idx4 < H * W / 4) {
                        const int idx = idx4 * 4, y = idx / W, x = idx - y * W;
                        float* d = In + (y + 1) * Wp0 + x + 1;
                        d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
                    }
                }
            } else {
                load_sync(a.in + (size_t)bnext * H * W);
            }
        }
        __syncthreads();
    }
}

// (Round 3, measured and not kept: this kernel with its convolution as the transposed split-operand bf16 product of trunk_b.hip
// - 12 bf16 MFMAs per 32 pooled pixels and 16 channels, pooling in registers, float4 stores into P, four waves x two
// workgroups per CU at 238 registers: correct (|dlogit| 2.6e-6) and SLOWER, 1.06 vs 0.93 ms per 8192 clips - the stage is
// bound by its depthwise / staging phases, which want this version's sixteen waves per CU, not by the f32 MFMAs.)
// The same first conv fused with the FIRST BLOCK'S depthwise 3x3 (stride sh x sw, padding 1; architectures.py:632-647):
// the pooled 32-channel planes - 205 KB per clip, 1.7 GB per 8192 clips written and read back when the two stages are
// separate launches - stay in LDS, one strip of rows at a time, and only the depthwise output d [B][Ho][Wo][32] and the
// strided centres xs [B][Ho][Wo][32] (the shortcut's input) reach HBM.  Strip s covers depthwise rows [s * rows_dw, ...)
// and needs the conv rows sh * oy - 1 .. sh * oy + 1 of those; a workgroup takes whole clips (plane staged once, the next
// clip's plane prefetched into registers), conv -> barrier -> depthwise -> barrier per strip.  Depthwise taps in the
// order and fmaf chain of dwconv3x3_nhwc_kernel.
template <int ACT, int NW>
__global__ void __launch_bounds__(64 * NW) conv1_pool_dw_nhwc_kernel(Conv1DwArgs a) {
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2, Wp0 = W + 2;
    const int in_f = ((H + 2) * Wp0 + 3) & ~3;
    float* Wd = lds;                                           // depthwise weights, tap-major [9][32]
    float* In = lds + 288;
    float* P = In + in_f;                                      // [strip conv rows][W1][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < in_f; k += NTHR) In[k] = 0.0f;
    const int i1 = lane & 15, g1 = lane >> 4;
    float wreg[2][3];
    int tap_off[3];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int tap = 4 * st + g1;
        wreg[0][st] = tap < 9 ? a.w[(size_t)i1 * 9 + tap] : 0.0f;
        wreg[1][st] = tap < 9 ? a.w[(size_t)(16 + i1) * 9 + tap] : 0.0f;
        tap_off[st] = tap < 9 ? (tap / 3) * Wp0 + (tap % 3) : 0;
    }
    float bias[2], al[2], be[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        bias[cb] = a.bias ? a.bias[16 * cb + i1] : 0.0f;
        al[cb] = a.alpha ? a.alpha[16 * cb + i1] : 1.0f;
        be[cb] = a.alpha ? a.beta[16 * cb + i1] : 0.0f;
    }
    const bool bn = a.alpha != nullptr;
    const int nX1 = (2 * W1 + 7) / 8, ngx = (nX1 + 3) / 4;
    const int pix_off = ((i1 >> 1) & 1) * Wp0 + 2 * (i1 >> 2) + (i1 & 1);
    const int Ho = a.Ho, Wo = a.Wo, sh = a.sh, sw = a.sw;
    // depthwise: thread -> (four channels 4 cq .. 4 cq + 3, output slot); P and the weights (LDS: 36 registers more would cost
    // the second workgroup of the CU) are read 16 bytes at a time - 18 LDS reads and one index computation per four outputs
    // instead of 9 and one per output
    const int cq = tid & 7, dslot = tid >> 3;                  // NTHR / 8 output slots
    for (int k = tid; k < 288; k += NTHR) Wd[k] = a.dw_wt[k];
    const bool vec_in = (W & 3) == 0 && H * W <= 16 * NTHR;
    auto load_sync = [&](const float* xin) {
        for (int idx = tid; idx < H * W; idx += NTHR) {
            const int y = idx / W, x = idx - y * W;
            In[(y + 1) * Wp0 + x + 1] = xin[idx];
        }
    };
    __syncthreads();
    if ((int)blockIdx.x < a.B) load_sync(a.in + (size_t)blockIdx.x * H * W);
    __syncthreads();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int bnext = b + gridDim.x;
        const bool fetch = bnext < a.B;
        for (int oy0 = 0; oy0 < Ho; oy0 += a.rows_dw) {
            const int oy1 = min(Ho, oy0 + a.rows_dw);                              // depthwise rows [oy0, oy1)
            // the next clip's plane travels in registers during the LAST strip's convolution only (after it nobody reads In):
            // sixteen registers live through the depthwise phase would cost the CU's second workgroup
            float4 pre[4];
            const bool prefetch = fetch && vec_in && oy1 >= Ho;
            if (prefetch) {
                const float4* xin4 = reinterpret_cast<const float4*>(a.in + (size_t)bnext * H * W);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx4 = tid + q * NTHR;
                    if (idx4 < H * W / 4) pre[q] = xin4[idx4];
                }
            }
            const int r_lo = max(0, sh * oy0 - 1), r_hi = min(H1 - 1, sh * (oy1 - 1) + 1);   // conv (pooled) rows kept in P
            // ---- conv + BN + act + pool of rows r_lo .. r_hi -> P
            const int nG = (r_hi - r_lo + 1) * ngx;
            for (int g = wave; g < nG; g += NW) {
                const int Rl = g / ngx, X = g - Rl * ngx, R = r_lo + Rl;
                const int X0 = 4 * X;
                const float* rowp = In + (2 * R) * Wp0 + 8 * X0 + pix_off;
                f32x4 acc[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[cb][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    const float* q = rowp + tap_off[st];
                    float av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = q[8 * u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], wreg[0][st], acc[0][u], 0, 0, 0);
                        acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], wreg[1][st], acc[1][u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int col = 4 * (X0 + u) + g1;
                    if (X0 + u < nX1 && col < W1) {
                        float* dst = P + ((size_t)Rl * W1 + col) * 32 + i1;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            float m;
                            if (ACT == ACT_RELU) {
                                // bias + BN affine + ReLU is monotone in the accumulator (rising for alpha >= 0, falling below): the
                                // window's max or min goes through it ONCE - 8 instead of 16 operations, bit-identical (trunk_b.hip)
                                const f32x4 c4 = acc[cb][u];
                                const float mx = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
                                const float mn = fminf(fminf(c4[0], c4[1]), fminf(c4[2], c4[3]));
                                float v = ((bn && al[cb] < 0.0f) ? mn : mx) + bias[cb];
                                if (bn) v = v * al[cb] + be[cb];
                                m = fmaxf(v, 0.0f);
                            } else {
                                m = -INFINITY;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    float v = acc[cb][u][q] + bias[cb];
                                    if (bn) v = v * al[cb] + be[cb];
                                    m = fmaxf(m, trunk_act<ACT>(v));
                                }
                            }
                            dst[16 * cb] = m;
                        }
                    }
                }
            }
            __syncthreads();
            if (prefetch) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx4 = tid + q * NTHR;
                    if (idx4 < H * W / 4) {
                        const int idx = idx4 * 4, y = idx / W, x = idx - y * W;
                        float* d = In + (y + 1) * Wp0 + x + 1;
                        d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
                    }
                }
            }
            // ---- depthwise 3x3 of the strip's rows out of P
            for (int o = dslot; o < (oy1 - oy0) * Wo; o += NTHR / 8) {
                const int oyl = o / Wo, ox = o - oyl * Wo, oy = oy0 + oyl;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), centre = acc;
#pragma unroll 1
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = oy * sh - 1 + dy;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xx = ox * sw - 1 + dx;
                        const bool ok = yy >= 0 && yy < H1 && xx >= 0 && xx < W1;
                        const float4 v = ok ? *reinterpret_cast<const float4*>(P + ((size_t)(yy - r_lo) * W1 + xx) * 32 + 4 * cq)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (dy == 1 && dx == 1) centre = v;
                        const float4 w = *reinterpret_cast<const float4*>(Wd + (dy * 3 + dx) * 32 + 4 * cq);
                        acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
                        acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
                    }
                }
                const size_t oi = ((size_t)b * Ho * Wo + (size_t)oy * Wo + ox) * 32 + 4 * cq;
                if (a.bf16_out) {                             // wave-uniform: 16-bit activations (split_h2.h)
                    const int k16 = a.bf16_out;
                    const uint2 pd = make_uint2(nww_pk_act16(k16, acc.x, acc.y, a.d_scale), nww_pk_act16(k16, acc.z, acc.w, a.d_scale));
                    const uint2 px = make_uint2(nww_pk_act16(k16, centre.x, centre.y, a.xs_scale), nww_pk_act16(k16, centre.z, centre.w, a.xs_scale));
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.d_out) + oi) = pd;
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.xs_out) + oi) = px;
                } else {
                    *reinterpret_cast<float4*>(a.d_out + oi) = acc;
                    *reinterpret_cast<float4*>(a.xs_out + oi) = centre;
                }
            }
            __syncthreads();
        }
        if (fetch && !vec_in) load_sync(a.in + (size_t)bnext * H * W);
        __syncthreads();
    }
}

// LDS budget of a workgroup: 80 KB = smaller strips (six depthwise rows), TWO workgroups per CU - one's MFMA / epilogue phases
// under the other's depthwise phase: 1.14 -> 0.95 ms at 8192 clips (160 KB = one per CU: 1.14; 100: 1.19, 56: 1.04)
static int front_lds_kb() { return 80; }
// depthwise rows per strip such that input plane + strip planes fit LDS; 0 = does not fit
int conv1_pool_dw_rows(int H, int W, int sh) {
    const int H1 = H / 2, W1 = W / 2, Ho = (H1 - 1) / sh + 1;
    const size_t in_b = ((((size_t)(H + 2) * (W + 2) + 3) & ~(size_t)3) + 16 + 288) * sizeof(float);
    if (H < 4 || W < 4) return 0;
    for (int strips = 1; strips <= Ho; ++strips) {
        const int rows = (Ho + strips - 1) / strips;
        const size_t conv_rows = (size_t)sh * (rows - 1) + 3;
        if (in_b + conv_rows * W1 * 32 * sizeof(float) <= (size_t)front_lds_kb() * 1024) return rows;
    }
    return 0;
}

hipError_t launch_conv1_pool_dw_nhwc(const Conv1DwArgs& a0, int max_grid, hipStream_t s) {
    Conv1DwArgs a = a0;
    a.rows_dw = conv1_pool_dw_rows(a.H, a.W, a.sh);
    if (a.rows_dw <= 0) return hipErrorInvalidValue;
    const int H1 = a.H / 2, W1 = a.W / 2;
    a.Ho = (H1 - 1) / a.sh + 1; a.Wo = (W1 - 1) / a.sw + 1;
    const size_t in_f = (((size_t)(a.H + 2) * (a.W + 2) + 3) & ~(size_t)3);
    const size_t lds = (288 + in_f + ((size_t)a.sh * (a.rows_dw - 1) + 3) * W1 * 32 + 16) * sizeof(float);
    const int per_cu = front_lds_kb() <= 80 ? 2 : 1;
    int grid = a.B < max_grid * per_cu ? a.B : max_grid * per_cu;
    if (grid < 1) grid = 1;
#define C1DW_GO(ACTV)                                                                                              \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv1_pool_dw_nhwc_kernel<ACTV, 8>), lds);      \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((conv1_pool_dw_nhwc_kernel<ACTV, 8>), dim3(grid), dim3(512), lds, s, a);                \
    }
    switch (a.act) {
        case ACT_RELU: C1DW_GO(ACT_RELU) break;
        case ACT_GELU: C1DW_GO(ACT_GELU) break;
        case ACT_SILU: C1DW_GO(ACT_SILU) break;
        default: return hipErrorInvalidValue;
    }
#undef C1DW_GO
    return hipGetLastError();
}

bool conv1_pool_nhwc_mfma_fits(int H, int W) { return H >= 4 && W >= 4 && (size_t)(H + 2) * (W + 2) * 4 + 64 <= 64 * 1024; }

hipError_t launch_conv1_pool_nhwc_mfma(const Conv1NhwcArgs& a, int max_grid, hipStream_t s) {
    if (!conv1_pool_nhwc_mfma_fits(a.H, a.W)) return hipErrorInvalidValue;
    const size_t lds = ((((size_t)(a.H + 2) * (a.W + 2) + 3) & ~(size_t)3) + 16) * sizeof(float);
    int grid = a.B < 2 * max_grid ? a.B : 2 * max_grid;        // 27 KB of LDS, 8 waves: two workgroups per CU
    if (grid < 1) grid = 1;
    switch (a.act) {
        case ACT_RELU: hipLaunchKernelGGL((conv1_pool_nhwc_mfma_kernel<ACT_RELU, 8>), dim3(grid), dim3(512), lds, s, a); break;
        case ACT_GELU: hipLaunchKernelGGL((conv1_pool_nhwc_mfma_kernel<ACT_GELU, 8>), dim3(grid), dim3(512), lds, s, a); break;
        case ACT_SILU: hipLaunchKernelGGL((conv1_pool_nhwc_mfma_kernel<ACT_SILU, 8>), dim3(grid), dim3(512), lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
