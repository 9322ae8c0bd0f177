// trunk_x3.hip - the fused conv trunk (see trunk.hip) with conv2 on the bf16 matrix cores by exact operand splitting.
//
//     x[H][W] -> Conv2d(1,16,3,p1) (+BN) + act + MaxPool2 -> Conv2d(16,32,3,p1) (+BN) + act + MaxPool2 -> [32][H/4][W/4]
//
// conv2 is 89 % of the trunk's flops and v_mfma_f32_32x32x2_f32 tops out at 157 TF; v_mfma_f32_32x32x16_bf16 runs 16x
// faster (measured 2365 vs 150 TFLOP/s, tools/ubench/mfma_rate.hip).  Every float32 value v is hi + mid + lo, three
// bf16 numbers holding its 24 significant bits exactly, so a float32 product is the sum of nine bf16 x bf16 products,
// each exact in float32, accumulated in float32 by the MFMA:
//     PRODUCTS = 9 : all nine - exact products, float32 accumulation (only the order of the float32 additions differs
//                    from an fmaf chain); 16/9 = 1.8x the float32 MFMA rate
//     PRODUCTS = 6 : hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the dropped terms are < 2^-23 of the product;
//                    16/6 = 2.7x
//
// Data flow per workgroup (8 waves) and work item (clip, row strip):
//   P0  input rows -> LDS (float32, zero halo), next item's rows prefetched into registers during conv2
//   P1  conv1 on v_mfma_f32_16x16x4_f32 exactly as in trunk.hip; bias/BN/act/2x2 max in the C layout; each pooled value
//       is split and stored channels-last: A1[row][col][term][16 ch] bf16 = 96 B per pixel, zero halo
//   P2  conv2 as 9 taps x (K = 16 input channels = one bf16 MFMA): per tap the lane's pixel contributes three 16-byte
//       fragments (ds_read_b128: 8 channels of one term), the weights of that tap are three register-resident
//       fragments; PRODUCTS MFMAs per tap and tile; two tiles in flight.  Epilogue as in trunk.hip.
// A1 for a whole (101,64) clip would be 170 KB, so clips are cut into row strips (trunk_strip): 2 strips -> 106 KB.
// Tried and rejected (measured, 6 products): conv1 and conv2 on different waves of the SIMD with double-buffered A1 -
// 0.53 vs 0.48 ms, four waves of either kind are too few to hide their own latencies; issuing the next conv1 group's
// MFMAs before the current group's pool/split epilogue, with the conv2 weight fragments parked in LDS to make room -
// 0.57 ms, hipcc answers with 65 spilled VGPRs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "layers.h"
#include "trunk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int PS = 96;                  // bytes per A1 pixel: 3 terms x 16 channels x bf16
constexpr int C1 = 16, C2 = 32;

template <int ACT>
__device__ __forceinline__ float x3_trunk_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}
// v -> three float32 bit patterns whose upper 16 bits are the bf16 terms (lo has at most 8 significant bits left)
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// products of term ta of the activation with term tb of the weight, smallest first
template <int PRODUCTS>
__device__ __forceinline__ void tap_mfma(const bf16x8 (&a)[3], const bf16x8* w, f32x16& acc) {
    if (PRODUCTS == 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[2], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[0], acc, 0, 0, 0);
}

// bias/BN/act of the four values of a pooling window, then their maximum.  Without BN and with ReLU the maximum
// commutes with the (monotone) bias add and ReLU, bit for bit: 5 operations instead of 11.
template <int ACT, bool BN>
__device__ __forceinline__ float pool_quad(float v0, float v1, float v2, float v3, float bias, float al, float be) {
    if (ACT == ACT_RELU && !BN) return fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) + bias, 0.0f);
    float m = -INFINITY;
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t = v[q] + bias;
        if (BN) t = t * al + be;
        m = fmaxf(m, x3_trunk_act<ACT>(t));
    }
    return m;
}

// conv2 for tile t (and t+1 when TWO): 32 pixels (2 rows x 16 columns) x 32 output channels
// wl != nullptr: the weight fragments come from LDS (wl = the lane's slot of the [27][64] x 16-byte table) instead of bw
template <int ACT, int PRODUCTS, bool BN, bool TWO>
__device__ __forceinline__ void conv2_tiles_x3(const unsigned char* A1, int lane_off, int rowB, int nX, int t,
                                               const bf16x8 (&bw)[27], float bias2, float al2, float be2,
                                               float* outb, int i, int hi, int H2, int W2, int blk_kt, int blk_k0,
                                               const unsigned char* wl = nullptr) {
    const int R0 = t / nX, X0 = t - R0 * nX;
    const int t1 = TWO ? t + 1 : t;
    const int R1 = t1 / nX, X1 = t1 - R1 * nX;
    const unsigned char* pa = A1 + lane_off + (2 * R0) * rowB + 16 * X0 * PS;
    const unsigned char* pb = A1 + lane_off + (2 * R1) * rowB + 16 * X1 * PS;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    bf16x8 na[3], nb[3], nw[3];
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
        na[tm] = *reinterpret_cast<const bf16x8*>(pa + 32 * tm);
        if (TWO) nb[tm] = *reinterpret_cast<const bf16x8*>(pb + 32 * tm);
        if (wl) nw[tm] = *reinterpret_cast<const bf16x8*>(wl + tm * 1024);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        bf16x8 ca[3], cb[3], cw[3];
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) { ca[tm] = na[tm]; if (TWO) cb[tm] = nb[tm]; if (wl) cw[tm] = nw[tm]; }
        if (tap + 1 < 9) {
            const int off = ((tap + 1) / 3) * rowB + ((tap + 1) % 3) * PS;
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                na[tm] = *reinterpret_cast<const bf16x8*>(pa + off + 32 * tm);
                if (TWO) nb[tm] = *reinterpret_cast<const bf16x8*>(pb + off + 32 * tm);
                if (wl) nw[tm] = *reinterpret_cast<const bf16x8*>(wl + (3 * (tap + 1) + tm) * 1024);
            }
        }
        // next tap's LDS reads stay ABOVE this tap's MFMAs (hipcc otherwise sinks them to their first use)
        __builtin_amdgcn_sched_barrier(0);
        tap_mfma<PRODUCTS>(ca, wl ? cw : &bw[3 * tap], acc0);
        if (TWO) tap_mfma<PRODUCTS>(cb, wl ? cw : &bw[3 * tap], acc1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int which = 0; which < (TWO ? 2 : 1); ++which) {
        const f32x16& acc = which ? acc1 : acc0;
        const int R = which ? R1 : R0, X = which ? X1 : X0;
        float own[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)                      // pooled column 8X + 2k + hi
            own[k] = pool_quad<ACT, BN>(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3], bias2, al2, be2);
        // half 0 keeps columns 0..3 of the 8-column segment, half 1 keeps 4..7
        const float s0 = hi ? own[0] : own[2], s1 = hi ? own[1] : own[3];
        const float r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
        float4 o;
        if (hi == 0) { o.x = own[0]; o.y = r0; o.z = own[1]; o.w = r1; }
        else         { o.x = r0; o.y = own[2]; o.z = r1; o.w = own[3]; }
        const int pcol = 8 * X + 4 * hi;
        float* dst = outb + ((size_t)i * H2 + R) * W2 + pcol;
        if (blk_kt) {      // blocked: outb = the clip's row inside its 128-row block, k = feature index of (channel, row, column)
            const int k = i * H2 * W2 + blk_k0 + R * W2 + pcol;
            dst = outb + (size_t)(k >> 5) * (128 * 32) + (k & 31);
        }
        if ((W2 & 3) == 0 && pcol + 3 < W2) {
            *reinterpret_cast<float4*>(dst) = o;
        } else {
            if (pcol + 0 < W2) dst[0] = o.x;
            if (pcol + 1 < W2) dst[1] = o.y;
            if (pcol + 2 < W2) dst[2] = o.z;
            if (pcol + 3 < W2) dst[3] = o.w;
        }
    }
}

// BN: either conv carries a folded BatchNorm (a missing one is alpha = 1, beta = 0, which is exact).
// A workgroup keeps ONE strip index for its whole life, so the zero halos written once stay valid.  (Walking whole clips
// strip by strip instead - equal work per workgroup, halo rows re-zeroed per item - measured 0.417 vs 0.406 ms.)
template <int ACT, int PRODUCTS, bool BN, int NW, bool V1 = false, bool WLDS = false>
__device__ __forceinline__ void cnn_trunk_x3_body(const TrunkArgs& a) {
    // conv2's bf16 MFMAs overlap with the SIMD's other wave's VALU / LDS work only when their accumulators live in
    // AGPRs (tools/ubench/mfma_valu_overlap.hip: max(a, b) instead of a + b).  hipcc picks the VGPR form for kernels
    // bounded to <= 256 registers unless something mentions an AGPR, and once AGPRs are in play it splits a 256-register
    // budget 128 / 128 - too few VGPRs for the 108 registers of conv2 weight fragments (it then parks them in AGPRs and
    // copies them back before every MFMA: measured 0.46 vs 0.39 ms).  The 4-wave form has a 512-register budget, so
    // 224 VGPRs + 32 accumulator AGPRs = 256 registers, two workgroups per CU, works: the empty asm flips the form.
    // WLDS (8-wave shape): conv2's weight fragments live in LDS (27 KB) instead of 108 VGPRs, which leaves room for the AGPR
    // form inside the 128 + 128 split of a 256-register budget
    if constexpr (NW == 4 || WLDS) { float agpr_hint = 0.0f; asm volatile("; mfma accumulators in AGPRs" : "+a"(agpr_hint)); }
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
    const int S = a.strips;
    const int Wp0 = W + 2, Wp1 = W1 + 2, rowB = Wp1 * PS;
    int n_a1, a1_shift, nR2, n_in, row_shift, a1_bytes, in_f;
    size_t in_off, out_off;
    {
        // strip of this workgroup (kept for life).  Default: blockIdx % S.  With two strips of unequal cost (25 pooled rows =
        // 13 + 12: 26 conv2 tiles take the 8 waves four rounds, 24 take three) the first a.n0 workgroups take strip 0 and
        // the rest strip 1, each group walking all clips - more CUs on the dearer strip instead of idle ones at the end.
        const int sidx = (a.n0 > 0 && S == 2) ? ((int)blockIdx.x < a.n0 ? 0 : 1) : (int)blockIdx.x % S;
        const TrunkStrip sg = trunk_strip(H, S, sidx);
        n_a1 = sg.a1_hi - sg.a1_lo + 1;
        a1_shift = sg.a1_lo - sg.a1_base;
        nR2 = sg.R2b - sg.R2a;
        n_in = (sg.y_hi - sg.y_lo + 1) * W;
        row_shift = sg.y_lo - sg.iy0;
        a1_bytes = sg.a1_rows * rowB;
        in_f = (sg.in_rows * Wp0 + 3) & ~3;
        in_off = (size_t)sg.y_lo * W;
        out_off = (size_t)sg.R2a * W2;
    }
    float* In = reinterpret_cast<float*>(lds_raw);
    unsigned char* A1 = lds_raw + (size_t)in_f * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;

    // zero both LDS regions once: halos stay zero, interiors are rewritten per item
    for (int k = tid; k < in_f + a1_bytes / 4 + 16; k += NTHR) reinterpret_cast<uint32_t*>(lds_raw)[k] = 0u;

    // conv2 weights -> B fragments: tap, term: lane (cout = i, channels 8*hi .. 8*hi+7), split once per workgroup
    bf16x8 bw[27];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        uint32_t th[8], tm[8], tl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(a.w2[((size_t)i * C1 + 8 * hi + j) * 9 + tap], th[j], tm[j], tl[j]);
        const uint4 vh = make_uint4(pack_hi16(th[0], th[1]), pack_hi16(th[2], th[3]), pack_hi16(th[4], th[5]), pack_hi16(th[6], th[7]));
        const uint4 vm = make_uint4(pack_hi16(tm[0], tm[1]), pack_hi16(tm[2], tm[3]), pack_hi16(tm[4], tm[5]), pack_hi16(tm[6], tm[7]));
        const uint4 vl = make_uint4(pack_hi16(tl[0], tl[1]), pack_hi16(tl[2], tl[3]), pack_hi16(tl[4], tl[5]), pack_hi16(tl[6], tl[7]));
        if constexpr (WLDS) {                                  // straight to the LDS table (wave 0; identical in every wave)
            if (wave == 0) {
                unsigned char* wt0 = lds_raw + (size_t)in_f * 4 + (((size_t)a1_bytes + 64 + 15) & ~(size_t)15) + lane * 16;
                *reinterpret_cast<uint4*>(wt0 + (3 * tap + 0) * 1024) = vh;
                *reinterpret_cast<uint4*>(wt0 + (3 * tap + 1) * 1024) = vm;
                *reinterpret_cast<uint4*>(wt0 + (3 * tap + 2) * 1024) = vl;
            }
        } else {
            bw[3 * tap + 0] = __builtin_bit_cast(bf16x8, vh);
            bw[3 * tap + 1] = __builtin_bit_cast(bf16x8, vm);
            bw[3 * tap + 2] = __builtin_bit_cast(bf16x8, vl);
        }
    }
    const float bias2 = a.b2 ? a.b2[i] : 0.0f;
    const float al2 = a.al2 ? a.al2[i] : 1.0f, be2 = a.al2 ? a.be2[i] : 0.0f;
    // WLDS: wave 0 parks the fragments (identical in every wave) behind A1; everyone reads them back per tap
    unsigned char* Wt = A1 + (((size_t)a1_bytes + 64 + 15) & ~(size_t)15);
    const unsigned char* wl = WLDS ? Wt + lane * 16 : nullptr;

    // conv1 weights -> B fragments of the 16x16x4 MFMA: lane (g = l>>4, channel = l&15), step st: tap 4*st + g
    float w1reg[3];
    int tap_off1[3];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int tap = 4 * st + (lane >> 4);
        w1reg[st] = tap < 9 ? a.w1[(size_t)(lane & 15) * 9 + tap] : 0.0f;
        tap_off1[st] = tap < 9 ? (tap / 3) * Wp0 + (tap % 3) : 0;
    }
    const float bias1 = a.b1 ? a.b1[lane & 15] : 0.0f;
    const float al1 = a.al1 ? a.al1[lane & 15] : 1.0f, be1 = a.al1 ? a.be1[lane & 15] : 0.0f;

    // conv2 tiling of this strip (tile rows are local pooled rows)
    const int nX = (W1 + 15) / 16, nT = nR2 * nX;
    const int t_base = nT / NW, t_rem = nT - t_base * NW;
    const int t_begin = wave * t_base + min(wave, t_rem), t_end = t_begin + t_base + (wave < t_rem ? 1 : 0);
    // lane's pixel inside a tile: i = 4*quad + 2*dy + dx (quad along x); its 8 channels of a term start at 16*hi bytes
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const int lane_off = (dyi * Wp1 + xi) * PS + 16 * hi;

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

    // conv1 building blocks: a group = 4 horizontally adjacent tiles of 2 x 8 conv1 pixels (16 x 16 x 12 MFMA each)
    const int nX1 = (2 * W1 + 7) / 8, ngx = (nX1 + 3) / 4, nG = n_a1 * ngx;
    const int i1 = lane & 15, g1 = lane >> 4;
    const int pix_off = ((i1 >> 1) & 1) * Wp0 + 2 * (i1 >> 2) + (i1 & 1);
    unsigned char* a1lane = A1 + (a1_shift * Wp1 + g1 + 1) * PS + 2 * i1;              // channel i1 of term 0
    auto conv1_mfma = [&](int R, int X0, f32x4 (&acc)[4]) {                              // R = A1 row - a1_lo
        const float* rowp = In + (2 * R) * Wp0 + 8 * X0 + pix_off;
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
    };
    // FULL: every tile of the group lies inside the row (W1 a multiple of 16) - no per-store predicate
    auto conv1_store = [&](int R, int X0, const f32x4 (&acc)[4], auto full) {
        constexpr bool FULL = decltype(full)::value;
        unsigned char* wr = a1lane + (R * Wp1 + 4 * X0) * PS;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float m = pool_quad<ACT, BN>(acc[u][0], acc[u][1], acc[u][2], acc[u][3], bias1, al1, be1);
            uint32_t vh, vm, vl;
            split3(m, vh, vm, vl);
            if (!(a.dbg & 4) && (FULL || (X0 + u < nX1 && 4 * (X0 + u) + g1 < W1))) {
                unsigned char* wp = wr + 4 * u * PS;
                *reinterpret_cast<uint16_t*>(wp) = (uint16_t)(vh >> 16);
                *reinterpret_cast<uint16_t*>(wp + 32) = (uint16_t)(vm >> 16);
                *reinterpret_cast<uint16_t*>(wp + 64) = (uint16_t)(vl >> 16);
            }
        }
    };
    // group g -> (R, X) with X = X0/4, walked incrementally (g += NW) instead of dividing per group
    const int dR = NW / ngx, dX = NW - dR * ngx;
    const int R_first = wave / ngx, X_first = wave - R_first * ngx;
    const bool full1 = (W1 & 15) == 0;

    const bool uneven = a.n0 > 0 && S == 2;
    const int b0 = uneven ? ((int)blockIdx.x < a.n0 ? (int)blockIdx.x : (int)blockIdx.x - a.n0) : (int)blockIdx.x / S;
    const int bstep = uneven ? ((int)blockIdx.x < a.n0 ? a.n0 : (int)gridDim.x - a.n0) : (int)gridDim.x / S;
    if (a.skew > 0) {
        // two 4-wave workgroups per CU run identical items and would stay in the same phase (conv1: VALU/latency-heavy,
        // conv2: matrix-pipe-bound) forever; delay the one that got the SIMD's second wave slot by part of an item once
        const int slot = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 1;
        for (int k = 0; k < slot * a.skew; ++k) __builtin_amdgcn_s_sleep(127);
    }
    __syncthreads();
    if (b0 < a.B) load_plane_sync(a.in + (size_t)b0 * H * W + in_off);
    __syncthreads();
    for (int b = b0; b < a.B; b += bstep) {
        const int bnext = b + bstep;
        // ---------------- P1: conv1 + act + pool, split into bf16 terms -> A1 (channels last)
        if constexpr (V1) {
            // conv1 on the VALU ALONE (no float32 MFMA, which blocks every other wave's VALU and cannot run beside the
            // other workgroup's bf16 MFMAs): lane = pooled pixel, its 4 x 4 input patch in registers (eight 8-byte LDS
            // reads), the channel's nine weights and bias / BN come in as scalar operands (uniform s_loads), four conv
            // pixels x nine fmaf per channel, pool, three-way split; the pixel's 16 channels leave as six 16-byte LDS
            // stores (the MFMA form writes 48 two-byte pieces).  With two 4-wave workgroups per CU in opposite phases
            // this phase runs under the other workgroup's conv2 MFMAs (AGPR accumulators: DESIGN.md 4.2a).
            const int npx = n_a1 * W1;
            for (int p0 = wave * 64; p0 < npx && !(a.dbg & 1); p0 += 64 * NW) {
                const int p = p0 + lane;
                const bool ok = p < npx;
                const int pp = ok ? p : npx - 1;
                const int R = pp / W1, x = pp - R * W1;
                const float* base = In + (2 * R) * Wp0 + 2 * x;
                float pt[4][4];
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const float2 lo2 = *reinterpret_cast<const float2*>(base + dy * Wp0);
                    const float2 hi2 = *reinterpret_cast<const float2*>(base + dy * Wp0 + 2);
                    pt[dy][0] = lo2.x; pt[dy][1] = lo2.y; pt[dy][2] = hi2.x; pt[dy][3] = hi2.y;
                }
                uint32_t ph[C1 / 2], pm[C1 / 2], pl[C1 / 2];      // channel pairs packed as they are produced (register budget)
#pragma unroll
                for (int c2 = 0; c2 < C1 / 2; ++c2) {
                    uint32_t th[2], tm[2], tl[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = 2 * c2 + e;
                        float acc[4];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) {
                            const float w = a.w1[c * 9 + tap];
                            const int ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float v = pt[(q >> 1) + ty][(q & 1) + tx];
                                acc[q] = tap == 0 ? v * w : fmaf(v, w, acc[q]);
                            }
                        }
                        const float m = pool_quad<ACT, BN>(acc[0], acc[1], acc[2], acc[3], a.b1 ? a.b1[c] : 0.0f,
                                                           a.al1 ? a.al1[c] : 1.0f, a.al1 ? a.be1[c] : 0.0f);
                        split3(m, th[e], tm[e], tl[e]);
                    }
                    ph[c2] = pack_hi16(th[0], th[1]); pm[c2] = pack_hi16(tm[0], tm[1]); pl[c2] = pack_hi16(tl[0], tl[1]);
                    asm volatile("" : "+v"(ph[c2]), "+v"(pm[c2]), "+v"(pl[c2]));      // keeps the pairs from being computed all at once
                }
                if (ok) {
                    unsigned char* wp = A1 + ((a1_shift + R) * Wp1 + x + 1) * PS;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int o = 4 * half;
                        *reinterpret_cast<uint4*>(wp + 16 * half) = make_uint4(ph[o], ph[o + 1], ph[o + 2], ph[o + 3]);
                        *reinterpret_cast<uint4*>(wp + 32 + 16 * half) = make_uint4(pm[o], pm[o + 1], pm[o + 2], pm[o + 3]);
                        *reinterpret_cast<uint4*>(wp + 64 + 16 * half) = make_uint4(pl[o], pl[o + 1], pl[o + 2], pl[o + 3]);
                    }
                }
            }
        } else if (!(a.dbg & 1)) {
            f32x4 acc[4];
            int R = R_first, X = X_first;
            for (int g = wave; g < nG; g += NW) {
                conv1_mfma(R, 4 * X, acc);
                if (full1) conv1_store(R, 4 * X, acc, std::true_type{});
                else conv1_store(R, 4 * X, acc, std::false_type{});
                R += dR; X += dX;
                if (X >= ngx) { X -= ngx; ++R; }
            }
        }
        __syncthreads();
        // ---------------- P2: conv2 on the bf16 MFMA; tiles in pairs, a lone tile alone
        float* outb = a.out + (size_t)b * C2 * H2 * W2 + out_off;
        const int blk_kt = a.out_blocked, blk_k0 = (int)out_off;
        if (blk_kt) outb = a.out + ((size_t)(b >> 7) * blk_kt * 128 + (b & 127)) * 32;
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
            if (wave >= NW / 2 && t < t_end) {           // out of phase with the SIMD's other wave (see trunk.hip)
                conv2_tiles_x3<ACT, PRODUCTS, BN, false>(A1, lane_off, rowB, nX, t, bw, bias2, al2, be2, outb, i, hi, H2, W2, blk_kt, blk_k0, wl);
                t += 1;
            }
            if (NW == 8)                                     // the 4-wave shape has no registers for a second tile in flight
                for (; t + 1 < t_end; t += 2)
                    conv2_tiles_x3<ACT, PRODUCTS, BN, true>(A1, lane_off, rowB, nX, t, bw, bias2, al2, be2, outb, i, hi, H2, W2, blk_kt, blk_k0, wl);
            else
                for (; t + 1 < t_end; t += 1)
                    conv2_tiles_x3<ACT, PRODUCTS, BN, false>(A1, lane_off, rowB, nX, t, bw, bias2, al2, be2, outb, i, hi, H2, W2, blk_kt, blk_k0, wl);
            if (t < t_end)
                conv2_tiles_x3<ACT, PRODUCTS, BN, false>(A1, lane_off, rowB, nX, t, bw, bias2, al2, be2, outb, i, hi, H2, W2, blk_kt, blk_k0, wl);
        }
        if (fetch) {
            if (vec_in) store_plane_regs(pre);
            else load_plane_sync(a.in + (size_t)bnext * H * W + in_off);
        }
        __syncthreads();                                     // A1 is free for the next item's P1, In holds its rows
    }
}

// Two launch shapes of the same body (an attribute cannot depend on a template parameter):
//   8 waves, one workgroup per CU, 256 registers per lane, MFMAs in the VGPR form
//   4 waves, two workgroups per CU (three row strips), 224 VGPRs + AGPR accumulators
template <int ACT, int PRODUCTS, bool BN, int NW>
__global__ void __launch_bounds__(512, 2) cnn_trunk_x3_kernel(TrunkArgs a) {
    static_assert(NW == 8, "8-wave shape");
    cnn_trunk_x3_body<ACT, PRODUCTS, BN, 8>(a);
}
template <int ACT, int PRODUCTS, bool BN>
__global__ void __launch_bounds__(512, 2) cnn_trunk_x3_kernel_wlds(TrunkArgs a) {
    cnn_trunk_x3_body<ACT, PRODUCTS, BN, 8, false, true>(a);
}
template <int ACT, int PRODUCTS, bool BN, bool V1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(224))) cnn_trunk_x3_kernel4(TrunkArgs a) {
    cnn_trunk_x3_body<ACT, PRODUCTS, BN, 4, V1>(a);
}
}  // namespace

size_t trunk_x3_lds_bytes(int H, int W, int S) {
    const int W1 = W / 2;
    size_t worst = 0;
    for (int s = 0; s < S; ++s) {
        const TrunkStrip g = trunk_strip(H, S, s);
        const size_t in_b = (((size_t)g.in_rows * (W + 2) + 3) & ~(size_t)3) * 4;
        const size_t a1_b = (size_t)g.a1_rows * (W1 + 2) * PS + 64;
        if (in_b + a1_b > worst) worst = in_b + a1_b;
    }
    return worst;
}
int trunk_x3_pick_strips(int H, int W) {
    const int H2 = H / 4;
    for (int S = 1; S <= H2; ++S)
        if (trunk_x3_lds_bytes(H, W, S) <= 160 * 1024) return S;
    return 0;
}

hipError_t launch_cnn_trunk_x3(const TrunkArgs& a, int products, int max_grid, hipStream_t s) {
    static const int dbg = [] { const char* e = getenv("NWW_TRUNK_DBG"); return e ? atoi(e) : 0; }();
    static const int force_strips = [] { const char* e = getenv("NWW_TRUNK_STRIPS"); return e ? atoi(e) : 0; }();
    static const int skew = [] { const char* e = getenv("NWW_X3_SKEW"); return e ? atoi(e) : 0; }();
    TrunkArgs aa = a;
    aa.dbg = dbg;
    aa.skew = skew;
    int S = trunk_x3_pick_strips(a.H, a.W);
    if (force_strips > S && force_strips <= a.H / 4) S = force_strips;
    if (S < 1) return hipErrorInvalidValue;
    // a handful of clips (the interpreter's B = 1 .. 16 calls) would occupy a handful of CUs for a whole clip each: cut
    // every clip into four row strips on four CUs instead (23 -> 9 us at B = 1).  Seam rows are recomputed by both
    // neighbours with the same arithmetic, so the result does not depend on the strip count (bit for bit).
    static const int small_strips = [] { const char* e = getenv("NWW_X3_SMALL_STRIPS"); return e ? atoi(e) : 4; }();
    if (!force_strips && small_strips > S && (long)a.B * small_strips * 4 <= max_grid && small_strips <= a.H / 4) S = small_strips;
    aa.strips = S;
    const size_t lds = trunk_x3_lds_bytes(a.H, a.W, S);
    // experiments: NWW_X3_WAVES=4 runs 4-wave workgroups, two per CU when the strip fits 80 KB
    static const int force_nw = [] { const char* e = getenv("NWW_X3_WAVES"); return e ? atoi(e) : 0; }();
    const int nw = force_nw == 4 ? 4 : 8;
    // NWW_X3_V1=1 (with NWW_X3_WAVES=4): conv1 on the VALU alone.  Parity-green and measured NEUTRAL (0.429 vs 0.425 ms; phase
    // ablation: conv1 0.17 + conv2 0.20 + skeleton 0.03 add up exactly, with either conv1): the two workgroups of a CU
    // stay in phase, and even in perfect anti-phase a lone VALU wave per SIMD issues at half the two-wave rate.
    static const int valu_conv1 = [] { const char* e = getenv("NWW_X3_V1"); return e ? atoi(e) : 0; }();
    // NWW_X3_WLDS=1: conv2 weight fragments in LDS + AGPR accumulators in the 8-wave shape; parity-green, measured slower
    // (0.403 vs 0.383 ms), kept for A/B
    static const int wlds_env = [] { const char* e = getenv("NWW_X3_WLDS"); return e ? atoi(e) : 0; }();
    const bool wlds = wlds_env && nw == 8 && lds + 27 * 1024 + 96 <= 160 * 1024;
    const int per_cu = (nw == 4 && lds <= 80 * 1024) ? 2 : 1;
    long want = (long)a.B * S, cap = (long)max_grid * per_cu;
    int grid = (int)(want < cap ? want : cap);
    grid -= grid % S;
    if (grid < S) grid = S;
    // uneven split of the workgroups over two strips (see the kernel): share of strip 0 in 1/256ths, when the grid is full
    // Measured on the (101,64) clip (13 + 12 pooled rows), trunk ms for strip 0's share 128 / 130 / 132 / 134 / 136 / 142 of
    // 256: 0.436 / 0.433 / 0.430 / 0.423 / 0.432 / 0.449 against 0.446 with alternating strips.  NWW_X3_N0 overrides
    // (-1 = alternate).
    static const int n0_env = [] { const char* e = getenv("NWW_X3_N0"); return e ? atoi(e) : 0; }();
    int n0_share = n0_env;
    if (n0_env == 0 && S == 2) {
        const TrunkStrip s0 = trunk_strip(a.H, 2, 0), s1 = trunk_strip(a.H, 2, 1);
        if (s0.R2b - s0.R2a > s1.R2b - s1.R2a) n0_share = 134;     // the odd pooled row makes strip 0 ~10 % dearer
    }
    aa.n0 = 0;
    if (S == 2 && n0_share > 0 && n0_share < 256 && nw == 8 && grid >= 64 && (long)a.B * 2 >= 2L * grid) {
        aa.n0 = (int)((long)grid * n0_share / 256);
        if (aa.n0 < 1 || aa.n0 >= grid) aa.n0 = 0;
    }
    const bool bn = a.al1 != nullptr || a.al2 != nullptr;
#define X3T_LAUNCH(ACTV, PRODV, BNV, NWV)                                                                          \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_x3_kernel<ACTV, PRODV, BNV, NWV>), lds);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((cnn_trunk_x3_kernel<ACTV, PRODV, BNV, NWV>), dim3(grid), dim3(64 * NWV), lds, s, aa);  \
    }
#define X3T_LAUNCH4V(ACTV, PRODV, BNV, V1V)                                                                        \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_x3_kernel4<ACTV, PRODV, BNV, V1V>), lds);  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((cnn_trunk_x3_kernel4<ACTV, PRODV, BNV, V1V>), dim3(grid), dim3(256), lds, s, aa);      \
    }
#define X3T_LAUNCH4(ACTV, PRODV, BNV)                                                                              \
    if (valu_conv1) X3T_LAUNCH4V(ACTV, PRODV, BNV, true) else X3T_LAUNCH4V(ACTV, PRODV, BNV, false)
#define X3T_LAUNCHW(ACTV, PRODV, BNV)                                                                              \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_x3_kernel_wlds<ACTV, PRODV, BNV>), lds + 27 * 1024 + 96);  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((cnn_trunk_x3_kernel_wlds<ACTV, PRODV, BNV>), dim3(grid), dim3(512), lds + 27 * 1024 + 96, s, aa);   \
    }
#define X3T_NW(ACTV, PRODV, BNV)                                                                                   \
    if (nw == 4) X3T_LAUNCH4(ACTV, PRODV, BNV) else if (wlds) X3T_LAUNCHW(ACTV, PRODV, BNV) else X3T_LAUNCH(ACTV, PRODV, BNV, 8)
#define X3T_BN(ACTV, PRODV)                                                                                        \
    if (bn) X3T_NW(ACTV, PRODV, true) else X3T_NW(ACTV, PRODV, false)
#define X3T_ACT(ACTV)                                                                                              \
    if (products == 6) X3T_BN(ACTV, 6) else X3T_BN(ACTV, 9)
    switch (a.act) {
        case ACT_RELU: X3T_ACT(ACT_RELU) break;
        case ACT_GELU: X3T_ACT(ACT_GELU) break;
        case ACT_SILU: X3T_ACT(ACT_SILU) break;
        default: return hipErrorInvalidValue;
    }
#undef X3T_LAUNCH
#undef X3T_LAUNCH4
#undef X3T_LAUNCH4V
#undef X3T_LAUNCHW
#undef X3T_NW
#undef X3T_BN
#undef X3T_ACT
    return hipGetLastError();
}
