// trunk_b.hip - the fused conv trunk with BOTH convolutions on the bf16 matrix cores (gfx950), software-pipelined.
//
//     x[H][W] -> Conv2d(1,16,3,p1) (+BN) + act + MaxPool2 -> Conv2d(16,32,3,p1) (+BN) + act + MaxPool2 -> [32][H/4][W/4]
//     (reference: CNNModel / CRNNModel / E2E_MelSpectrogram_CNN stems, nanowakeword/modules/architectures.py:60-71,217-225,840-852)
//
// Arithmetic: every float32 operand v is hi + mid + lo, three bf16 numbers holding its 24 significant bits exactly; a
// float32 product is the sum of nine bf16 x bf16 partial products, each exact in float32, accumulated in float32 by
// v_mfma_f32_32x32x16_bf16.  PRODUCTS = 9 issues all of them, PRODUCTS = 6 drops the three below 2^-23 of the product.
// PRODUCTS = 3 ("f16x3"): every operand v, scaled by a power of two fixed at plan time so that it cannot leave the binary16 range,
// is hi + lo with hi = RN16(v), lo = RN16(v - hi) - two binary16 numbers holding 22-23 of its 24 significant bits (the
// remainder v - hi has at most 12 significant bits left, lo keeps 11 of them: a value is off by at most 2^-23 of itself, and
// exact half of the time); the products hi*hi, hi*lo, lo*hi go to v_mfma_f32_32x32x16_f16 (each exact in float32, float32
// accumulation), lo*lo is < 2^-22 of the product.  Half the matrix instructions of the six-product form; measured against
// float64 the logits are as close as with the float32 MFMA (tools/x3_accuracy.py).  The scales: TrunkArgs::f16_*.
//
// conv1 is a TRANSPOSED split-operand product per pooled pixel:
//     M = 32 rows = (16 channels x 2 conv columns dx) for one conv row dy,   K = 16 = the pixel's 4 x 4 input patch,
//     N = 32 pooled pixels of one A1 row
// A = the conv1 weights scattered into the patch positions they touch (zeros elsewhere), B = the patch of the lane's
// pixel (two 8-byte LDS reads per term from bf16 input planes).  In the C layout a lane then holds, for ITS pixel,
// registers r = (channel 8 hi + r/2, dx = r & 1) of both conv rows: the 2 x 2 max-pool is three register maxima, and the
// eight pooled channels leave as ONE 16-byte LDS store per term - exactly the fragment conv2 reads back.
// conv2: tile = 32 pixels (2 rows x 16 columns, each 4-register group of the C layout a pooling window) x 32 output
// channels, 9 taps x (K = 16 input channels = one MFMA per product); accumulators in AGPRs, weight fragments resident
// in registers (24 of 27; the last tap's three come from LDS per tile).
//
// Schedule: an item is (clip, row strip); all eight waves run conv1 of the item (planes -> A1), a barrier, then its conv2
// tiles while the next item's rows travel from HBM into registers, a barrier.  The weight fragments of both convolutions
// are split and laid out ONCE at plan time (launch_trunk_b_pack; splitting them in every workgroup cost 20 us of a 380 us
// launch).  What bounds the kernel (tools/ubench/trunk_trace.hip): with all 256 CUs busy the chip is
// power-limited - the shader clock drops from 2.4 to 1.9-2.0 GHz while this kernel runs (2.37 GHz on 128 CUs, same clock
// COUNT per item) - and schedules with very different overlap (conv1 of item k + 1 on four waves beside conv2 of item k
// on the other four with double-buffered A1; the pooling epilogue of tile n inside tile n + 1's MFMA stream; wave
// priorities) all land within 2 % of this one in wall time even where they save 5-8 % of the clocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "split_h2.h"
#include "trunk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef NWW_TRACE      // phase stamps of the first items of the first and last eight workgroups (tools/ubench/trunk_trace.hip)
#define TB_SLOT ((int)blockIdx.x < 8 ? (int)blockIdx.x : (int)blockIdx.x >= (int)gridDim.x - 8 ? (int)blockIdx.x - ((int)gridDim.x - 16) : -1)
#define TB_STAMP(k)                                                                                                 \
    do {                                                                                                            \
        if (a.trace && TB_SLOT >= 0 && lane == 0 && item_no < 6)                                                    \
            a.trace[(((size_t)TB_SLOT * 6 + item_no) * NW + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#define TB_STAMP_W(k)                                                                                               \
    do {                                                                                                            \
        if (a.trace && TB_SLOT >= 0 && lane == 0 && item_no < 6)                                                    \
            a.trace[(((size_t)TB_SLOT * 6 + item_no) * NW + wave) * 8 + (k)] = wall_clock64();                      \
    } while (0)
#define TB_STAMP_WG(k)                                                                                              \
    do {                                                                                                            \
        if (a.trace && TB_SLOT >= 0 && threadIdx.x == 0) a.trace[16 * 6 * 8 * 8 + TB_SLOT * 4 + (k)] = wall_clock64(); \
    } while (0)
#else
#define TB_STAMP(k)
#define TB_STAMP_W(k)
#define TB_STAMP_WG(k)
#endif

#ifndef TB_ABL
#define TB_ABL 0      // ablation mask of tools/ubench/trunk_trace.hip builds: 1 conv2 without per-tap LDS reads, 2 no conv2 epilogue,
#endif                // 4 no conv1, 8 no conv2, 16 no conv1 epilogue (pool / split / A1 stores), 32 no input staging, 64 no input loads either
#ifdef NWW_TRACE      // tools/ubench/front_trace.hip: s_memtime of workgroup 0's eight waves at the phase boundaries of bc_front_b_kernel's strips
__device__ unsigned long long g_front_trace[8 * 16 * 8];
#define BF_STAMP(it, k) if (blockIdx.x == 0 && lane == 0 && (it) < 16) g_front_trace[(wave * 16 + (it)) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define BF_STAMP(it, k)
#endif

namespace {
constexpr int C1 = 16, C2 = 32;
constexpr int NW = 8;                   // waves per workgroup
constexpr int NTHR = 64 * NW;
// what depends on the arithmetic: terms per value, bytes per A1 pixel (terms x 16 channels x 2 bytes; the two-term form pads
// the pixel to 80 bytes: at 64 the sixteen pixels of a conv2 tile row would share four 16-byte bank slots), conv2 weight
// fragments fetched from LDS per tile (the last tap's terms) / resident in registers, packed fragments (conv2 (tap, term),
// then conv1 (dy, term); 1 KB each)
template <int PRODUCTS>
struct TbA {
    static constexpr bool F16 = PRODUCTS == 3;
    static constexpr int NT = F16 ? 2 : 3;
    static constexpr int PS = F16 ? 80 : 96;
    static constexpr int NF2 = 9 * NT, NF1 = 2 * NT;
    static constexpr int NWL = F16 ? 0 : 3;
    static constexpr int NWA = NF2 - NWL;
};
constexpr int NFRAG = 27 + 6;           // packed image size (the three-term form's; the two-term form uses 18 + 4 of them)

template <int ACT>
__device__ __forceinline__ float tb_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return nww_silu(v);
    return v;
}
// v -> three float32 bit patterns whose upper 16 bits are the bf16 terms (lo has at most 8 significant bits left)
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
// (a >> 16) | (b & 0xffff0000): bf16 of a in the low half, of b in the high half
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ bf16x8 frag4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
// pin a fragment into AGPRs: the compiler copies it there once and hands it to the MFMAs as an AGPR operand
__device__ __forceinline__ bf16x8 to_agpr(bf16x8 f) {
    u32x4 v = __builtin_bit_cast(u32x4, f);
    asm volatile("" : "+a"(v[0]), "+a"(v[1]), "+a"(v[2]), "+a"(v[3]));
    return __builtin_bit_cast(bf16x8, v);
}

// acc += the PRODUCTS largest partial products of x (MFMA A operand, terms hi/mid/lo) and y (B operand), smallest first
template <int PRODUCTS>
__device__ __forceinline__ void x3_mfma(const bf16x8* x, const bf16x8* y, f32x16& acc) {
    if (PRODUCTS == 3) {        // binary16 terms (fragments are carried as 128-bit bags typed bf16x8): lo*hi, hi*lo, hi*hi
        const f16x8 xh = __builtin_bit_cast(f16x8, x[0]), xl = __builtin_bit_cast(f16x8, x[1]);
        const f16x8 yh = __builtin_bit_cast(f16x8, y[0]), yl = __builtin_bit_cast(f16x8, y[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc, 0, 0, 0);
        return;
    }
    if (PRODUCTS == 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[2], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], acc, 0, 0, 0);
}

// bias/BN/act of the four values of a pooling window, then their maximum.  Without BN and with ReLU the maximum
// commutes with the (monotone) bias add and ReLU, bit for bit, and relu(m + b) = max(m, -b) + b exactly: two v_max3_f32
// and one add (the asm also keeps hipcc from canonicalising every MFMA output with a v_max x, x first).
// SC (two-term binary16 form): the accumulators are k times the true sums (k = input scale x weight scale) and the result
// leaves s times the true value (the next operand's scale); k, s powers of two, so every rounding below is the unscaled
// one's: the caller passes bias k, and
//   ReLU without BN: post = s / k                          ((u + bias k) s / k)
//   ReLU with BN:    al s / k, be s                        (nothing else changes)
//   other:           al / k (1 / k without BN), be, post = s   (the activation sees the true value)
// pos (wave-uniform, BN only): every folded-BN factor of the layer is >= 0 (checked at plan time) - the window's maximum alone decides
// NB: the layer has no bias (bias = 0 exactly): the add is left out
template <int ACT, bool BN, bool SC = false, bool NB = false>
__device__ __forceinline__ float pool_quad(float v0, float v1, float v2, float v3, float bias, float nbias, float al, float be, float post = 1.0f,
                                           bool pos = false) {
    if (ACT == ACT_RELU && !BN) {
        float t, u;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v0), "v"(v1), "v"(v2));
        asm("v_max3_f32 %0, %1, %2, -%3" : "=v"(u) : "v"(t), "v"(v3), "v"(bias));      // (-bias as a source modifier: no register, no v_xor for it)
        (void)nbias;
        // (u + bias) * post with post a power of two = fma(u, post, bias * post), bit for bit (one rounding either way, scaled exactly);
        // bias * post is loop-invariant: one instruction less per pooled value
        return SC ? fmaf(u, post, bias * post) : u + bias;
    }
    if (ACT == ACT_RELU && BN) {
        // v -> relu((v + bias) * al + be) is a chain of monotone roundings: non-decreasing for al >= 0, non-increasing for al < 0,
        // so its maximum over the window is its value at the window's maximum (minimum) - bit for bit, 8 operations instead of 16
        float t, mx, mn;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v0), "v"(v1), "v"(v2));
        asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(t), "v"(v3));
        if (pos) return fmaxf((NB ? mx : mx + bias) * al + be, 0.0f);
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v0), "v"(v1), "v"(v2));
        asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(t), "v"(v3));
        const float e = al < 0.0f ? mn : mx;
        return fmaxf((NB ? e : e + bias) * al + be, 0.0f);
    }
    // GELU / SiLU fall to their single minimum (x = -0.75 / -1.28) and rise after it, and bias + folded BN is monotone: over the
    // window, act(bn(v)) is largest at the window's LARGEST or SMALLEST v - two activations per pooled value instead of four
    // (the activation, 6-22 instructions, was 2/3 of the non-ReLU trunk: 0.68 ms against ReLU's 0.25)
    float t, mx, mn;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v0), "v"(v1), "v"(v2));
    asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(t), "v"(v3));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v0), "v"(v1), "v"(v2));
    asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(t), "v"(v3));
    float a = NB ? mx : mx + bias, b = NB ? mn : mn + bias;
    if (BN) { a = a * al + be; b = b * al + be; }
    const float m = fmaxf(tb_act<ACT>(a), tb_act<ACT>(b));
    return SC ? m * post : m;
}

// conv2 of one tile: 32 pixels (2 rows x 16 columns) x 32 output channels, 9 taps x (K = 16 input channels = one MFMA
// per product).  The lane's pixel contributes three 16-byte fragments per tap (8 channels of one term).  pa: the
// lane's pixel of the tile, dst: where its four pooled columns of output channel i go, nv: how many of them exist.
template <int ACT, int PRODUCTS, bool BN, bool POS = false>
__device__ __forceinline__ void conv2_tile(const unsigned char* pa, int rowB, const bf16x8 (&bw)[TbA<PRODUCTS>::NWA], const unsigned char* wl,
                                           float bias2, float nbias2, float al2, float be2, float post2, float* dst, int nv) {
    using AR = TbA<PRODUCTS>;
    constexpr int NT = AR::NT, PS = AR::PS, NWL = AR::NWL, NWA = AR::NWA;
    f32x16 acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
    bf16x8 na[NT], wlast[NWL > 0 ? NWL : 1];
#pragma unroll
    for (int tm = 0; tm < NT; ++tm) na[tm] = *reinterpret_cast<const bf16x8*>(pa + 32 * tm);
#pragma unroll
    for (int j = 0; j < NWL; ++j) wlast[j] = *reinterpret_cast<const bf16x8*>(wl + j * 1024);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        bf16x8 ca[NT];
#pragma unroll
        for (int tm = 0; tm < NT; ++tm) ca[tm] = na[tm];
        if (tap + 1 < 9 && !(TB_ABL & 1)) {
            const int off = ((tap + 1) / 3) * rowB + ((tap + 1) % 3) * PS;
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) na[tm] = *reinterpret_cast<const bf16x8*>(pa + off + 32 * tm);
        }
        // next tap's LDS reads stay ABOVE this tap's MFMAs (hipcc otherwise sinks them to their first use)
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8* w = NT * tap < NWA ? &bw[NT * tap] : &wlast[NT * tap - NWA];
        x3_mfma<PRODUCTS>(ca, w, acc0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (TB_ABL & 2) {
        asm volatile("" ::"a"(acc0));
        return;
    }
    float own[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)                      // pooled column 8X + 2k + hi
        own[k] = pool_quad<ACT, BN, AR::F16>(acc0[4 * k], acc0[4 * k + 1], acc0[4 * k + 2], acc0[4 * k + 3], bias2, nbias2, al2, be2, post2, POS);
    // half 0 keeps columns 0..3 of the 8-column segment, half 1 keeps 4..7: v_permlane32_swap hands the upper half of
    // its first operand to the lower half of the second and vice versa - after it both halves hold (x, y) and (z, w)
    const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[0]), __float_as_uint(own[2]), false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[1]), __float_as_uint(own[3]), false, false);
    float4 o;
    o.x = __uint_as_float(s02[0]); o.y = __uint_as_float(s02[1]); o.z = __uint_as_float(s13[0]); o.w = __uint_as_float(s13[1]);
    if (nv >= 4) {
        *reinterpret_cast<float4*>(dst) = o;
    } else {
        if (nv > 0) dst[0] = o.x;
        if (nv > 1) dst[1] = o.y;
        if (nv > 2) dst[2] = o.z;
    }
}

// The two-term form of conv2_tile.  A tap is three MFMAs (96 matrix-pipe clocks) here, less than an LDS read takes to come back
// with eight waves reading: fetched one tap ahead (conv2_tile) every tap waited for its fragments and a tile took 2.3 k clocks
// per wave for 864 clocks of MFMAs (tools/ubench/trunk_trace.hip).  The fragments travel THREE taps ahead in a ring of
// registers instead, across the tile boundary too: the last three taps of a tile request the first three of the wave's next
// tile (pa_next; null behind the last tile).  ring: taps 0..2 of this tile on entry, of the next tile on exit.
template <int ACT, bool BN, bool POS = false>
__device__ __forceinline__ void conv2_tile_h2(const unsigned char* pa, const unsigned char* pa_next, int rowB, const bf16x8 (&bw)[18],
                                              bf16x8 (&ring)[3][2], float bias2, float nbias2, float al2, float be2, float post2,
                                              float* dst, int nv) {
    constexpr int PS = TbA<3>::PS;
    f32x16 acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        bf16x8 ca[2] = {ring[tap % 3][0], ring[tap % 3][1]};
        if (!(TB_ABL & 1)) {
            const int nt = tap + 3 < 9 ? tap + 3 : tap + 3 - 9;
            const unsigned char* src = tap + 3 < 9 ? pa : pa_next;
            if (src) {
                const int off = (nt / 3) * rowB + (nt % 3) * PS;
                ring[tap % 3][0] = *reinterpret_cast<const bf16x8*>(src + off);
                ring[tap % 3][1] = *reinterpret_cast<const bf16x8*>(src + off + 32);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // the reads stay ABOVE this tap's MFMAs (hipcc otherwise sinks them to their first use)
        x3_mfma<3>(ca, &bw[2 * tap], acc0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (TB_ABL & 2) {
        asm volatile("" ::"a"(acc0));
        return;
    }
    float own[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        own[k] = pool_quad<ACT, BN, true>(acc0[4 * k], acc0[4 * k + 1], acc0[4 * k + 2], acc0[4 * k + 3], bias2, nbias2, al2, be2, post2, POS);
    const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[0]), __float_as_uint(own[2]), false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[1]), __float_as_uint(own[3]), false, false);
    float4 o;
    o.x = __uint_as_float(s02[0]); o.y = __uint_as_float(s02[1]); o.z = __uint_as_float(s13[0]); o.w = __uint_as_float(s13[1]);
    if (nv >= 4) {
        *reinterpret_cast<float4*>(dst) = o;
    } else {
        if (nv > 0) dst[0] = o.x;
        if (nv > 1) dst[1] = o.y;
        if (nv > 2) dst[2] = o.z;
    }
}

// Two tiles of one wave in flight: a wave with one accumulator chain leaves the matrix pipe to its partner (or idle) through its own
// epilogue and prologue - a lone wave takes 2.0 k clocks per tile for 864 clocks of MFMAs, and the SIMD's two waves own 3 + 4 tiles.
// Per tap the fragments of both tiles are fetched one tap ahead (six MFMAs cover the LDS round trip) and share the weight fragments.
template <int ACT, bool BN, bool POS = false>
__device__ __forceinline__ void conv2_pair_h2(const unsigned char* pa0, const unsigned char* pa1, int rowB, const bf16x8 (&bw)[18],
                                              float bias2, float nbias2, float al2, float be2, float post2,
                                              float* dst0, int nv0, float* dst1, int nv1) {
    constexpr int PS = TbA<3>::PS;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    bf16x8 n0[2], n1[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        n0[tm] = *reinterpret_cast<const bf16x8*>(pa0 + 32 * tm);
        n1[tm] = *reinterpret_cast<const bf16x8*>(pa1 + 32 * tm);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        bf16x8 c0[2] = {n0[0], n0[1]}, c1[2] = {n1[0], n1[1]};
        if (tap + 1 < 9) {
            const int off = ((tap + 1) / 3) * rowB + ((tap + 1) % 3) * PS;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                n0[tm] = *reinterpret_cast<const bf16x8*>(pa0 + off + 32 * tm);
                n1[tm] = *reinterpret_cast<const bf16x8*>(pa1 + off + 32 * tm);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        x3_mfma<3>(c0, &bw[2 * tap], acc0);
        x3_mfma<3>(c1, &bw[2 * tap], acc1);
        __builtin_amdgcn_sched_barrier(0);
    }
    auto finish = [&](const f32x16& acc, float* dst, int nv) {
        float own[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            own[k] = pool_quad<ACT, BN, true>(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3], bias2, nbias2, al2, be2, post2, POS);
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[0]), __float_as_uint(own[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[1]), __float_as_uint(own[3]), false, false);
        float4 o;
        o.x = __uint_as_float(s02[0]); o.y = __uint_as_float(s02[1]); o.z = __uint_as_float(s13[0]); o.w = __uint_as_float(s13[1]);
        if (nv >= 4) {
            *reinterpret_cast<float4*>(dst) = o;
        } else {
            if (nv > 0) dst[0] = o.x;
            if (nv > 1) dst[1] = o.y;
            if (nv > 2) dst[2] = o.z;
        }
    };
    finish(acc0, dst0, nv0);
    finish(acc1, dst1, nv1);
}

// LDS map (bytes): [3 input planes: in_rows x Wp0 bf16 each][A1: a1_rows x (Wp1 x 96 + 16) (+64 slack)][conv1
// weight fragments 6 KB][conv2 last-tap fragments 3 KB].  A workgroup keeps ONE strip index for its whole life, so the
// zero halos written once stay valid.  The 16 bytes of padding per A1 row put the two pixel rows of a
// conv2 tile on disjoint 16-byte bank slots (rows 3264 bytes apart land on the SAME eight slots of the 256-byte bank
// row: every ds_read_b128 of the round-2 layout was a two-way conflict).
// Two-term form (80-byte pixels): the eight x positions of a 16-lane read group land on eight distinct 16-byte slots whose
// complement is the same set shifted by eight slots, so the row pitch is 128 mod 256.
struct TbGeom { int Wp0, Wp1, rowB, plane_b, a1_b; };
__host__ __device__ inline TbGeom tb_geom(int W, const TrunkStrip& g, bool f16) {
    TbGeom r;
    r.Wp0 = (W + 5) & ~3;                                  // W + 2 columns (zero halo), a multiple of four: a row is whole 8-byte chunks of four columns
    r.Wp1 = W / 2 + 2;
    if (f16) {
        const int raw = r.Wp1 * 80;
        r.rowB = raw + ((128 - raw % 256) + 256) % 256;
    } else {
        r.rowB = r.Wp1 * 96 + 16;
    }
    r.plane_b = (g.in_rows * r.Wp0 * 2 + 15) & ~15;
    r.a1_b = (g.a1_rows * r.rowB + 64 + 15) & ~15;
    return r;
}
__host__ __device__ inline size_t tb_lds_total(const TbGeom& gg, bool f16) {
    return (f16 ? 2 : 3) * (size_t)gg.plane_b + (size_t)gg.a1_b + (f16 ? 4 : 6) * 1024 + (f16 ? 0 : 3) * 1024;
}

// Weight fragments, computed once per model: [conv2 (tap, term)][64 lanes] then [conv1 (dy, term)][64 lanes], 16 bytes
// per lane - the register images the kernel's MFMAs consume.
//   conv2: B fragment of tap, lane = (cout i = lane & 31, channels 8 hi .. 8 hi + 7)
//   conv1: A fragment of conv row dy: row m = 8 q + 4 h + 2 cj + dx of the product is (channel 8 h + 2 q + cj, conv column
//          dx) - so that C register r of lane half h is (channel 8 h + r / 2, dx = r & 1); k = 4 py + px is the patch
//          position, and the weight of tap (py - dy, px - dx) sits there (0 where the tap falls outside 3 x 3)
__global__ void __launch_bounds__(64) trunk_b_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, unsigned char* __restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    bf16x8* o = reinterpret_cast<bf16x8*>(out);
    for (int tap = 0; tap < 9; ++tap) {
        uint32_t th[8], tm[8], tl[8];
        for (int j = 0; j < 8; ++j) split3(w2[((size_t)i * C1 + 8 * hi + j) * 9 + tap], th[j], tm[j], tl[j]);
        o[(3 * tap + 0) * 64 + lane] = frag4(pack_hi16(th[0], th[1]), pack_hi16(th[2], th[3]), pack_hi16(th[4], th[5]), pack_hi16(th[6], th[7]));
        o[(3 * tap + 1) * 64 + lane] = frag4(pack_hi16(tm[0], tm[1]), pack_hi16(tm[2], tm[3]), pack_hi16(tm[4], tm[5]), pack_hi16(tm[6], tm[7]));
        o[(3 * tap + 2) * 64 + lane] = frag4(pack_hi16(tl[0], tl[1]), pack_hi16(tl[2], tl[3]), pack_hi16(tl[4], tl[5]), pack_hi16(tl[6], tl[7]));
    }
    const int m = i, cw = 8 * ((m >> 2) & 1) + 2 * (m >> 3) + ((m >> 1) & 1), dxw = m & 1;
    for (int dy = 0; dy < 2; ++dy) {
        uint32_t th[8], tm[8], tl[8];
        for (int kk = 0; kk < 8; ++kk) {
            const int ty = 2 * hi + (kk >> 2) - dy, tx = (kk & 3) - dxw;
            const bool in = ty >= 0 && ty < 3 && tx >= 0 && tx < 3;
            const float v = in ? w1[cw * 9 + ty * 3 + tx] : 0.0f;
            split3(v, th[kk], tm[kk], tl[kk]);
        }
        o[(27 + dy * 3 + 0) * 64 + lane] = frag4(pack_hi16(th[0], th[1]), pack_hi16(th[2], th[3]), pack_hi16(th[4], th[5]), pack_hi16(th[6], th[7]));
        o[(27 + dy * 3 + 1) * 64 + lane] = frag4(pack_hi16(tm[0], tm[1]), pack_hi16(tm[2], tm[3]), pack_hi16(tm[4], tm[5]), pack_hi16(tm[6], tm[7]));
        o[(27 + dy * 3 + 2) * 64 + lane] = frag4(pack_hi16(tl[0], tl[1]), pack_hi16(tl[2], tl[3]), pack_hi16(tl[4], tl[5]), pack_hi16(tl[6], tl[7]));
    }
}

// the two-term binary16 image: [conv2 (tap, term)] 18 fragments then [conv1 (dy, term)] 4, weights times sw2 / sw1 (powers of two)
__global__ void __launch_bounds__(64) trunk_b_pack_f16_kernel(const float* __restrict__ w1, const float* __restrict__ w2, unsigned char* __restrict__ out,
                                                              float sw1, float sw2) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    bf16x8* o = reinterpret_cast<bf16x8*>(out);
    for (int tap = 0; tap < 9; ++tap) {
        uint32_t th[4], tl[4];
        for (int j = 0; j < 4; ++j)
            nww_split2h(w2[((size_t)i * C1 + 8 * hi + 2 * j) * 9 + tap] * sw2, w2[((size_t)i * C1 + 8 * hi + 2 * j + 1) * 9 + tap] * sw2, th[j], tl[j]);
        o[(2 * tap + 0) * 64 + lane] = frag4(th[0], th[1], th[2], th[3]);
        o[(2 * tap + 1) * 64 + lane] = frag4(tl[0], tl[1], tl[2], tl[3]);
    }
    const int m = i, cw = 8 * ((m >> 2) & 1) + 2 * (m >> 3) + ((m >> 1) & 1), dxw = m & 1;
    for (int dy = 0; dy < 2; ++dy) {
        float v[8];
        for (int kk = 0; kk < 8; ++kk) {
            const int ty = 2 * hi + (kk >> 2) - dy, tx = (kk & 3) - dxw;
            const bool in = ty >= 0 && ty < 3 && tx >= 0 && tx < 3;
            v[kk] = in ? w1[cw * 9 + ty * 3 + tx] * sw1 : 0.0f;
        }
        uint32_t th[4], tl[4];
        for (int j = 0; j < 4; ++j) nww_split2h(v[2 * j], v[2 * j + 1], th[j], tl[j]);
        o[(18 + dy * 2 + 0) * 64 + lane] = frag4(th[0], th[1], th[2], th[3]);
        o[(18 + dy * 2 + 1) * 64 + lane] = frag4(tl[0], tl[1], tl[2], tl[3]);
    }
}

// Measured and not kept for the two-term form (round 4): conv2 of item k on waves 0-3 - one per SIMD, two tiles in flight - BESIDE
// conv1 of item k + 1 on waves 4-7, planes and A1 double-buffered in three strips, one barrier per item.  conv1's phase is VALU-bound
// (pooling, splitting: ~110 instructions per 6 MFMAs), conv2's MFMA-bound, and one after the other the matrix pipe is busy 45 % of
// an item - but beside a wave that issues MFMAs back to back the VALU wave ran 2.4 k clocks per group instead of 1.2 k and set the
// pace: 12.5 k clocks per third of a clip against 2 x 16.3 k per clip here (0.247 ms against 0.233; tools/ubench/trunk_trace.hip).
// With the roles swapped - conv1 on waves 0-3, the OLDER half that the SIMD's issue arbitration favours - 0.230 against 0.223, and
// s_setprio 2 on the conv1 waves changes nothing: one after the other or side by side, the launch takes about the SUM of its matrix
// time (0.10 ms at the clock it runs at) and its VALU time (0.09 ms).
// POS (BN + ReLU, two-term instances): every folded-BN factor of both layers is >= 0 (TrunkArgs::bn_pos, checked at plan time) - the pooled
// value is the window's maximum pushed through BN + ReLU, five operations instead of nine (pool_quad)
template <int ACT, int PRODUCTS, bool BN, bool POS = false>
__device__ __forceinline__ void cnn_trunk_b_body(const TrunkArgs& a) {
    using AR = TbA<PRODUCTS>;
    constexpr bool F16 = AR::F16;
    constexpr int NT = AR::NT, PS = AR::PS, NWA = AR::NWA, NWL = AR::NWL, NF2 = AR::NF2, NF1 = AR::NF1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    TB_STAMP_WG(0);
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
    const int S = a.strips;
    int n_a1, a1_shift, nR2, n_in, row_shift, Wp0, rowB, plane_b, a1_b, b0, bstep;
    size_t in_off, out_off;
    {
        // strip of this workgroup (kept for life) and the clips it walks
        int sidx;
        if (a.n_sub > 0) {                                  // explicit strips (streaming hop): equal shares of the grid
            sidx = (int)blockIdx.x % a.n_sub;
            b0 = (int)blockIdx.x / a.n_sub;
            bstep = (int)gridDim.x / a.n_sub;
        } else if (a.wg_end[0] > 0) {
            sidx = 0;
            while (sidx + 1 < S && (int)blockIdx.x >= a.wg_end[sidx]) ++sidx;
            const int first = sidx ? a.wg_end[sidx - 1] : 0;
            b0 = (int)blockIdx.x - first;
            bstep = a.wg_end[sidx] - first;
        } else {
            sidx = (int)blockIdx.x % S;
            b0 = (int)blockIdx.x / S;
            bstep = (int)gridDim.x / S;
        }
        const TrunkStrip sg = a.n_sub > 0 ? trunk_strip_rows(H, a.sub_a[sidx], a.sub_b[sidx]) : trunk_strip(H, S, sidx);
        const TbGeom gg = tb_geom(W, sg, F16);
        Wp0 = gg.Wp0; rowB = gg.rowB; plane_b = gg.plane_b; a1_b = gg.a1_b;
        n_a1 = sg.a1_hi - sg.a1_lo + 1;
        a1_shift = sg.a1_lo - sg.a1_base;
        nR2 = sg.R2b - sg.R2a;
        n_in = (sg.y_hi - sg.y_lo + 1) * W;
        row_shift = sg.y_lo - sg.iy0;
        in_off = (size_t)sg.y_lo * W;
        out_off = (size_t)sg.R2a * W2;
    }
    const size_t in_clip = a.in_clip_stride ? a.in_clip_stride : (size_t)H * W;
    const int ring = a.out_ring_rows;
    const int ring_r0 = ring ? (a.out_row0 + (int)(out_off / W2)) % ring : 0;      // ring row of the strip's first pooled row
    const int pitch0 = 2 * Wp0;
    unsigned char* const In3 = lds_raw;
    unsigned char* const A1 = lds_raw + NT * plane_b;
    unsigned char* const W1F = A1 + a1_b;
    unsigned char* const W2L = W1F + NF1 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;

    // zero the planes and A1 once: halos stay zero, interiors are rewritten per item
    for (int k = tid; k < (NT * plane_b + a1_b) / 4; k += NTHR) reinterpret_cast<uint32_t*>(lds_raw)[k] = 0u;

    // weight fragments: conv2's first NWA (24 of 27, or all 18) stay in registers, the rest and conv1's are parked in LDS by wave 0
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wpack) + lane;
    bf16x8 bw[NWA];
#pragma unroll
    for (int j = 0; j < NWA; ++j) bw[j] = wp[j * 64];
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NWL; ++j) *reinterpret_cast<bf16x8*>(W2L + j * 1024 + lane * 16) = wp[(NWA + j) * 64];
#pragma unroll
        for (int j = 0; j < NF1; ++j) *reinterpret_cast<bf16x8*>(W1F + j * 1024 + lane * 16) = wp[(NF2 + j) * 64];
    }
    // (measured for the two-term form: without any AGPR constraint hipcc allocates 183 VGPRs and VGPR-form MFMAs, whose operand
    // traffic starves the SIMD's other wave of VALU issue - 0.254 ms against 0.236 with the fragments pinned here)
#pragma unroll
    for (int j = 0; j < NWA; ++j) bw[j] = to_agpr(bw[j]);
    const unsigned char* wl = W2L + lane * 16;
    // epilogue constants; in the two-term form rescaled as pool_quad's comment says (k = accumulator scale, s = output scale)
    const float k1 = F16 ? a.f16_k1 : 1.0f, s1 = F16 ? a.f16_s1 : 1.0f, k2 = F16 ? a.f16_k2 : 1.0f, s2 = F16 ? a.f16_so : 1.0f;
    constexpr bool RELU = ACT == ACT_RELU;
    const float post1 = RELU ? s1 / k1 : s1, post2 = RELU ? s2 / k2 : s2;
    float bias2 = a.b2 ? a.b2[i] : 0.0f;
    float al2 = (BN && a.al2) ? a.al2[i] : 1.0f, be2 = (BN && a.al2) ? a.be2[i] : 0.0f;
    if (F16) {
        bias2 *= k2;
        if (RELU) { al2 *= s2 / k2; be2 *= s2; } else al2 /= k2;
    }
    float b1v[8], nb1v[8], al1v[8], be1v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        b1v[cc] = a.b1 ? a.b1[8 * hi + cc] : 0.0f;
        al1v[cc] = (BN && a.al1) ? a.al1[8 * hi + cc] : 1.0f;
        be1v[cc] = (BN && a.al1) ? a.be1[8 * hi + cc] : 0.0f;
        if (F16) {
            b1v[cc] *= k1;
            if (RELU) { al1v[cc] *= s1 / k1; be1v[cc] *= s1; } else al1v[cc] /= k1;
        }
        nb1v[cc] = -b1v[cc];
    }

    // conv2 tiling of this strip (tile rows are local pooled rows).  SIMD s (waves s and s + 4) owns tiles s, s + 4, ...
    const int nX = (W1 + 15) / 16, nT = nR2 * nX;
    const int simd = wave & 3;
    const int T_s = (nT - simd + 3) / 4, tA = T_s / 2, tB = T_s - tA;      // waves s + 4 / s take the first tB / the other tA of them
    // lane's pixel inside a tile: i = 4*quad + 2*dy + dx (quad along x); its 8 channels of a term start at 16*hi bytes
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const int a1_lane = dyi * rowB + xi * PS + 16 * hi;
    // output: lane = (channel i, pooled columns 8 X + 4 hi .. + 3)
    const int blk_kt = a.out_blocked;
    const int out_lane = ring ? i * (int)a.out_ch_stride + 4 * hi : i * H2 * W2 + 4 * hi;

    // input rows -> the NT planes of 16-bit terms (zero halo): value (y, x) at column x + 1 of local row y + row_shift.
    // A thread moves CHUNKS of four plane columns 4 j .. 4 j + 3 = inputs x = 4 j - 1 .. 4 j + 2 (W / 4 + 1 chunks per row): one
    // 16-byte load (dword aligned) and one aligned 8-byte LDS store per term.  Chunks of four INPUTS land two bytes off an
    // 8-byte boundary, and those misaligned LDS stores cost 3.7 k of an item's 19 k clocks (tools/ubench/trunk_trace.hip, -DTB_ABL=32).
    // (two-term form: the input is clamped to +-f16_clamp, the bound the plan-time scales were derived from - whatever comes in,
    // no term can leave the binary16 range.  Log-mel dB values never reach it.)
    const float s_in = F16 ? a.f16_in : 1.0f, c_in = F16 ? a.f16_clamp * s_in : 0.0f;
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NPRE = 2;                                     // chunks per thread for the rows of an item in flight
    const int CPR = W / 4 + 1, n_chunks = (n_in / W) * CPR;
    const bool vec_in = (W & 3) == 0 && (in_clip & 3) == 0 && n_chunks <= NPRE * NTHR;
    int ch_src[NPRE], ch_dst[NPRE], ch_kind[NPRE];              // float offset in the item's rows, byte offset in a plane, 0 / 1 / 2 = first / inner / last chunk of a row
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int c = tid + q * NTHR, y = c / CPR, j = c - y * CPR;
        ch_kind[q] = c >= n_chunks ? -1 : j == 0 ? 0 : j == CPR - 1 ? 2 : 1;
        ch_src[q] = y * W + (j == 0 ? 0 : j == CPR - 1 ? W - 4 : 4 * j - 1);
        ch_dst[q] = ((y + row_shift) * Wp0 + 4 * j) * 2;
    }
    auto store_chunk = [&](unsigned char* planes, int q, float4 v) {
        if (ch_kind[q] == 0) v = make_float4(0.0f, v.x, v.y, v.z);             // column 0 is the halo
        else if (ch_kind[q] == 2) v = make_float4(v.w, 0.0f, 0.0f, 0.0f);     // input W - 1, then the halo
        unsigned char* d = planes + ch_dst[q];
        if (F16) {
            uint32_t h0, l0, h1, l1;
            nww_split2h(__builtin_amdgcn_fmed3f(v.x * s_in, -c_in, c_in), __builtin_amdgcn_fmed3f(v.y * s_in, -c_in, c_in), h0, l0);
            nww_split2h(__builtin_amdgcn_fmed3f(v.z * s_in, -c_in, c_in), __builtin_amdgcn_fmed3f(v.w * s_in, -c_in, c_in), h1, l1);
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + plane_b) = make_uint2(l0, l1);
        } else {
            uint32_t w[3][4];
            split3(v.x, w[0][0], w[1][0], w[2][0]); split3(v.y, w[0][1], w[1][1], w[2][1]);
            split3(v.z, w[0][2], w[1][2], w[2][2]); split3(v.w, w[0][3], w[1][3], w[2][3]);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                *reinterpret_cast<uint2*>(d + t * plane_b) = make_uint2(pack_hi16(w[t][0], w[t][1]), pack_hi16(w[t][2], w[t][3]));
        }
    };
    auto load_plane_sync = [&](unsigned char* planes, const float* xin) {
        for (int idx = tid; idx < n_in; idx += NTHR) {
            const int y = idx / W, x = idx - y * W;
            unsigned char* d = planes + ((y + row_shift) * Wp0 + x + 1) * 2;
            if (F16) {
                uint32_t h, l;
                nww_split2h(__builtin_amdgcn_fmed3f(xin[idx] * s_in, -c_in, c_in), 0.0f, h, l);
                *reinterpret_cast<uint16_t*>(d) = (uint16_t)h;
                *reinterpret_cast<uint16_t*>(d + plane_b) = (uint16_t)l;
            } else {
                uint32_t wh, wm, wlo;
                split3(xin[idx], wh, wm, wlo);
                *reinterpret_cast<uint16_t*>(d) = (uint16_t)(wh >> 16);
                *reinterpret_cast<uint16_t*>(d + plane_b) = (uint16_t)(wm >> 16);
                *reinterpret_cast<uint16_t*>(d + 2 * plane_b) = (uint16_t)(wlo >> 16);
            }
        }
    };

    [[maybe_unused]] int item_no = -1;                          // trace builds only
    // conv1 groups: 32 consecutive pooled pixels of one A1 row; group g = (row g / ngx, block g % ngx)
    const int ngx = (W1 + 31) / 32, nG = n_a1 * ngx;
    // conv1 of one item: groups first, first + stride, ... from `planes` into `a1buf`
    auto conv1_groups = [&](const unsigned char* planes, unsigned char* a1buf, int first, int stride) {
        if (TB_ABL & 4) return;
        bf16x8 wf[2][NT];
#pragma unroll
        for (int q = 0; q < NF1; ++q) wf[q / NT][q % NT] = *reinterpret_cast<const bf16x8*>(W1F + q * 1024 + lane * 16);
        const unsigned char* in_lane = planes + (2 * hi) * pitch0;
        unsigned char* a1w_lane = a1buf + a1_shift * rowB + PS + 16 * hi;        // pixel x of A1 row R at + R * rowB + x * PS
        auto load_patch = [&](int R, int gx, bf16x8 (&f)[NT]) {
            const int xc = min(32 * gx + i, W1 - 1);
            const unsigned char* base = in_lane + (2 * R) * pitch0 + 4 * xc;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(base + t * plane_b);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(base + t * plane_b + pitch0);
                f[t] = frag4(p[0], p[1], q[0], q[1]);
            }
        };
        const int dR = stride / ngx, dX = stride - dR * ngx;
        int R = first / ngx, X = first - R * ngx;
        // one group: the patch fragments cf -> 6 (two-term) MFMAs -> bias / act / pool / split -> the 16-byte A1 stores
        auto group = [&](const bf16x8 (&cf)[NT], int Rc, int Xc) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            x3_mfma<PRODUCTS>(wf[0], cf, acc0);
            x3_mfma<PRODUCTS>(wf[1], cf, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (TB_ABL & 16) { asm volatile("" ::"a"(acc0), "a"(acc1)); return; }
            uint32_t ph[4], pm[4], pl[4];
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                float m2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cc = 2 * c2 + e;
                    m2[e] = pool_quad<ACT, BN, F16>(acc0[2 * cc], acc0[2 * cc + 1], acc1[2 * cc], acc1[2 * cc + 1],
                                                    b1v[cc], nb1v[cc], al1v[cc], be1v[cc], post1, POS);
                }
                if (F16) {
                    nww_split2h(m2[0], m2[1], ph[c2], pm[c2]);
                } else {
                    uint32_t th[2], tm[2], tl[2];
                    split3(m2[0], th[0], tm[0], tl[0]);
                    split3(m2[1], th[1], tm[1], tl[1]);
                    ph[c2] = pack_hi16(th[0], th[1]); pm[c2] = pack_hi16(tm[0], tm[1]); pl[c2] = pack_hi16(tl[0], tl[1]);
                }
            }
            const int x = 32 * Xc + i;
            if (x < W1) {
                unsigned char* wq = a1w_lane + Rc * rowB + x * PS;
                *reinterpret_cast<uint4*>(wq) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                *reinterpret_cast<uint4*>(wq + 32) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
                if (!F16) *reinterpret_cast<uint4*>(wq + 64) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
        };
        auto advance = [&](int& Rr, int& Xx) { Rr += dR; Xx += dX; if (Xx >= ngx) { Xx -= ngx; ++Rr; } };
        // Groups in PAIRS with two named fragment buffers: the next group's patch is fetched into the other buffer above this group's
        // MFMAs.  (One buffer rotated through a copy - cf = nf - cost 16 v_mov per group: hipcc moved the fragments both ways at the
        // loop edge, 15 % of the phase's VALU instructions.)
        bf16x8 fa[NT], fb[NT];
        if (first < nG) load_patch(R, X, fa);
        for (int g = first; g < nG; g += 2 * stride) {
            const int Ra = R, Xa = X;
            advance(R, X);
            const bool has_b = g + stride < nG;
            if (has_b) load_patch(R, X, fb);
            __builtin_amdgcn_sched_barrier(0);
            group(fa, Ra, Xa);
            if (has_b) {
                const int Rb = R, Xb = X;
                advance(R, X);
                if (g + 2 * stride < nG) load_patch(R, X, fa);
                __builtin_amdgcn_sched_barrier(0);
                group(fb, Rb, Xb);
            }
        }
    };
    // conv2 tiles t0, t0 + 4, ... (cnt of them) of the item whose A1 is a1buf and whose output block starts at outb
    auto conv2_tiles = [&](const unsigned char* a1buf, float* outb, int t0, int cnt) {
        if ((TB_ABL & 8) || cnt <= 0) return;
        int R = t0 / nX, X = t0 - R * nX;
        auto tile_dst = [&](int Rr, int Xx) -> float* {
            if (ring) {                                                        // (wave-uniform) pooled row Rr of the strip -> its ring row
                int row = ring_r0 + Rr;
                if (row >= ring) row -= ring;
                return outb + row * W2 + 8 * Xx + out_lane;
            }
            const int rel = Rr * W2 + 8 * Xx;                                  // wave-uniform
            if (!blk_kt) return outb + rel + out_lane;
            const int k = out_lane + (int)out_off + rel;                       // feature index of (channel, row, column)
            return outb + (size_t)(k >> 5) * (128 * 32) + (k & 31);
        };
        auto tile_nv = [&](int Xx) { return (W2 & 3) == 0 ? (8 * Xx + 4 * hi + 3 < W2 ? 4 : 0) : max(0, min(4, W2 - (8 * Xx + 4 * hi))); };
        auto tile_pa = [&](int Rr, int Xx) { return a1buf + a1_lane + (2 * Rr) * rowB + (16 * Xx) * PS; };
        auto step = [&](int& Rr, int& Xx) { Xx += 4; while (Xx >= nX) { Xx -= nX; ++Rr; } };
        if constexpr (F16) {
            // tiles in pairs (two accumulator chains per wave, conv2_pair_h2; -3.6 % of the launch), an odd last one by itself
            int n = 0;
            for (; n + 1 < cnt; n += 2) {
                int R1 = R, X1 = X;
                step(R1, X1);
                conv2_pair_h2<ACT, BN, POS>(tile_pa(R, X), tile_pa(R1, X1), rowB, bw, bias2, -bias2, al2, be2, post2,
                                       tile_dst(R, X), tile_nv(X), tile_dst(R1, X1), tile_nv(X1));
                if (n < 4) TB_STAMP(2 + n / 2);
                R = R1; X = X1;
                step(R, X);
            }
            if (n < cnt) {
                bf16x8 ring[3][2];
                const unsigned char* pa = tile_pa(R, X);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ring[t][0] = *reinterpret_cast<const bf16x8*>(pa + t * PS);
                    ring[t][1] = *reinterpret_cast<const bf16x8*>(pa + t * PS + 32);
                }
                conv2_tile_h2<ACT, BN, POS>(pa, nullptr, rowB, bw, ring, bias2, -bias2, al2, be2, post2, tile_dst(R, X), tile_nv(X));
            }
        } else {
            for (int n = 0; n < cnt; ++n) {
                conv2_tile<ACT, PRODUCTS, BN, POS>(tile_pa(R, X), rowB, bw, wl, bias2, -bias2, al2, be2, post2, tile_dst(R, X), tile_nv(X));
                if (n < 4) TB_STAMP(2 + n);
                step(R, X);
            }
        }
    };

    // rows of item bb -> registers (a float4 or two per thread)
    float4 pre[NPRE];
    auto fetch_rows = [&](int bb) {
        if (TB_ABL & 64) return;
        const float* xin = a.in + (size_t)bb * in_clip + in_off;
#pragma unroll
        for (int q = 0; q < NPRE; ++q)
            if (ch_kind[q] >= 0) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(xin + ch_src[q]);
                pre[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
    };
    __syncthreads();
    if (b0 < a.B) load_plane_sync(In3, a.in + (size_t)b0 * in_clip + in_off);
    if (vec_in && b0 + bstep < a.B) fetch_rows(b0 + bstep);
    __syncthreads();
    // ---- two phases per item, all eight waves in each: conv1 (planes -> A1) | the next item's rows -> planes, conv2.
    // The rows are requested a whole item ahead and stored right behind the barrier that frees the planes: stored behind conv2
    // instead, their s_waitcnt vmcnt(0) also covered conv2's last output stores (gfx9 counts loads and stores in one counter) -
    // 1.5-3 k clocks per item with every wave parked (tools/ubench/trunk_trace.hip).
    TB_STAMP_WG(1);
    for (int b = b0; b < a.B; b += bstep) {
        const int b1 = b + bstep, b2 = b1 + bstep;
        const bool has1 = b1 < a.B;
        ++item_no;
        TB_STAMP(0);
        TB_STAMP_W(6);
        conv1_groups(In3, A1, wave, NW);
        TB_STAMP(1);
        __syncthreads();
        float* outb = ring ? a.out + (size_t)b * a.out_clip_stride
                    : blk_kt ? a.out + ((size_t)(b >> 7) * blk_kt * 128 + (b & 127)) * 32
                             : a.out + (size_t)b * C2 * H2 * W2 + out_off;
        if (has1 && !(TB_ABL & (32 | 64))) {
            if (vec_in) {
#pragma unroll
                for (int q = 0; q < NPRE; ++q)
                    if (ch_kind[q] >= 0) store_chunk(In3, q, pre[q]);
            } else {
                load_plane_sync(In3, a.in + (size_t)b1 * in_clip + in_off);
            }
        }
        if (vec_in && b2 < a.B) fetch_rows(b2);
        if (wave < 4) conv2_tiles(A1, outb, simd + 4 * tB, tA);
        else conv2_tiles(A1, outb, simd, tB);
        __syncthreads();
        TB_STAMP(7);
    }
    TB_STAMP_WG(2);
}
template <int ACT, int PRODUCTS, bool BN>
__global__ void __launch_bounds__(512, 2) cnn_trunk_b_kernel(TrunkArgs a) { cnn_trunk_b_body<ACT, PRODUCTS, BN>(a); }
template <int ACT, bool BN, bool POS = false>
__global__ void __launch_bounds__(512, 2) cnn_trunk_h2_kernel(TrunkArgs a) { cnn_trunk_b_body<ACT, 3, BN, POS>(a); }



// ---------------------------------------------------------------------------------------------- BcResNet front
// Conv2d(1,32,3,p1) + BN + act + MaxPool2 fused with block 1's depthwise 3x3 (architectures.py:632-647, 653-660), the
// convolution as the transposed split-operand product above, twice (channels 0-15 / 16-31 = two weight sets).  The
// float32-MFMA version (trunk.hip: conv1_pool_dw_nhwc_kernel) spends 24 x 32 matrix-pipe clocks per 16 pooled pixels AND
// blocks the SIMD's VALU while it does; here 32 pooled pixels cost 24 x 32, and the pooling of other waves runs beside them.
// A workgroup walks clips; a clip is cut into strips of rows_dw depthwise rows: the strip's input rows (three bf16 planes,
// a float4 per thread, in flight during the previous strip's convolution) -> conv tasks (conv row, 32 pixels, weight set:
// a wave keeps ONE set for the whole launch) -> P [rows][W1][32] float32 in LDS -> depthwise out of P, four channels per
// thread.  80 KB of LDS and <= 128 registers: two workgroups per CU, one's MFMAs under the other's depthwise phase.
struct BfGeom { int Wp0, pitch0, max_conv, max_in, plane_b; };
__host__ __device__ inline BfGeom bf_geom(int W, int sh, int rows_dw) {
    BfGeom g;
    g.Wp0 = (W + 3) & ~1;
    g.pitch0 = 2 * g.Wp0;
    g.max_conv = sh * (rows_dw - 1) + 3;
    g.max_in = 2 * g.max_conv + 2;
    g.plane_b = (g.max_in * g.pitch0 + 15) & ~15;
    return g;
}
// P [row][pixel slot][8 quads of 4 channels]: pixel x sits in slot pslot(x) = x with bit 0 flipped where bit 2 is set, its quad q at
// position q ^ (x & 7).  The layout follows the LDS's lane groups (MI355X_MICROARCH.md, LDS): a ds_write_b128 is served in groups of 8
// consecutive lanes over 32 banks - the conv tasks' eight consecutive pixels store one quad each at eight different 16-byte positions of
// the 128-byte window; a ds_read_b128 in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) over 64 banks - the depthwise's lanes
// (quad cq = lane & 7, pixel 4 xg + m, xg = lane >> 3) of such a group hold xg = 0 .. 3 with cq 0-3 | 4-7 | 4-7 | 0-3: the XOR puts xg 0, 1
// and xg 2, 3 on different quad halves, the slot's parity (bit 0 of pslot = (m ^ xg) & 1) xg 0 and 1, 2 and 3 on different 128-byte
// halves of the 256-byte bank row: sixteen different positions.  (Rounds 3-5 swapped bits 0 and 1 of x and XOR-ed two bits - made for
// groups of consecutive lanes over 64 banks: every store and every depthwise read was a 2-way conflict, SQ_LDS_BANK_CONFLICT 2.6e7 of 9.1e7
// LDS cycles, half from each: tools/ubench/front_ab with one access class ablated at a time.)
__device__ __forceinline__ int bf_pslot(int x) { return x ^ ((x >> 2) & 1); }
__device__ __forceinline__ int bf_pswz(int x) { return x & 7; }
constexpr int BF_HEAD = 1552;                 // bytes: depthwise weights [9][32] + bias / alpha / beta [3][32] floats + 16 zero bytes
constexpr int BF_W1F = 12 * 1024;             // conv weight fragments [set][dy][term][64 lanes] x 16 bytes

__global__ void __launch_bounds__(64) bc_front_pack_kernel(const float* __restrict__ w1, unsigned char* __restrict__ out) {
    const int lane = threadIdx.x, m = lane & 31, hi = lane >> 5;
    bf16x8* o = reinterpret_cast<bf16x8*>(out);
    const int cw = 8 * ((m >> 2) & 1) + 2 * (m >> 3) + ((m >> 1) & 1), dxw = m & 1;      // trunk_b_pack_kernel's row map
    for (int set = 0; set < 2; ++set)
        for (int dy = 0; dy < 2; ++dy) {
            uint32_t th[8], tm[8], tl[8];
            for (int kk = 0; kk < 8; ++kk) {
                const int ty = 2 * hi + (kk >> 2) - dy, tx = (kk & 3) - dxw;
                const bool in = ty >= 0 && ty < 3 && tx >= 0 && tx < 3;
                split3(in ? w1[(16 * set + cw) * 9 + ty * 3 + tx] : 0.0f, th[kk], tm[kk], tl[kk]);
            }
            const int f = (set * 2 + dy) * 3;
            o[(f + 0) * 64 + lane] = frag4(pack_hi16(th[0], th[1]), pack_hi16(th[2], th[3]), pack_hi16(th[4], th[5]), pack_hi16(th[6], th[7]));
            o[(f + 1) * 64 + lane] = frag4(pack_hi16(tm[0], tm[1]), pack_hi16(tm[2], tm[3]), pack_hi16(tm[4], tm[5]), pack_hi16(tm[6], tm[7]));
            o[(f + 2) * 64 + lane] = frag4(pack_hi16(tl[0], tl[1]), pack_hi16(tl[2], tl[3]), pack_hi16(tl[4], tl[5]), pack_hi16(tl[6], tl[7]));
        }
}

// two binary16 terms of weight x sw: fragments [set][dy][term]
__global__ void __launch_bounds__(64) bc_front_pack_f16_kernel(const float* __restrict__ w1, unsigned char* __restrict__ out, float sw) {
    const int lane = threadIdx.x, m = lane & 31, hi = lane >> 5;
    bf16x8* o = reinterpret_cast<bf16x8*>(out);
    const int cw = 8 * ((m >> 2) & 1) + 2 * (m >> 3) + ((m >> 1) & 1), dxw = m & 1;
    for (int set = 0; set < 2; ++set)
        for (int dy = 0; dy < 2; ++dy) {
            float v[8];
            for (int kk = 0; kk < 8; ++kk) {
                const int ty = 2 * hi + (kk >> 2) - dy, tx = (kk & 3) - dxw;
                const bool in = ty >= 0 && ty < 3 && tx >= 0 && tx < 3;
                v[kk] = in ? w1[(16 * set + cw) * 9 + ty * 3 + tx] * sw : 0.0f;
            }
            uint32_t th[4], tl[4];
            for (int j = 0; j < 4; ++j) nww_split2h(v[2 * j], v[2 * j + 1], th[j], tl[j]);
            const int f = (set * 2 + dy) * 2;
            o[(f + 0) * 64 + lane] = frag4(th[0], th[1], th[2], th[3]);
            o[(f + 1) * 64 + lane] = frag4(tl[0], tl[1], tl[2], tl[3]);
        }
}

// PRODUCTS = 3 (BN only): two binary16 terms per operand - the input times a.f16_in (clamped to +-a.f16_clamp first) in TWO planes, the
// weights packed by bc_front_pack_f16_kernel; the accumulators are k = f16_in x weight scale times the true sums and a.f16_unscale =
// 1 / k goes into the folded-BN factor (pool_quad's comment)
template <int ACT, int PRODUCTS, bool BN>
__global__ void __launch_bounds__(512, 4) bc_front_b_kernel(Conv1DwArgs a) {
    constexpr bool F16 = PRODUCTS == 3;
    constexpr int NT = F16 ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int H = a.H, W = a.W, H1 = H / 2, W1 = W / 2;
    const int Ho = a.Ho, Wo = a.Wo, sh = a.sh, sw = a.sw;
    const BfGeom gg = bf_geom(W, sh, a.rows_dw);
    const int Wp0 = gg.Wp0, pitch0 = gg.pitch0, plane_b = gg.plane_b;
    float* const Wd = reinterpret_cast<float*>(lds_raw);
    float* const BNp = Wd + 288;
    float* const Zp = BNp + 96;                               // four zeros: where the depthwise's out-of-plane taps read
    unsigned char* const In3 = lds_raw + BF_HEAD;
    float* const P = reinterpret_cast<float*>(In3 + NT * plane_b);
    const int W1p = (W1 + 3) & ~3;                            // pixel slots per P row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hi = lane >> 5;

    for (int k = tid; k < NT * plane_b / 4; k += NTHR) reinterpret_cast<uint32_t*>(In3)[k] = 0u;      // halo columns stay zero
    // P's pixel slots W1 .. W1p - 1 (never written) and the row behind the strip's max_conv rows stay zero: the depthwise reads them as
    // its out-of-plane columns / rows
    for (int k = tid; k < (gg.max_conv + 1) * W1p * 8; k += NTHR) reinterpret_cast<float4*>(P)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = tid; k < 288; k += NTHR) Wd[k] = a.dw_wt[k];
    if (tid < 4) Zp[tid] = 0.0f;
    if (tid < 32) {
        BNp[tid] = a.bias ? a.bias[tid] * (F16 ? 1.0f / a.f16_unscale : 1.0f) : 0.0f;
        BNp[32 + tid] = (BN && a.alpha) ? a.alpha[tid] * (F16 ? a.f16_unscale : 1.0f) : 1.0f;
        BNp[64 + tid] = (BN && a.alpha) ? a.beta[tid] : 0.0f;
    }
    __syncthreads();
    // this wave's weight set (tasks wave, wave + 8, ... all have the wave's parity) stays in registers
    const int set = wave & 1;
    bf16x8 wf[2][3];
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) wf[q / NT][q % NT] = *reinterpret_cast<const bf16x8*>(a.wpack + (set * 2 * NT + q) * 1024 + lane * 16);      // (straight from the packed image: 12 KB of LDS more for P)

    const int nstrips = (Ho + a.rows_dw - 1) / a.rows_dw;
    const int ngx = (W1 + 31) / 32;
    // strip geometry: depthwise rows [oy0, oy1), conv (pooled) rows [r_lo, r_hi], input rows from y0 = 2 r_lo - 1
    auto strip_rows = [&](int sidx, int& oy0, int& oy1, int& r_lo, int& r_hi) {
        oy0 = sidx * a.rows_dw; oy1 = min(Ho, oy0 + a.rows_dw);
        r_lo = max(0, sh * oy0 - 1); r_hi = min(H1 - 1, sh * (oy1 - 1) + 1);
    };
    constexpr int NPRE = 2;
    const bool vec_in = (W & 3) == 0 && gg.max_in * W <= 4 * NPRE * NTHR;
    // input rows of a strip -> registers (out-of-range rows are zeros) / -> the three planes
    auto fetch = [&](int b, int sidx, float4 (&pre)[NPRE]) {
        int oy0, oy1, r_lo, r_hi;
        strip_rows(sidx, oy0, oy1, r_lo, r_hi);
        const int y0 = 2 * r_lo - 1, n4 = (2 * (r_hi - r_lo + 1) + 2) * W / 4;
        const float* xin = a.in + (size_t)b * H * W;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int idx4 = tid + q * NTHR;
            pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx4 < n4) {
                const int lr = (4 * idx4) / W, x = 4 * idx4 - lr * W, y = y0 + lr;
                if (y >= 0 && y < H) pre[q] = *reinterpret_cast<const float4*>(xin + (y * W + x));      // (a clip's plane: 32-bit offsets)
            }
        }
    };
    auto put = [&](int sidx, const float4 (&pre)[NPRE]) {
        int oy0, oy1, r_lo, r_hi;
        strip_rows(sidx, oy0, oy1, r_lo, r_hi);
        const int n4 = (2 * (r_hi - r_lo + 1) + 2) * W / 4;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int idx4 = tid + q * NTHR;
            if (idx4 < n4) {
                const int lr = (4 * idx4) / W, x = 4 * idx4 - lr * W;
                unsigned char* d = In3 + (lr * Wp0 + x + 1) * 2;
                if constexpr (F16) {
                    const float cl = a.f16_clamp, sc = a.f16_in;
                    uint32_t h01, l01, h23, l23;
                    nww_split2h(__builtin_amdgcn_fmed3f(pre[q].x, -cl, cl) * sc, __builtin_amdgcn_fmed3f(pre[q].y, -cl, cl) * sc, h01, l01);
                    nww_split2h(__builtin_amdgcn_fmed3f(pre[q].z, -cl, cl) * sc, __builtin_amdgcn_fmed3f(pre[q].w, -cl, cl) * sc, h23, l23);
                    const uint32_t hh[2][2] = {{h01, h23}, {l01, l23}};
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        unsigned char* p = d + t * plane_b;
                        *reinterpret_cast<uint16_t*>(p) = (uint16_t)(hh[t][0] & 0xffffu);
                        *reinterpret_cast<uint32_t*>(p + 2) = (hh[t][0] >> 16) | (hh[t][1] << 16);
                        *reinterpret_cast<uint16_t*>(p + 6) = (uint16_t)(hh[t][1] >> 16);
                    }
                } else {
                    uint32_t w[3][4];
                    split3(pre[q].x, w[0][0], w[1][0], w[2][0]); split3(pre[q].y, w[0][1], w[1][1], w[2][1]);
                    split3(pre[q].z, w[0][2], w[1][2], w[2][2]); split3(pre[q].w, w[0][3], w[1][3], w[2][3]);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        unsigned char* p = d + t * plane_b;
                        *reinterpret_cast<uint16_t*>(p) = (uint16_t)(w[t][0] >> 16);
                        *reinterpret_cast<uint32_t*>(p + 2) = pack_hi16(w[t][1], w[t][2]);
                        *reinterpret_cast<uint16_t*>(p + 6) = (uint16_t)(w[t][3] >> 16);
                    }
                }
            }
        }
    };
    auto stage_sync = [&](int b, int sidx) {                   // generic shapes: no prefetch
        int oy0, oy1, r_lo, r_hi;
        strip_rows(sidx, oy0, oy1, r_lo, r_hi);
        const int y0 = 2 * r_lo - 1, n = (2 * (r_hi - r_lo + 1) + 2) * W;
        const float* xin = a.in + (size_t)b * H * W;
        int t0 = tid;
        asm volatile("" : "+v"(t0));                          // (nothing of this loop hoisted out of the clip loop: the fast path has no registers for it)
        for (int idx = t0; idx < n; idx += NTHR) {
            const int lr = idx / W, x = idx - lr * W, y = y0 + lr;
            const float v = (y >= 0 && y < H) ? xin[y * W + x] : 0.0f;
            unsigned char* d = In3 + (lr * Wp0 + x + 1) * 2;
            if constexpr (F16) {
                uint32_t hh, ll;
                nww_split2h(__builtin_amdgcn_fmed3f(v, -a.f16_clamp, a.f16_clamp) * a.f16_in, 0.0f, hh, ll);
                *reinterpret_cast<uint16_t*>(d) = (uint16_t)(hh & 0xffffu);
                *reinterpret_cast<uint16_t*>(d + plane_b) = (uint16_t)(ll & 0xffffu);
            } else {
                uint32_t wh, wm, wlo;
                split3(v, wh, wm, wlo);
                *reinterpret_cast<uint16_t*>(d) = (uint16_t)(wh >> 16);
                *reinterpret_cast<uint16_t*>(d + plane_b) = (uint16_t)(wm >> 16);
                *reinterpret_cast<uint16_t*>(d + 2 * plane_b) = (uint16_t)(wlo >> 16);
            }
        }
    };
    const int cq = tid & 7, dslot = tid >> 3;                  // depthwise: four channels 4 cq .., NTHR / 8 output slots
    const bool bn_pos = __builtin_amdgcn_readfirstlane(a.bn_pos) != 0, has_bias = a.bias != nullptr;
    // depthwise, stride 2 along x (the BcResNet geometry): a thread owns TWO outputs along x of one row for the whole launch - its five
    // input columns' offsets in P (slot, swizzle) are computed once here, out-of-plane columns marked -1
    const int Wg2 = (Wo + 1) >> 1, dw_rpp = (NTHR >> 3) / max(Wg2, 1);
    const bool dw_fast = F16 && sw == 2 && dw_rpp >= 1;      // (the three-term instances have no registers to spare for it)
    const int dw_xg = dslot % max(Wg2, 1), dw_ry = dslot / max(Wg2, 1);
    // byte offsets inside a P row of the lane's quad of its five columns 4 xg - 1 .. 4 xg + 3 (bf_pslot, bf_pswz): column 4 xg at colE,
    // 4 xg + 1 at colO, and with o the offset of column x, column x + 2 sits at (o ^ 32) + 256 (two slots on, bit 1 of the quad position
    // flipped); column 4 xg - 1 at colL - for xg = 0 it lies outside the plane and the lane reads the zero row instead (colL = 0 there).
    // Columns >= W1 are zero slots of P.
    auto dw_col = [&](int xx) { return bf_pslot(xx) * 128 + 16 * (cq ^ bf_pswz(xx)); };
    const int colE0 = dw_col(4 * dw_xg), colO0 = dw_col(4 * dw_xg + 1);
    const int colL0 = dw_xg > 0 ? dw_col(4 * dw_xg - 1) : 0;
    const int prow_b = W1p * 128;                               // bytes per P row
    const int dw_row0 = dw_ry * sh * prow_b;                    // the lane's row of a depthwise pass, in bytes of P
    const int dw_out0 = (dw_ry * Wo + 2 * dw_xg) * 32 + 4 * cq; // its first output inside the pass's rows
    const int oy_last2 = (H1 - 2) / max(sh, 1);                 // the last depthwise row whose third input row (oy sh + 1) lies in the plane
    const unsigned char* const Pb = reinterpret_cast<const unsigned char*>(P);
    // convolution tasks with one 32-pixel group per row (W1 <= 32): the lane's operand / result addresses are a lane constant plus a wave-uniform
    // row term
    const bool conv1g = ngx == 1;
    const int cv_xc = min(i, W1 - 1);
    const int cv_in0 = 2 * hi * pitch0 + 4 * cv_xc;
    const int cv_out0 = bf_pslot(cv_xc) * 128 + 16 * ((4 * set + 2 * hi) ^ bf_pswz(cv_xc));

    int b = blockIdx.x, sidx = 0;
#ifdef NWW_TRACE
    int trace_it = 0;
#endif
    if (b < a.B) {
        if (vec_in) { float4 p0[NPRE]; fetch(b, 0, p0); put(0, p0); }
        else stage_sync(b, 0);
    }
    __syncthreads();
    while (b < a.B) {
        int nb = b, ns = sidx + 1;
        if (ns >= nstrips) { ns = 0; nb = b + gridDim.x; }
        const bool has_next = nb < a.B;
        float4 pre[NPRE];
        BF_STAMP(trace_it, 0)
        if (has_next && vec_in) fetch(nb, ns, pre);
        BF_STAMP(trace_it, 1)
        int oy0, oy1, r_lo, r_hi;
        strip_rows(sidx, oy0, oy1, r_lo, r_hi);
        // the lane constants pass through an opaque copy per strip: left loop-invariant, every address derived from them (one per LDS access of
        // the two phases) is hoisted out of the clip loop and spilled - 59 registers' worth
        int cv_in = cv_in0, cv_out = cv_out0, colE = colE0, colO = colO0, colL = colL0, cqv = cq, xgv = dw_xg, rowv = dw_row0, outv = dw_out0;
        asm volatile("" : "+v"(cv_in), "+v"(cv_out), "+v"(colE), "+v"(colO), "+v"(colL), "+v"(cqv), "+v"(xgv), "+v"(rowv), "+v"(outv));
        const int cv_out1 = cv_out ^ 16;                          // ((q0 + 1) ^ sz = (q0 ^ sz) ^ 1: q0 is even)
        // ---- conv + BN + act + pool of rows r_lo .. r_hi -> P
        const int ntask = (r_hi - r_lo + 1) * ngx * 2;
        for (int t = wave; t < ntask; t += NW) {
            const int g = t >> 1, Rl = conv1g ? g : g / ngx, gx = conv1g ? 0 : g - Rl * ngx;
            const int xc = min(32 * gx + i, W1 - 1);
            const unsigned char* base = conv1g ? In3 + cv_in + 2 * Rl * pitch0 : In3 + (2 * Rl + 2 * hi) * pitch0 + 4 * xc;
            bf16x8 cf[3];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(base + tt * plane_b);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(base + tt * plane_b + pitch0);
                cf[tt] = frag4(p[0], p[1], q[0], q[1]);
            }
            // the wave's folded-BN rows (and bias rows, if the layer has a bias - BcResNet's init conv has none) travel under the MFMAs:
            // requested behind them they put an LDS round trip into every task
            const float4* bp = reinterpret_cast<const float4*>(BNp + 16 * set + 8 * hi);
            float4 b0v = make_float4(0.f, 0.f, 0.f, 0.f), b1v = b0v;
            if (has_bias) { b0v = bp[0]; b1v = bp[1]; }
            const float4 a0v = bp[8], a1v = bp[9], e0v = bp[16], e1v = bp[17];
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            x3_mfma<PRODUCTS>(wf[0], cf, acc0);
            x3_mfma<PRODUCTS>(wf[1], cf, acc1);
            // pool_quad reads the accumulators (VGPRs in this kernel) from inline asm, which the compiler's hazard recogniser does
            // not see as a VALU read of an MFMA result: without these wait states the first windows read acc1 before the last
            // MFMA has written it (tools/ubench/front_cmp.hip caught it as sporadic wrong values in the first channels)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float bs[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
            const float al[8] = {a0v.x, a0v.y, a0v.z, a0v.w, a1v.x, a1v.y, a1v.z, a1v.w};
            const float be[8] = {e0v.x, e0v.y, e0v.z, e0v.w, e1v.x, e1v.y, e1v.z, e1v.w};
            float m[8];
            if (BN && bn_pos && !has_bias) {                      // (one wave-uniform branch per task instead of one per channel)
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    m[cc] = pool_quad<ACT, BN, false, true>(acc0[2 * cc], acc0[2 * cc + 1], acc1[2 * cc], acc1[2 * cc + 1], 0.0f, 0.0f, al[cc], be[cc], 1.0f, true);
            } else if (bn_pos) {
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    m[cc] = pool_quad<ACT, BN>(acc0[2 * cc], acc0[2 * cc + 1], acc1[2 * cc], acc1[2 * cc + 1], bs[cc], -bs[cc], al[cc], be[cc], 1.0f, true);
            } else {
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    m[cc] = pool_quad<ACT, BN>(acc0[2 * cc], acc0[2 * cc + 1], acc1[2 * cc], acc1[2 * cc + 1], bs[cc], -bs[cc], al[cc], be[cc], 1.0f, false);
            }
            const int x = 32 * gx + i;
            int o0 = cv_out, o1 = cv_out1;                        // byte offsets of the lane's two quads in its P row
            if (!conv1g) {
                const int q0 = 4 * set + 2 * hi, sz = bf_pswz(x);
                o0 = bf_pslot(x) * 128 + 16 * (q0 ^ sz); o1 = o0 ^ 16;
            }
            if (x < W1) {
                unsigned char* const prow = reinterpret_cast<unsigned char*>(P) + Rl * prow_b;
                *reinterpret_cast<float4*>(__builtin_assume_aligned(prow + o0, 16)) = make_float4(m[0], m[1], m[2], m[3]);
                *reinterpret_cast<float4*>(__builtin_assume_aligned(prow + o1, 16)) = make_float4(m[4], m[5], m[6], m[7]);
            }
        }
        BF_STAMP(trace_it, 2)
        __syncthreads();
        BF_STAMP(trace_it, 3)
        // the planes are free: the next strip's rows land while this strip's depthwise runs
        if (has_next) {
            if (vec_in) put(ns, pre);
            else stage_sync(nb, ns);
        }
        BF_STAMP(trace_it, 4)
        // ---- depthwise 3x3 of the strip's rows out of P (taps in dwconv3x3_nhwc_kernel's order and fmaf chain)
        if (F16 && dw_fast) {
            if (dw_ry < dw_rpp) {
                // every lane term of the addresses is a launch constant (rowv, outv); what changes with the strip / the pass is wave-uniform
                int oyb = oy0;                                    // the pass's first row
                for (int oyl = dw_ry; oyl < oy1 - oy0; oyl += dw_rpp, oyb += dw_rpp) {
                    const int oy = oy0 + oyl;
                    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, centre[2] = {acc[0], acc[0]};
                    // rows oy sh - 1 .. + 1 of the plane (the middle one always exists), an out-of-plane row = the zero row
                    // (row offsets in bytes of P: multiples of 128, so the XOR of a column offset's bit 5 commutes with adding them)
                    const int r1 = (oyb * sh - r_lo) * prow_b + rowv, zr = gg.max_conv * prow_b;
                    const int rb[3] = {oy > 0 ? r1 - prow_b : zr, r1, oy <= oy_last2 ? r1 + prow_b : zr};
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int pe = rb[dy] + colE, po = rb[dy] + colO;
                        const int pl = (xgv > 0 ? rb[dy] : zr) + colL;
                        float4 v[5];
                        v[0] = *reinterpret_cast<const float4*>(Pb + pl);
                        v[1] = *reinterpret_cast<const float4*>(Pb + pe);
                        v[2] = *reinterpret_cast<const float4*>(Pb + po);
                        v[3] = *reinterpret_cast<const float4*>(Pb + (pe ^ 32) + 256);
                        v[4] = *reinterpret_cast<const float4*>(Pb + (po ^ 32) + 256);
                        if (dy == 1) { centre[0] = v[1]; centre[1] = v[3]; }
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float4 w = *reinterpret_cast<const float4*>(Wd + (dy * 3 + dx) * 32 + 4 * cqv);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float4 pv = v[2 * j + dx];
                                acc[j].x = fmaf(pv.x, w.x, acc[j].x); acc[j].y = fmaf(pv.y, w.y, acc[j].y);
                                acc[j].z = fmaf(pv.z, w.z, acc[j].z); acc[j].w = fmaf(pv.w, w.w, acc[j].w);
                            }
                        }
                        // a row at a time: left alone, hipcc issues all 24 loads first and sinks the arithmetic into the stores' branches - 96
                        // registers of operands
#pragma unroll
                        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[j].x), "+v"(acc[j].y), "+v"(acc[j].z), "+v"(acc[j].w));
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ox = 2 * xgv + j;
                        if (ox < Wo) {
                            // the clip's plane (wave-uniform, 64 bits) + the lane's 32-bit offset inside it
                            const size_t ob = (size_t)b * Ho * Wo * 32 + (size_t)(oyb * Wo * 32 + 32 * j);
                            const unsigned ol = (unsigned)outv;
                            if (a.bf16_out) {                     // wave-uniform: 16-bit activations (split_h2.h)
                                const int k16 = a.bf16_out;
                                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.d_out) + ob + ol) =
                                    make_uint2(nww_pk_act16(k16, acc[j].x, acc[j].y, a.d_scale), nww_pk_act16(k16, acc[j].z, acc[j].w, a.d_scale));
                                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.xs_out) + ob + ol) =
                                    make_uint2(nww_pk_act16(k16, centre[j].x, centre[j].y, a.xs_scale), nww_pk_act16(k16, centre[j].z, centre[j].w, a.xs_scale));
                            } else {
                                *reinterpret_cast<float4*>(a.d_out + ob + ol) = acc[j];
                                *reinterpret_cast<float4*>(a.xs_out + ob + ol) = centre[j];
                            }
                        }
                    }
                }
            }
        } else
        for (int o = dslot; o < (oy1 - oy0) * Wo; o += NTHR / 8) {
            const int oyl = o / Wo, ox = o - oyl * Wo, oy = oy0 + oyl;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), centre = acc;
            int xoff[3];                                      // the lane's quad of the three columns (slot and swizzle depend on x only)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = min(max(ox * sw - 1 + dx, 0), W1 - 1);
                xoff[dx] = bf_pslot(xx) * 32 + 4 * (cq ^ bf_pswz(xx));
            }
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = oy * sh - 1 + dy;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = ox * sw - 1 + dx;
                    const bool ok = yy >= 0 && yy < H1 && xx >= 0 && xx < W1;
                    const float4 v = ok ? *reinterpret_cast<const float4*>(P + (size_t)(yy - r_lo) * W1p * 32 + xoff[dx])
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
        BF_STAMP(trace_it, 5)
        __syncthreads();
        BF_STAMP(trace_it, 6)
#ifdef NWW_TRACE
        ++trace_it;
#endif
        b = nb; sidx = ns;
    }
}
}  // namespace

size_t trunk_b_packed_bytes() { return (size_t)NFRAG * 1024; }
hipError_t launch_trunk_b_pack(const float* w1, const float* w2, unsigned char* packed, hipStream_t s) {
    hipLaunchKernelGGL(trunk_b_pack_kernel, dim3(1), dim3(64), 0, s, w1, w2, packed);
    return hipGetLastError();
}

hipError_t launch_trunk_b_pack_f16(const float* w1, const float* w2, unsigned char* packed, float sw1, float sw2, hipStream_t s) {
    hipLaunchKernelGGL(trunk_b_pack_f16_kernel, dim3(1), dim3(64), 0, s, w1, w2, packed, sw1, sw2);
    return hipGetLastError();
}

size_t trunk_b_lds_bytes(int H, int W, int S, int products) {
    size_t worst = 0;
    for (int s = 0; s < S; ++s) {
        const TrunkStrip g = trunk_strip(H, S, s);
        const size_t b = tb_lds_total(tb_geom(W, g, products == 3), products == 3);
        if (b > worst) worst = b;
    }
    return worst;
}
// The strip count follows the three-term form's LDS need in EVERY arithmetic (the two-term form would fit fewer, larger strips
// for some shapes): how a clip is cut then never depends on the arithmetic switch.
int trunk_b_pick_strips(int H, int W) {
    const int H2 = H / 4;
    for (int S = 1; S <= H2; ++S)
        if (trunk_b_lds_bytes(H, W, S, 6) <= 160 * 1024) return S;
    return 0;
}

bool trunk_b_rows_fit(int H, int W, int r2a, int r2b) {
    if (r2a < 0 || r2b <= r2a || r2b > H / 4) return false;
    return tb_lds_total(tb_geom(W, trunk_strip_rows(H, r2a, r2b), false), false) <= 160 * 1024;
}

#define TB_DISPATCH(aa, grid, lds)                                                                                      \
    {                                                                                                                  \
        const bool bn_ = (aa).al1 != nullptr || (aa).al2 != nullptr;                                                   \
        hipError_t e_ = hipErrorInvalidValue;                                                                          \
        int key_ = ((aa).act == ACT_RELU ? 0 : (aa).act == ACT_GELU ? 1 : (aa).act == ACT_SILU ? 2 : 3) * 4 + (products == 6 ? 0 : 2) + (bn_ ? 1 : 0); \
        if (products == 3) key_ = (aa).act == ACT_RELU ? (bn_ ? ((aa).bn_pos ? 16 : 13) : 12) : (aa).act == ACT_GELU ? 14 : (aa).act == ACT_SILU ? 15 : 99; \
        switch (key_) {                                                                                                \
            TB_CASE_H2(12, ACT_RELU, false) TB_CASE_H2(13, ACT_RELU, true) TB_CASE_H2(14, ACT_GELU, true) TB_CASE_H2(15, ACT_SILU, true) \
            TB_CASE_H2P(16, ACT_RELU)                                                                                  \
            TB_CASE(0, ACT_RELU, 6, false) TB_CASE(1, ACT_RELU, 6, true) TB_CASE(2, ACT_RELU, 9, false) TB_CASE(3, ACT_RELU, 9, true)   \
            TB_CASE(4, ACT_GELU, 6, false) TB_CASE(5, ACT_GELU, 6, true) TB_CASE(6, ACT_GELU, 9, false) TB_CASE(7, ACT_GELU, 9, true)   \
            TB_CASE(8, ACT_SILU, 6, false) TB_CASE(9, ACT_SILU, 6, true) TB_CASE(10, ACT_SILU, 9, false) TB_CASE(11, ACT_SILU, 9, true) \
            default: return hipErrorInvalidValue;                                                                      \
        }                                                                                                              \
        if (e_ != hipSuccess) return e_;                                                                               \
    }
#define TB_CASE_H2(K, ACTV, BNV)                                                                                       \
    case K:                                                                                                            \
        e_ = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_h2_kernel<ACTV, BNV>), lds);                        \
        if (e_ == hipSuccess) hipLaunchKernelGGL((cnn_trunk_h2_kernel<ACTV, BNV>), dim3(grid), dim3(NTHR), lds, s, aa); \
        break;
#define TB_CASE_H2P(K, ACTV)                                                                                           \
    case K:                                                                                                            \
        e_ = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_h2_kernel<ACTV, true, true>), lds);                 \
        if (e_ == hipSuccess) hipLaunchKernelGGL((cnn_trunk_h2_kernel<ACTV, true, true>), dim3(grid), dim3(NTHR), lds, s, aa); \
        break;
#define TB_CASE(K, ACTV, PRODV, BNV)                                                                                   \
    case K:                                                                                                            \
        e_ = nww_allow_lds(reinterpret_cast<const void*>(cnn_trunk_b_kernel<ACTV, PRODV, BNV>), lds);                  \
        if (e_ == hipSuccess) hipLaunchKernelGGL((cnn_trunk_b_kernel<ACTV, PRODV, BNV>), dim3(grid), dim3(NTHR), lds, s, aa); \
        break;

// Streaming hop: explicit strips of pooled rows and / or ring-addressed output (TrunkArgs::n_sub, out_ring_rows).  Explicit strips get
// equal shares of the grid; without them the clip is cut as usual.
static hipError_t launch_cnn_trunk_b_stream(TrunkArgs aa, int products, int max_grid, hipStream_t s) {
    const int H2 = aa.H / 4, W2 = aa.W / 4;
    if (aa.out_blocked || aa.n_sub < 0 || aa.n_sub > 4) return hipErrorInvalidValue;
    if (aa.out_ring_rows > 0 && (aa.out_ring_rows < H2 || aa.out_row0 < 0 || aa.out_row0 >= aa.out_ring_rows || aa.out_ch_stride < (size_t)aa.out_ring_rows * W2 ||
                                 aa.out_clip_stride < C2 * aa.out_ch_stride || (aa.out_ch_stride & 3) || (W2 & 3)))
        return hipErrorInvalidValue;
    for (int q = 0; q < 8; ++q) aa.wg_end[q] = 0;
    size_t lds = 0;
    int S;
    if (aa.n_sub > 0) {
        S = aa.n_sub;
        for (int q = 0; q < S; ++q) {
            if (aa.sub_a[q] < 0 || aa.sub_b[q] <= aa.sub_a[q] || aa.sub_b[q] > H2) return hipErrorInvalidValue;
            const size_t b = tb_lds_total(tb_geom(aa.W, trunk_strip_rows(aa.H, aa.sub_a[q], aa.sub_b[q]), products == 3), products == 3);
            if (b > lds) lds = b;
        }
        if (lds > 160 * 1024) return hipErrorInvalidValue;
    } else {
        S = trunk_b_pick_strips(aa.H, aa.W);
        if (S < 1) return hipErrorInvalidValue;
        for (int small_strips : {8, 6, 4})
            if (small_strips > S && (long)aa.B * small_strips * 4 <= max_grid && small_strips <= aa.H / 4) { S = small_strips; break; }
        lds = trunk_b_lds_bytes(aa.H, aa.W, S, products);
    }
    aa.strips = S;
    const int per_cu = 1;                                        // eight waves x ~220 registers: one workgroup per CU whatever its LDS
    long want = (long)aa.B * S;
    long cap = (long)max_grid * per_cu;
    int grid = (int)(want < cap ? want : cap);
    grid -= grid % S;
    if (grid < S) grid = S;
    TB_DISPATCH(aa, grid, lds)
    return hipGetLastError();
}

hipError_t launch_cnn_trunk_b(const TrunkArgs& a, int products, int max_grid, hipStream_t s) {
    if (!a.wpack || (products != 3 && products != 6 && products != 9)) return hipErrorInvalidValue;
    TrunkArgs aa = a;
    static const int force_strips = 0;
    if (a.n_sub > 0 || a.out_ring_rows > 0) return launch_cnn_trunk_b_stream(aa, products, max_grid, s);
    int S = trunk_b_pick_strips(a.H, a.W);
    if (S < 1) return hipErrorInvalidValue;
    if (force_strips > S && force_strips <= a.H / 4) S = force_strips;
    // a handful of clips (the interpreter's B = 1 .. 16 calls) would occupy a handful of CUs for a whole clip each: cut
    // every clip into more row strips on more CUs instead.  Seam rows are recomputed by both neighbours with the same
    // arithmetic, so the result does not depend on the strip count (bit for bit).
    if (!force_strips)
        for (int small_strips : {8, 6, 4})
            if (small_strips > S && (long)a.B * small_strips * 4 <= max_grid && small_strips <= a.H / 4) { S = small_strips; break; }
    aa.strips = S;
    const size_t lds = trunk_b_lds_bytes(a.H, a.W, S, products);
    long want = (long)a.B * S;
    int grid = (int)(want < max_grid ? want : (long)max_grid);
    for (int q = 0; q < 8; ++q) aa.wg_end[q] = 0;
    if (S <= 8 && S > 1 && (long)a.B >= 2L * grid) {
        // Full grid: the workgroups are divided over the strips in proportion to a strip's cost, each group of workgroups
        // walking all clips.  Cost model fitted to the traces: the SIMD with the most conv2 tiles sets the pace of the
        // conv2 phase (T4 tiles), everything else is worth about five tiles ((101,64): 7 + 5 : 6 + 5 -> 134 : 122).
        const int W1 = a.W / 2, nX = (W1 + 15) / 16;
        double cost[8], total = 0;
        for (int q = 0; q < S; ++q) {
            const TrunkStrip g = trunk_strip(a.H, S, q);
            const int nT = (g.R2b - g.R2a) * nX, T4 = (nT + 3) / 4;
            cost[q] = T4 + 5.0;
            total += cost[q];
        }
        int used = 0;
        for (int q = 0; q < S; ++q) {
            int c = (int)(grid * cost[q] / total + 0.5);
            if (c < 1) c = 1;
            if (q == S - 1) c = grid - used;
            used += c;
            aa.wg_end[q] = used;
        }
        if (aa.wg_end[S - 1] != grid || (S > 1 && aa.wg_end[S - 1] <= aa.wg_end[S - 2]))
            for (int q = 0; q < 8; ++q) aa.wg_end[q] = 0;
    }
    if (aa.wg_end[0] == 0) {
        grid -= grid % S;
        if (grid < S) grid = S;
    }
    TB_DISPATCH(aa, grid, lds)
    return hipGetLastError();
}


// ---- BcResNet front on the bf16 matrix cores
size_t bc_front_b_packed_bytes() { return BF_W1F; }
hipError_t launch_bc_front_b_pack(const float* w1, unsigned char* packed, hipStream_t s) {
    hipLaunchKernelGGL(bc_front_pack_kernel, dim3(1), dim3(64), 0, s, w1, packed);
    return hipGetLastError();
}
hipError_t launch_bc_front_b_pack_f16(const float* w1, unsigned char* packed, float sw, hipStream_t s) {
    hipLaunchKernelGGL(bc_front_pack_f16_kernel, dim3(1), dim3(64), 0, s, w1, packed, sw);
    return hipGetLastError();
}
// (the strip height follows the three-term form's LDS need in every arithmetic: how a clip is cut does not depend on the switch)
static size_t bc_front_b_lds(int W, int sh, int rows_dw) {
    const BfGeom g = bf_geom(W, sh, rows_dw);
    return (size_t)BF_HEAD + 3 * (size_t)g.plane_b + (size_t)(g.max_conv + 1) * ((W / 2 + 3) & ~3) * 32 * sizeof(float) + 16;      // (+ 1: P's zero row)
}
// depthwise rows per strip such that two workgroups share a CU (80 KB each); 0 = does not fit
int bc_front_b_rows(int H, int W, int sh) {
    if (H < 4 || W < 4) return 0;
    const int Ho = (H / 2 - 1) / sh + 1;
    int best = 0;
    for (int rows = 1; rows <= Ho; ++rows)
        if (bc_front_b_lds(W, sh, rows) <= 80 * 1024) best = rows;
    return best;
}
hipError_t launch_bc_front_b(const Conv1DwArgs& a0, int products, int max_grid, hipStream_t s) {
    if (!a0.wpack || (products != 6 && products != 9 && products != 3)) return hipErrorInvalidValue;
    if (products == 3 && !(a0.alpha && a0.f16_in > 0.0f && a0.f16_unscale > 0.0f)) return hipErrorInvalidValue;
    Conv1DwArgs a = a0;
    a.rows_dw = bc_front_b_rows(a.H, a.W, a.sh);
    if (a.rows_dw <= 0) return hipErrorInvalidValue;
    const int H1 = a.H / 2, W1 = a.W / 2;
    a.Ho = (H1 - 1) / a.sh + 1; a.Wo = (W1 - 1) / a.sw + 1;
    // the largest strips that fit (measured at (101,64), 8192 clips: 3 / 4 / 5 / 6 depthwise rows per strip 0.636 / 0.589 / 0.569 /
    // 0.556 ms - fewer barriers and less halo recomputation beat evenly sized strips)
    const size_t lds = bc_front_b_lds(a.W, a.sh, a.rows_dw);
    int grid = a.B < max_grid * 2 ? a.B : max_grid * 2;
    if (grid < 1) grid = 1;
    const bool bn = a.alpha != nullptr;
#define BF_LAUNCH(ACTV, PRODV, BNV)                                                                                    \
    {                                                                                                                  \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(bc_front_b_kernel<ACTV, PRODV, BNV>), lds);         \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((bc_front_b_kernel<ACTV, PRODV, BNV>), dim3(grid), dim3(NTHR), lds, s, a);                  \
    }
#define BF_BN(ACTV, PRODV)                                                                                             \
    if (bn) BF_LAUNCH(ACTV, PRODV, true) else BF_LAUNCH(ACTV, PRODV, false)
#define BF_ACT(ACTV)                                                                                                   \
    if (products == 3) BF_LAUNCH(ACTV, 3, true) else if (products == 6) BF_BN(ACTV, 6) else BF_BN(ACTV, 9)
    switch (a.act) {
        case ACT_RELU: BF_ACT(ACT_RELU) break;
        case ACT_GELU: BF_ACT(ACT_GELU) break;
        case ACT_SILU: BF_ACT(ACT_SILU) break;
        default: return hipErrorInvalidValue;
    }
#undef BF_LAUNCH
#undef BF_BN
#undef BF_ACT
    return hipGetLastError();
}
