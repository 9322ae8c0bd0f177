// ffn_x3.hip - the Conformer feed-forward module in ONE kernel on the bf16 matrix cores (exact operand splitting):
//     h <- h + rscale * ( W2 . swish( W1 . LayerNorm(h) + b1 ) + b2 )          (architectures.py:441-470: FeedForwardModule,
//                                                                               half-step residual in ConformerBlock)
// As separate launches (LayerNorm, gemm_x3 linear1+swish, gemm_x3 linear2+res) the 4D-wide hidden activations make a
// round trip through HBM - 476 MB written and 476 MB read per module at the BASELINE batch (206 848 rows x 576) - and
// the K = 144 GEMM spends most of its time in prologues and epilogues: 0.76 ms per module where the matrix pipe needs
// 0.17.  Here the hidden activations never leave the registers.
//
// Everything is computed TRANSPOSED so that the accumulator layout of the first product is already the operand layout
// of the second (v_mfma_f32_32x32x16_bf16: lane (n, half) of the C matrix holds rows 8g + 4 half + q, g, q < 4, of
// column n; lane (n, half) of the B operand holds k = 8 half + e, e < 8, of column n - a permutation of k that the
// other operand, the pre-packed W2, simply follows):
//     Ht [32 hidden x 32 rows] = W1 block [32 x D] . Xt [D x 32 rows]       A = W1 fragments (LDS), B = X fragments (registers)
//     Yt [D x 32 rows]        += W2 block [D x 32 hidden] . swish(Ht + b1)  A = W2 fragments (LDS), B = Ht re-split in registers
// A wave owns 32 rows for the whole kernel: its LayerNorm-ed rows live in registers as 3 x D/16 B fragments (108 VGPRs
// for D = 144), its D x 32 output tile in ceil(D/32) accumulators (80 registers, AGPRs), and it walks the 4D/32 hidden
// blocks; the four waves of a workgroup (128 rows) share each block's weights through LDS - one contiguous, plan-time
// packed 60 KB block (W1 fragments, W2 fragments, b1), double buffered, fetched straight into LDS one block ahead, one
// barrier per block.
// One wave per SIMD (the kernel needs ~300 of the 512 registers a lone wave may use).
// Per 128 rows: 2 x 6 x 128 x D x 4D multiply-adds on the matrix pipe, 2 x 128 x D x 4 bytes of HBM traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
// H2 instances ("f16x3", FfnArgs::h2_x > 0): every operand - the LayerNorm-ed rows, W1, the hidden activations, W2 - times a
// plan-time power of two is split into TWO binary16 terms (split_h2.h) and each product is three v_mfma_f32_32x32x16_f16:
// half the matrix instructions of the six-product form, which runs at the package's sustained matrix rate.
#include "layers.h"
#include "ffn_x3.h"
#include "split_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef NWW_TRACE      // tools/ubench/ffn_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of the main loop's hidden blocks
__device__ unsigned long long g_ffn_trace[4 * 32 * 8];
#define FFN_STAMP(blk, k) if (blockIdx.x == 0 && lane == 0) g_ffn_trace[(wave * 32 + (blk)) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define FFN_STAMP(blk, k)
#endif

namespace {

__device__ __forceinline__ void split3f(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// eight float32 -> three bf16x8 fragments (hi, mid, lo)
__device__ __forceinline__ void split_frag(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3f(v[e], h[e], m[e], l[e]);
    union { uint4 u; bf16x8 b; } ch, cm, cl;
    ch.u = make_uint4(pack16(h[0], h[1]), pack16(h[2], h[3]), pack16(h[4], h[5]), pack16(h[6], h[7]));
    cm.u = make_uint4(pack16(m[0], m[1]), pack16(m[2], m[3]), pack16(m[4], m[5]), pack16(m[6], m[7]));
    cl.u = make_uint4(pack16(l[0], l[1]), pack16(l[2], l[3]), pack16(l[4], l[5]), pack16(l[6], l[7]));
    fh = ch.b; fm = cm.b; fl = cl.b;
}

__device__ __forceinline__ float ffn_swish(float v) {          // v * sigmoid(v) on the hardware exp2 / rcp (as gemm_x3's epilogue)
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// six products, small terms first (the order of gemm_x3.hip): w = weight fragments (A operand), x = activation fragments (B)
__device__ __forceinline__ void mfma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
}

// one MFMA of the split products: bf16 terms or binary16 terms (fragments are carried as 128-bit bags typed bf16x8)
template <bool H2>
__device__ __forceinline__ f32x16 ffn_mma(const bf16x8& w, const bf16x8& x, const f32x16& acc) {
    if (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0);
}
// eight float32 (already scaled) -> two binary16 fragments
__device__ __forceinline__ void split_frag2(const float (&v)[8], bf16x8& fh, bf16x8& fl) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nww_split2h(v[2 * j], v[2 * j + 1], h[j], l[j]);
    union { uint4 u; bf16x8 b; } ch, cl;
    ch.u = make_uint4(h[0], h[1], h[2], h[3]);
    cl.u = make_uint4(l[0], l[1], l[2], l[3]);
    fh = ch.b; fl = cl.b;
}

// ---- plan-time packing: one thread per (hidden block, fragment, lane)
__global__ void __launch_bounds__(256) ffn_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, unsigned char* __restrict__ out, int D, float sw1, float sw2, int perm) {
    const int D16 = D / 16, NOB = (D + 31) / 32, NHB = D / 8, H4 = 4 * D;
    const int frags = D16 + 2 * NOB;
    const int NT = sw1 > 0.0f ? 2 : 3;
    const size_t blk = ffn_x3_block_bytes(D, NT);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)NHB * frags * 64) return;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) % frags), hb = (int)((idx >> 6) / frags);
    const int i = lane & 31, h = lane >> 5;
    unsigned char* base = out + (size_t)hb * blk;
    float v[8];
    unsigned char* dst;
    if (f < D16) {                                             // W1 fragment kb = f: row = hidden 32hb + i, k = 16kb + 8h + e
        const int kb = f;
        // perm: k slot (kb, h, e) carries the feature the ACCUMULATOR layout puts there - tile kb / 2, register group 2 (kb & 1) + (e >> 2)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = W1[(size_t)(32 * hb + i) * D + (perm ? 32 * (kb >> 1) + 8 * (2 * (kb & 1) + (e >> 2)) + 4 * h + (e & 3) : 16 * kb + 8 * h + e)];
        dst = base + ((size_t)(kb * NT) * 64 + lane) * 16;
    } else {                                                   // W2 fragment (ob, kb2): row = out feature 32ob + i, slot e <-> hidden
        const int ob = (f - D16) >> 1, kb2 = (f - D16) & 1;    //   32hb + 8 (2 kb2 + (e >> 2)) + 4h + (e & 3)
        const int m = 32 * ob + i;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = m < D ? W2[(size_t)m * H4 + 32 * hb + 8 * (2 * kb2 + (e >> 2)) + 4 * h + (e & 3)] : 0.0f;
        dst = base + ffn_x3_w1_bytes(D, NT) + ((size_t)((ob * 2 + kb2) * NT) * 64 + lane) * 16;
    }
    if (NT == 2) {
        const float sc = f < D16 ? sw1 : sw2;
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nww_split2h(v[2 * j] * sc, v[2 * j + 1] * sc, hh[j], ll[j]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    } else {
        uint32_t hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3f(v[e], hh[e], mm[e], ll[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack16(hh[0], hh[1]), pack16(hh[2], hh[3]), pack16(hh[4], hh[5]), pack16(hh[6], hh[7]));
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(pack16(mm[0], mm[1]), pack16(mm[2], mm[3]), pack16(mm[4], mm[5]), pack16(mm[6], mm[7]));
        *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(pack16(ll[0], ll[1]), pack16(ll[2], ll[3]), pack16(ll[4], ll[5]), pack16(ll[6], ll[7]));
    }
    if (f == 0 && lane < 32) reinterpret_cast<float*>(base + ffn_x3_w1_bytes(D, NT) + (size_t)NOB * 2 * NT * 1024)[lane] = b1[32 * hb + lane];
}

// prologue Linear W [D][KP] -> tiles [ob][kb][term][lane] of 16-byte fragments: lane (i, h) = row 32 ob + i, k slots 16 kb + 8 h + e
__global__ void __launch_bounds__(256) ffn_pro_pack_kernel(const float* __restrict__ W, unsigned char* __restrict__ out, int D, int KP, float ws) {
    const int KP16 = KP / 16, NOB = (D + 31) / 32;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)NOB * KP16 * 64) return;
    const int lane = (int)(idx & 63), kb = (int)((idx >> 6) % KP16), ob = (int)((idx >> 6) / KP16);
    const int i = lane & 31, h = lane >> 5, m = 32 * ob + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = m < D ? W[(size_t)m * KP + 16 * kb + 8 * h + e] : 0.0f;
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nww_split2h(v[2 * j] * ws, v[2 * j + 1] * ws, hh[j], ll[j]);
    unsigned char* dst = out + (size_t)ob * ffn_x3_pro_tile_bytes(KP) + ((size_t)(kb * 2) * 64 + lane) * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// out [b][c] = (sum of the clip's tile segments) / (scale T): tile t covers rows 32 t .. 32 t + 31, segment 0 = its rows of the clip its
// first row belongs to, segment 1 = its rows of the next clip (T >= 32: a tile touches at most two clips).  msum [tile][segment][plane hi /
// lo][D]: the addends are integers (below 2^24 each), so the order of the float64 sums does not matter.
__global__ void __launch_bounds__(256) ffn_mean_finish_kernel(const float* __restrict__ msum, float* __restrict__ out, int B, int T, int D, double inv) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * D) return;
    const int c = (int)(idx % D);
    const long long b = (long long)(idx / D);
    const long long t0 = (b * T) / 32, t1 = (b * T + T - 1) / 32;
    double hi = 0.0, lo = 0.0;                                 // integers: exact in any order
    for (long long t = t0; t <= t1; ++t) {
        const int seg = ((32 * t) / T == b) ? 0 : 1;
        const float* m = msum + ((size_t)t * 2 + seg) * 2 * D;
        hi += (double)m[c]; lo += (double)m[D + c];
    }
    out[idx] = (float)((hi * 262144.0 + lo) * inv);
}

// (Measured and not kept for the two-term form: 256 rows per workgroup on eight waves, two per SIMD, each running product -
// epilogue - product in turn so that they cover each other's issue stalls - the lone in-order wave here spends a third of its
// cycles in them, SQ_WAIT_INST_ANY 50 M of 146 M.  At D = 144 the 80 output accumulators + 72 X-fragment registers + the
// epilogue's temporaries do not fit 256 registers: 59 spilled, 0.315 ms against 0.320.)
// LATE_A (round 5): the first stage of a block's epilogue - bias, 1 + 2^(-v log2 e): 16 of the 40 pieces, half the transcendentals - runs
// between the MFMAs of the PREVIOUS block's second product, on the accumulator the first product has just finished; the rest stays between the
// MFMAs of the next first product.  (tools/ubench/ffn_trace: with all 40 pieces behind the first product's 27 MFMAs that phase took 2200 clocks,
// VALU-bound, and the second product's 30 MFMAs 1250 with the VALU idle.)  One accumulator instead of two; the next block's biases come from the
// packed blob in global memory (its LDS copy is still in flight then).
// KP16 > 0 (round 6): a row-local Linear of K = 16 KP16 inputs in front (FfnArgs::px ..; PRES: plus the rows of h), its result h0 in the
// accumulator layout is the module's input AND the start value of the output accumulators (h0 / (rscale ik2): the residual rides through
// the second product, nothing is re-read at the end).  EPI (round 6): the updated rows are not stored - LayerNorm, then exact per-tile,
// per-clip sums (FfnArgs::msum).
template <int D16, bool H2, int KP16 = 0, bool PRES = false, bool EPI = false>
__global__ void __launch_bounds__(256) ffn_x3_kernel(FfnArgs a) {
    static_assert(KP16 == 0 || H2, "the prologue product exists in the two-term form only");
    constexpr int D = 16 * D16, NOB = (D + 31) / 32, NHB = D / 8;
    constexpr int NT = H2 ? 2 : 3, NP = H2 ? 3 : 6;            // terms per value, partial products per operand pair
    constexpr int W1_PART = (D16 * NT * 1024 + 4095) & ~4095, W2_PART = (NOB * 2 * NT * 1024 + 128 + 4095) & ~4095, BLK = W1_PART + W2_PART;
    constexpr int B1_OFF = NOB * 2 * NT * 1024;                // the block's 32 biases sit behind the W2 fragments
    // two-term form: accumulators of the first product are h2_x h2_w1 times the true sums, of the second h2_h h2_w2 times
    const float s_x = H2 ? a.h2_x : 1.0f, ik1 = H2 ? 1.0f / (a.h2_x * a.h2_w1) : 1.0f, s_h = H2 ? a.h2_h : 1.0f, ik2 = H2 ? 1.0f / (a.h2_h * a.h2_w2) : 1.0f;
    // Four separate LDS objects, not four quarters of one: hipcc then knows that the reads of one buffer cannot alias
    // the LDS-DMA writes into another and does not wait for the fetches in flight (s_waitcnt vmcnt(0)) in mid-block.
    __shared__ __attribute__((aligned(16))) unsigned char w1b0[W1_PART];
    __shared__ __attribute__((aligned(16))) unsigned char w1b1[W1_PART];
    __shared__ __attribute__((aligned(16))) unsigned char w2b0[W2_PART];
    __shared__ __attribute__((aligned(16))) unsigned char w2b1[W2_PART];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int row = (int)blockIdx.x * 128 + wave * 32 + n;
    const bool row_ok = row < a.M;
    float* hrow = a.h + (size_t)(row_ok ? row : a.M - 1) * D;

    // ---- weights go global -> LDS directly (global_load_lds_dwordx4: lane i of a wave lands at base + 16 i), no staging
    // registers; the first parts are on their way while the rows are normalised
    auto fetch = [&](const unsigned char* src, unsigned char* buf, auto steps) {
        const unsigned char* sp = src + tid * 16;
        unsigned char* dst = buf + wave * 1024;                          // wave-uniform
#pragma unroll
        for (int j = 0; j < decltype(steps)::value; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sp + j * 4096),
                                             (void __attribute__((address_space(3)))*)(dst + j * 4096), 16, 0, 0);
    };
    // one 4 KB step of a part (the main loop issues a block's steps one at a time between the MFMAs of the first product: issued together
    // at the block's top they held every wave for ~650 clocks - 44 KB through the CU's 64 B/clk path - with the matrix pipe idle: tools/ubench/ffn_trace)
    auto fetch_step = [&](const unsigned char* src, unsigned char* buf, int j) {
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + tid * 16 + j * 4096),
                                         (void __attribute__((address_space(3)))*)(buf + wave * 1024 + j * 4096), 16, 0, 0);
    };
    constexpr int NF1 = W1_PART / 4096, NF2 = W2_PART / 4096;
    struct FetchPlan { const unsigned char* s1; unsigned char* d1; const unsigned char* s2; unsigned char* d2; };
    auto plan_of = [&](int hb1, unsigned char* b1, int hb2, unsigned char* b2) {      // W1 part of block hb1 -> b1, W2 part of block hb2 -> b2 (-1: none)
        return FetchPlan{hb1 >= 0 ? a.packed + (size_t)hb1 * BLK : nullptr, b1, hb2 >= 0 ? a.packed + (size_t)hb2 * BLK + W1_PART : nullptr, b2};
    };
    auto fetch_w1 = [&](int hb, unsigned char* buf) { fetch(a.packed + (size_t)hb * BLK, buf, std::integral_constant<int, W1_PART / 4096>{}); };
    auto fetch_w2 = [&](int hb, unsigned char* buf) { fetch(a.packed + (size_t)hb * BLK + W1_PART, buf, std::integral_constant<int, W2_PART / 4096>{}); };
    bf16x8 xf[D16][NT];
    f32x16 yacc[NOB];
    if constexpr (KP16 == 0) {
    FFN_STAMP(30, 0)
    fetch_w1(0, w1b0);
    fetch_w2(0, w2b0);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.0f;

    // ---- LayerNorm of the lane's half row (features 16kb + 8h + e) -> X fragments
    {
        float v[D16][8];
        float s = 0.0f;
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(hrow + 16 * kb + 8 * h);
            const float4 p1 = *reinterpret_cast<const float4*>(hrow + 16 * kb + 8 * h + 4);
            v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
            v[kb][4] = p1.x; v[kb][5] = p1.y; v[kb][6] = p1.z; v[kb][7] = p1.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[kb][e];
        }
        s += __shfl_xor(s, 32, 64);
        const float mu = s / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int kb = 0; kb < D16; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[kb][e] - mu; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + 16 * kb + 8 * h), w1 = *reinterpret_cast<const float4*>(a.ln_w + 16 * kb + 8 * h + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(a.ln_b + 16 * kb + 8 * h), c1 = *reinterpret_cast<const float4*>(a.ln_b + 16 * kb + 8 * h + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (v[kb][e] - mu) * rstd * w[e] + c[e];
            if constexpr (H2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] *= s_x;
                split_frag2(y, xf[kb][0], xf[kb][1]);
            } else {
                split_frag(y, xf[kb][0], xf[kb][1], xf[kb][NT - 1]);
            }
        }
    }

    } else {
        // ---- prologue product: h0 = [h +] (px . Wp^T + pb) in the accumulator layout (lane (row n, half): features 32 ob + 8 g + 4 half + q)
        constexpr int KP = 16 * KP16, PT = (KP16 * 2 * 1024 + 4095) & ~4095, PSTEPS = PT / 4096;
        constexpr int NPB = W2_PART / PT;                      // tiles that fit one W2 buffer
        constexpr bool SMALL = 2 * NPB >= NOB;                 // all tiles beside W1(0): K = 64; else one tile per weight buffer, the fifth behind the first
        static_assert(SMALL || (PT <= W1_PART && NOB <= 5), "prologue tiles do not fit the weight buffers");
        using PS = std::integral_constant<int, PSTEPS>;
        if constexpr (SMALL) {
            fetch_w1(0, w1b0);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) fetch(a.ppacked + (size_t)ob * PT, (ob < NPB ? w2b0 : w2b1) + (ob % NPB) * PT, PS{});
        } else {
            fetch(a.ppacked, w1b0, PS{}); fetch(a.ppacked + (size_t)PT, w1b1, PS{});
            fetch(a.ppacked + (size_t)2 * PT, w2b0, PS{}); fetch(a.ppacked + (size_t)3 * PT, w2b1, PS{});
        }
        FFN_STAMP(30, 0)
        const float* prow = a.px + (size_t)(row_ok ? row : a.M - 1) * KP;
        bf16x8 pf[KP16][2];
        float ppin;
        {
            float v[KP16][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KP16; ++kb) {
                const float4 p0 = *reinterpret_cast<const float4*>(prow + 16 * kb + 8 * h);
                const float4 p1 = *reinterpret_cast<const float4*>(prow + 16 * kb + 8 * h + 4);
                v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
                v[kb][4] = p1.x; v[kb][5] = p1.y; v[kb][6] = p1.z; v[kb][7] = p1.w;
            }
#pragma unroll
            for (int kb = 0; kb < KP16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[kb][e]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            // the row's own power of two (largest element into [2^14, 2^15): lin_x3.hip)
            const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
            const float sc = __uint_as_float((268u - eb) << 23);
            ppin = __uint_as_float((eb - 14u) << 23) * a.p_un;
#pragma unroll
            for (int kb = 0; kb < KP16; ++kb) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = v[kb][e] * sc;
                split_frag2(y, pf[kb][0], pf[kb][1]);
            }
        }
        // the residual rows in the accumulator layout (16-byte pieces of 64 rows per load: issued here, used behind the products)
        float4 pres[NOB][4];
        if constexpr (PRES) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (32 * ob + 8 * g < D) pres[ob][g] = *reinterpret_cast<const float4*>(hrow + 32 * ob + 8 * g + 4 * h);
        }
        f32x16 hc[NOB];
        auto ptile = [&](const unsigned char* buf, f32x16& acc) {
            const unsigned char* wp = buf + lane * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KP16; ++kb) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wp + (kb * 2) * 1024), wl = *reinterpret_cast<const bf16x8*>(wp + (kb * 2 + 1) * 1024);
                acc = ffn_mma<true>(wl, pf[kb][0], acc);
                acc = ffn_mma<true>(wh, pf[kb][1], acc);
                acc = ffn_mma<true>(wh, pf[kb][0], acc);
            }
        };
        FFN_STAMP(30, 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        FFN_STAMP(30, 2)
        if constexpr (SMALL) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) ptile((ob < NPB ? w2b0 : w2b1) + (ob % NPB) * PT, hc[ob]);
            __syncthreads();                                   // everyone has left the W2 buffers
            fetch_w2(0, w2b0);
        } else {
            ptile(w1b0, hc[0]);
            __syncthreads();
            if constexpr (NOB > 4) fetch(a.ppacked + (size_t)4 * PT, w1b0, PS{});      // lands under the next three tiles
            ptile(w1b1, hc[1]);
            if constexpr (NOB > 2) ptile(w2b0, hc[2]);
            if constexpr (NOB > 3) ptile(w2b1, hc[3]);
            if constexpr (NOB > 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ptile(w1b0, hc[4]);
            }
            __syncthreads();
            fetch_w1(0, w1b0);
            fetch_w2(0, w2b0);
        }
        FFN_STAMP(30, 3)
        // h0 = acc / (row scale x weight scale) + bias [+ h]; LayerNorm over the row (this lane's 16 x NOB slots, the partner half's)
        float s = 0.0f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (32 * ob + 8 * g < D) {
                    const float4 pb = *reinterpret_cast<const float4*>(a.pb + 32 * ob + 8 * g + 4 * h);
                    float v0 = fmaf(hc[ob][4 * g], ppin, pb.x), v1 = fmaf(hc[ob][4 * g + 1], ppin, pb.y);
                    float v2 = fmaf(hc[ob][4 * g + 2], ppin, pb.z), v3 = fmaf(hc[ob][4 * g + 3], ppin, pb.w);
                    if constexpr (PRES) { v0 = pres[ob][g].x + v0; v1 = pres[ob][g].y + v1; v2 = pres[ob][g].z + v2; v3 = pres[ob][g].w + v3; }
                    hc[ob][4 * g] = v0; hc[ob][4 * g + 1] = v1; hc[ob][4 * g + 2] = v2; hc[ob][4 * g + 3] = v3;
                    s += (v0 + v1) + (v2 + v3);
                }
        s += __shfl_xor(s, 32, 64);
        const float mu = s / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * ob + 8 * (r >> 2) < D) { const float d = hc[ob][r] - mu; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
        const float yi = 1.0f / (a.rscale * ik2);              // the residual enters the output accumulators (a power of two when rscale is)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                if (32 * ob + 16 * gp < D) {                   // k-block 2 ob + gp = registers 8 gp .. 8 gp + 7 of tile ob (D % 16 == 0)
                    float y[8];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = 2 * gp + gg;
                        const float4 w = *reinterpret_cast<const float4*>(a.ln_w + 32 * ob + 8 * g + 4 * h), c = *reinterpret_cast<const float4*>(a.ln_b + 32 * ob + 8 * g + 4 * h);
                        y[4 * gg] = ((hc[ob][4 * g] - mu) * rstd * w.x + c.x) * s_x; y[4 * gg + 1] = ((hc[ob][4 * g + 1] - mu) * rstd * w.y + c.y) * s_x;
                        y[4 * gg + 2] = ((hc[ob][4 * g + 2] - mu) * rstd * w.z + c.z) * s_x; y[4 * gg + 3] = ((hc[ob][4 * g + 3] - mu) * rstd * w.w + c.w) * s_x;
                    }
                    split_frag2(y, xf[2 * ob + gp][0], xf[2 * ob + gp][1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ob][r] = hc[ob][r] * yi;
        }
        FFN_STAMP(30, 4)
    }

    // Software pipeline over the hidden blocks (W1 parts and W2 parts double buffered separately, each fetched a full iteration before its
    // first use): the first product of block hb + 1 runs with the SECOND half of block hb's epilogue (y = v / u, split) between its MFMAs,
    // the second product of block hb with the FIRST half of block hb + 1's (bias, 1 + 2^(-v log2 e)) between its own - a lone wave issues in
    // order, and each MFMA of a dependent chain holds it for 32 clocks.  (The round-4 schedule - the whole epilogue behind the first product -
    // was removed in round 6: tools/ubench/ffn_trace, 4130 -> 3450 clocks per block.)
    // ---- LATE_A schedule
    float va[16], ua[16];                                      // stage-A results of the block whose fragments the next first product builds
    auto load_bias = [&](int hb, float (&bias)[16]) {          // b1 of block hb, hidden units 8 g + 4 h + 0..3, from the packed blob
        const float* b1p = reinterpret_cast<const float*>(a.packed + (size_t)hb * BLK + W1_PART + B1_OFF) + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 q = *reinterpret_cast<const float4*>(b1p + 8 * g);
            bias[4 * g] = q.x; bias[4 * g + 1] = q.y; bias[4 * g + 2] = q.z; bias[4 * g + 3] = q.w;
        }
    };
    auto piece_a = [&](int e, const f32x16& acc, const float (&bias)[16]) {
        va[e] = H2 ? fmaf(acc[e], ik1, bias[e]) : acc[e] + bias[e];
        ua[e] = 1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * va[e]);
        asm volatile("" : "+v"(va[e]), "+v"(ua[e]));
    };
    // first product of the NEXT block (acc = W1 . Xt) with the rest of THIS block's epilogue (va, ua -> hf) between its MFMAs
    auto phase1 = [&](auto has_next, const unsigned char* w1buf, f32x16& acc, bf16x8 (&hf)[2][NT], const FetchPlan fp) {
        constexpr bool NEXT = decltype(has_next)::value;
        constexpr int NSLOT = NEXT ? NP * D16 : 1;
        constexpr int NPIECE = H2 ? 24 : 32;
        constexpr int PER = (NPIECE + NSLOT - 1) / NSLOT;
        const unsigned char* w1p = w1buf + lane * 16;
        bf16x8 nw[NT], cw[NT];
        if (NEXT) {
#pragma unroll
            for (int t = 0; t < NT; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(w1p + t * 1024);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t th[16], tm[16], tl[16];
        auto piece = [&](int p) {
            if constexpr (H2) {
                const int j = p / 3, st = p - 3 * j;
                if (st < 2) {
                    const int e = 2 * j + st;
                    va[e] = (va[e] * s_h) * __builtin_amdgcn_rcpf(ua[e]);
                    asm volatile("" : "+v"(va[e]));
                } else {
                    nww_split2h(va[2 * j], va[2 * j + 1], th[j], tl[j]);
                    asm volatile("" : "+v"(th[j]), "+v"(tl[j]));
                }
            } else {
                const int e = p / 2, st = p - 2 * e;
                if (st == 0) {
                    const float y = va[e] * __builtin_amdgcn_rcpf(ua[e]);
                    th[e] = __float_as_uint(y) & 0xffff0000u;
                    ua[e] = y - __uint_as_float(th[e]);
                    asm volatile("" : "+v"(th[e]), "+v"(ua[e]));
                } else {
                    tm[e] = __float_as_uint(ua[e]) & 0xffff0000u;
                    tl[e] = __float_as_uint(ua[e] - __uint_as_float(tm[e]));
                    asm volatile("" : "+v"(tm[e]), "+v"(tl[e]));
                }
            }
        };
        constexpr int PW[6] = {H2 ? 1 : 1, H2 ? 0 : 2, 0, 1, 0, 0}, PX[6] = {H2 ? 0 : 1, H2 ? 1 : 0, H2 ? 0 : 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            if (NEXT) {
                const int kb = q / NP, m = q - NP * kb;
                if (m == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) cw[t] = nw[t];
                }
                if (m < NT && kb + 1 < D16) nw[m] = *reinterpret_cast<const bf16x8*>(w1p + ((kb + 1) * NT + m) * 1024);
                acc = ffn_mma<H2>(cw[PW[m]], xf[kb][PX[m]], acc);
                asm volatile("" : "+a"(acc));
                // the fetches of the blocks ahead: one per slot from the block's first MFMA on.  (Spread evenly over all 27 slots, as in
                // round 5, the last piece left ~1600 clocks before the block's end and the wave then waited ~300 for it at vmcnt(0): one per slot
                // -2.5 % plain / with the K = 64 prologue, -4.5 % with the K = 144 prologue + epilogue; tools/ubench/ffn_trace without -DNWW_TRACE -
                // with the stamps' own stores in vmcnt the trace cannot see this wait.)
                constexpr int FS = (NF1 + NF2 + 1) < NSLOT ? (NF1 + NF2 + 1) : NSLOT;
                const int j0 = (q < FS ? q : FS) * (NF1 + NF2) / FS, j1 = (q + 1 < FS ? q + 1 : FS) * (NF1 + NF2) / FS;
                for (int j = j0; j < j1; ++j) {
                    if (j < NF1) { if (fp.s1) fetch_step(fp.s1, fp.d1, j); }
                    else if (fp.s2) fetch_step(fp.s2, fp.d2, j - NF1);
                }
            }
#pragma unroll
            for (int p = q * PER; p < (q + 1) * PER && p < NPIECE; ++p) piece(p);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
            if constexpr (H2) {
                union { uint4 q; bf16x8 b; } ch, cl;
                const int o = 4 * kb2;
                ch.q = make_uint4(th[o], th[o + 1], th[o + 2], th[o + 3]);
                cl.q = make_uint4(tl[o], tl[o + 1], tl[o + 2], tl[o + 3]);
                hf[kb2][0] = ch.b; hf[kb2][1] = cl.b;
            } else {
                union { uint4 q; bf16x8 b; } ch, cm, cl;
                const int o = 8 * kb2;
                ch.q = make_uint4(pack16(th[o], th[o + 1]), pack16(th[o + 2], th[o + 3]), pack16(th[o + 4], th[o + 5]), pack16(th[o + 6], th[o + 7]));
                cm.q = make_uint4(pack16(tm[o], tm[o + 1]), pack16(tm[o + 2], tm[o + 3]), pack16(tm[o + 4], tm[o + 5]), pack16(tm[o + 6], tm[o + 7]));
                cl.q = make_uint4(pack16(tl[o], tl[o + 1]), pack16(tl[o + 2], tl[o + 3]), pack16(tl[o + 4], tl[o + 5]), pack16(tl[o + 6], tl[o + 7]));
                hf[kb2][0] = ch.b; hf[kb2][1] = cm.b; hf[kb2][NT - 1] = cl.b;
            }
        }
    };
    // second product of this block (Yt += W2 . hf) with stage A of the NEXT block (its accumulator is complete) between its MFMAs
    auto phase2 = [&](auto has_a, const unsigned char* w2buf, const bf16x8 (&hf)[2][NT], const f32x16& acc, const float (&bias)[16]) {
        constexpr bool HAS_A = decltype(has_a)::value;
        const unsigned char* w2p = w2buf + lane * 16;
        constexpr int PW[6] = {H2 ? 1 : 1, H2 ? 0 : 2, 0, 1, 0, 0}, PX[6] = {H2 ? 0 : 1, H2 ? 1 : 0, H2 ? 0 : 2, 0, 1, 0};
        constexpr int NG = 2 * ((NOB + 1) / 2);
        constexpr int NS2 = NG * NP, PER_A = (16 + NS2 - 1) / NS2;      // slots of the second product, stage-A pieces per slot
        auto frag = [&](int g, int which, int t) {
            const int ob = 2 * (g >> 1) + which, kb2 = g & 1;
            return *reinterpret_cast<const bf16x8*>(w2p + ((ob * 2 + kb2) * NT + t) * 1024);
        };
        bf16x8 na[NT], nb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { na[t] = frag(0, 0, t); if (NOB > 1) nb[t] = frag(0, 1, t); }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ob = 2 * (g >> 1), kb2 = g & 1;
            const bool two = ob + 1 < NOB;
            bf16x8 ca[NT], cb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
            if (g + 1 < NG) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    na[t] = frag(g + 1, 0, t);
                    if (2 * ((g + 1) >> 1) + 1 < NOB) nb[t] = frag(g + 1, 1, t);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                yacc[ob] = ffn_mma<H2>(ca[PW[m]], hf[kb2][PX[m]], yacc[ob]);
                if (two) yacc[ob + 1] = ffn_mma<H2>(cb[PW[m]], hf[kb2][PX[m]], yacc[ob + 1]);
                if constexpr (HAS_A) {
                    const int sl = g * NP + m;
#pragma unroll
                    for (int e = sl * PER_A; e < (sl + 1) * PER_A && e < 16; ++e) piece_a(e, acc, bias);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16 accA;
    bf16x8 hf[2][NT];
    float4 res[NOB][4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fetch_w1(1, w1b1);
    {                                                          // Ht(0), not overlapped
        const unsigned char* w1p = w1b0 + lane * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            bf16x8 cw[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) cw[t] = *reinterpret_cast<const bf16x8*>(w1p + (kb * NT + t) * 1024);
            if constexpr (H2) {
                accA = ffn_mma<true>(cw[1], xf[kb][0], accA);
                accA = ffn_mma<true>(cw[0], xf[kb][1], accA);
                accA = ffn_mma<true>(cw[0], xf[kb][0], accA);
            } else {
                const bf16x8 c3[3] = {cw[0], cw[1], cw[NT - 1]}, x3[3] = {xf[kb][0], xf[kb][1], xf[kb][NT - 1]};
                mfma6(c3, x3, accA);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        // Even block hb: W1(hb + 1) in w1b1, W2(hb) in w2b0; odd block: the other buffers.  accA holds Ht(hb + 1) from phase 1 to the end of phase 2.
        float bias[16];
        load_bias(0, bias);
#pragma unroll
        for (int e = 0; e < 16; ++e) piece_a(e, accA, bias);  // stage A of block 0, not overlapped
        for (int hb = 0; hb + 2 < NHB; hb += 2) {
            FFN_STAMP(hb, 0)
            load_bias(hb + 1, bias);
            __builtin_amdgcn_sched_barrier(0);
            FFN_STAMP(hb, 1)
            phase1(std::true_type{}, w1b1, accA, hf, plan_of(hb + 2, w1b0, hb + 1, w2b1));
            __builtin_amdgcn_sched_barrier(0);
            FFN_STAMP(hb, 2)
            phase2(std::true_type{}, w2b0, hf, accA, bias);
            FFN_STAMP(hb, 3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FFN_STAMP(hb, 4)
            __syncthreads();
            FFN_STAMP(hb, 5)
            load_bias(hb + 2, bias);
            __builtin_amdgcn_sched_barrier(0);
            phase1(std::true_type{}, w1b0, accA, hf, plan_of(hb + 3, w1b1, hb + 2, w2b0));
            __builtin_amdgcn_sched_barrier(0);
            phase2(std::true_type{}, w2b1, hf, accA, bias);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        {                                                      // the last two blocks
            load_bias(NHB - 1, bias);
            __builtin_amdgcn_sched_barrier(0);
            phase1(std::true_type{}, w1b1, accA, hf, plan_of(-1, nullptr, NHB - 1, w2b1));
            __builtin_amdgcn_sched_barrier(0);
            phase2(std::true_type{}, w2b0, hf, accA, bias);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // the X fragments are dead now: their registers take the residual rows for the final update (prologue instances: the residual is
            // already inside the accumulators)
            if constexpr (KP16 == 0) {
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (32 * ob + 8 * g < D) res[ob][g] = *reinterpret_cast<const float4*>(hrow + 32 * ob + 8 * g + 4 * h);
            }
            __builtin_amdgcn_sched_barrier(0);
            phase1(std::false_type{}, w1b0, accA, hf, plan_of(-1, nullptr, -1, nullptr));
            __builtin_amdgcn_sched_barrier(0);
            phase2(std::false_type{}, w2b1, hf, accA, bias);
        }
    }

    FFN_STAMP(31, 0)
    // ---- h <- h + rscale * (Yt + b2): lane (row n, half h) holds out features 32 ob + 8 g + 4 h + 0..3
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = 32 * ob + 8 * g + 4 * h;
            if (32 * ob + 8 * g < D) {                         // D % 16 == 0, so the four features (and both halves) are in or out together
                const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + m);
                if (H2) {                                      // the accumulators back to the true scale (a power of two: exact)
                    yacc[ob][4 * g + 0] *= ik2; yacc[ob][4 * g + 1] *= ik2; yacc[ob][4 * g + 2] *= ik2; yacc[ob][4 * g + 3] *= ik2;
                }
                float4 r;
                if constexpr (KP16 == 0) {
                    r = res[ob][g];
                    r.x += a.rscale * (yacc[ob][4 * g + 0] + b2.x);
                    r.y += a.rscale * (yacc[ob][4 * g + 1] + b2.y);
                    r.z += a.rscale * (yacc[ob][4 * g + 2] + b2.z);
                    r.w += a.rscale * (yacc[ob][4 * g + 3] + b2.w);
                } else {                                       // h0 rode through the accumulators: rscale (h0 / rscale + Y + b2)
                    r.x = a.rscale * (yacc[ob][4 * g + 0] + b2.x);
                    r.y = a.rscale * (yacc[ob][4 * g + 1] + b2.y);
                    r.z = a.rscale * (yacc[ob][4 * g + 2] + b2.z);
                    r.w = a.rscale * (yacc[ob][4 * g + 3] + b2.w);
                }
                if constexpr (!EPI) {
                    if (row_ok) *reinterpret_cast<float4*>(hrow + m) = r;
                } else {
                    yacc[ob][4 * g + 0] = r.x; yacc[ob][4 * g + 1] = r.y; yacc[ob][4 * g + 2] = r.z; yacc[ob][4 * g + 3] = r.w;
                }
            }
        }
    FFN_STAMP(31, 1)
    if constexpr (EPI) {
        // ---- LayerNorm of the updated rows, then per tile and clip segment the column sums - staged through the wave's own (dead) weight
        // buffer as [32 rows][D + 4] float32 (16-byte stores conflict-free at this pitch), summed by lane = column in row order
        constexpr int GP = D + 4;
        static_assert(32 * GP * 4 <= W1_PART, "the staging tile must fit a weight buffer");
        float sm = 0.0f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * ob + 8 * (r >> 2) < D) sm += yacc[ob][r];
        sm += __shfl_xor(sm, 32, 64);
        const float mu2 = sm / (float)D;
        float q2 = 0.0f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * ob + 8 * (r >> 2) < D) { const float d = yacc[ob][r] - mu2; q2 = fmaf(d, d, q2); }
        q2 += __shfl_xor(q2, 32, 64);
        const float rstd2 = 1.0f / sqrtf(q2 / (float)D + 1e-5f);
        FFN_STAMP(31, 3)
        __syncthreads();                                       // every wave has left the weight buffers
        FFN_STAMP(31, 4)
        float* st = reinterpret_cast<float*>(wave == 0 ? w1b0 : wave == 1 ? w1b1 : wave == 2 ? w2b0 : w2b1);
        // Exact sums in float32: y x scale (|.| <= 2^36) = hi 2^18 + lo, hi = rint(y scale 2^-18) (|hi| <= 2^18), lo = rint of the exact remainder
        // x 2^18 (|lo| <= 2^17) - integers whose sums over a tile's 32 rows stay below 2^24, i.e. exact in float32 whatever the order.  The lane
        // converts its own elements once; the hi plane and then the lo plane are staged and summed by lane = column.  (First versions: float64
        // rint / add per element in the column loop 12 k clocks per tile, int32 the same - the conversions, not the adds: tools/ubench/ffn_trace.)
        const float sc1 = a.m_scale * (1.0f / 262144.0f);
        const long long r0 = (long long)blockIdx.x * 128 + wave * 32;      // the tile's first row
        const int n0 = (int)min((long long)32, (long long)a.T - r0 % a.T);  // rows of the clip the first row belongs to
        const int nv = (int)max((long long)0, min((long long)32, (long long)a.M - r0));      // rows that exist
        const int e0 = min(n0, nv);
        constexpr int NC = (D + 63) / 64;
        float sums[2][2][NC];                                  // [plane hi / lo][segment][column]
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (32 * ob + 8 * g < D) {
                        const int m = 32 * ob + 8 * g + 4 * h;
                        float o[4];
                        if (pl == 0) {
                            const float4 w = *reinterpret_cast<const float4*>(a.ln2_w + m), c = *reinterpret_cast<const float4*>(a.ln2_b + m);
                            const float wv[4] = {w.x, w.y, w.z, w.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float t = ((yacc[ob][4 * g + q] - mu2) * rstd2 * wv[q] + cv[q]) * sc1, hf = __builtin_rintf(t);
                                o[q] = hf;
                                yacc[ob][4 * g + q] = __builtin_rintf((t - hf) * 262144.0f);      // t - hf is exact; kept for the second plane
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] = yacc[ob][4 * g + q];
                        }
                        *reinterpret_cast<float4*>(st + n * GP + m) = make_float4(o[0], o[1], o[2], o[3]);
                    }
            __builtin_amdgcn_wave_barrier();                   // (one wave: its LDS operations execute in order)
#pragma unroll
            for (int k = 0; k < NC; ++k) { sums[pl][0][k] = 0.0f; sums[pl][1][k] = 0.0f; }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const bool in0 = j < e0, in1 = j >= e0 && j < nv;      // (wave-uniform)
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const float v = (lane + 64 * k < D) ? st[j * GP + lane + 64 * k] : 0.0f;
                    sums[pl][0][k] += in0 ? v : 0.0f;
                    sums[pl][1][k] += in1 ? v : 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        FFN_STAMP(31, 6)
        // (the two planes stay float32 here: a handful of float64 conversions per lane cost this epilogue another 6 k clocks; the finish kernel combines them)
        float* mrow = a.msum + (size_t)(blockIdx.x * 4 + wave) * 4 * D;      // [segment][plane][D]
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (lane + 64 * k < D) {
                mrow[lane + 64 * k] = sums[0][0][k]; mrow[D + lane + 64 * k] = sums[1][0][k];
                mrow[2 * D + lane + 64 * k] = sums[0][1][k]; mrow[3 * D + lane + 64 * k] = sums[1][1][k];
            }
        FFN_STAMP(31, 2)
    }
}

}  // namespace

size_t ffn_x3_packed_bytes(int D) { return (size_t)(D / 8) * ffn_x3_block_bytes(D); }

// D = 192 / 256 (round 6): the two-term form only - its weight blocks (64 KB at 256, double buffered) and 460 registers fit one workgroup per
// CU; the three-term form's 96 KB blocks do not fit the LDS
bool ffn_x3_supported(int D, bool h2) { return D == 32 || D == 64 || D == 96 || D == 128 || D == 144 || (h2 && (D == 192 || D == 256)); }
// prologue / epilogue instances are compiled for the BASELINE width only (d_model 144; K = 64 log-mel bins or K = d_model)
bool ffn_x3_pro_supported(int D, int KP) { return D == 144 && (KP == 64 || KP == 144); }

hipError_t launch_ffn_x3_pack(const float* W1, const float* b1, const float* W2, void* out, int D, hipStream_t s, float sw1, float sw2, int perm) {
    const size_t total = (size_t)(D / 8) * (D / 16 + 2 * ((D + 31) / 32)) * 64;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W1, b1, W2,
                       reinterpret_cast<unsigned char*>(out), D, sw1, sw2, perm);
    return hipGetLastError();
}

hipError_t launch_ffn_x3_pro_pack(const float* W, void* out, int D, int KP, float ws, hipStream_t s) {
    if (!ffn_x3_pro_supported(D, KP) || !(ws > 0.0f)) return hipErrorInvalidValue;
    const size_t total = (size_t)((D + 31) / 32) * (KP / 16) * 64;
    hipLaunchKernelGGL(ffn_pro_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, reinterpret_cast<unsigned char*>(out), D, KP, ws);
    return hipGetLastError();
}

size_t ffn_x3_msum_bytes(int M, int D) { return (size_t)((M + 127) / 128) * 4 * 4 * D * sizeof(float); }

hipError_t launch_ffn_x3_mean_finish(const float* msum, float* out, int B, int T, int D, float m_scale, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (T < 32 || !(m_scale > 0.0f)) return hipErrorInvalidValue;
    const size_t total = (size_t)B * D;
    hipLaunchKernelGGL(ffn_mean_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, msum, out, B, T, D, 1.0 / ((double)m_scale * (double)T));
    return hipGetLastError();
}

hipError_t launch_ffn_x3(const FfnArgs& a, int D, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    const dim3 grid((a.M + 127) / 128);
    const bool epi = a.msum != nullptr;
    if (a.pro_k > 0 || epi) {                                  // round-6 instances: d_model 144, two-term form
        if (D != 144 || !(a.h2_x > 0.0f) || (a.pro_k > 0 && (!ffn_x3_pro_supported(D, a.pro_k) || !a.px || !a.ppacked || !a.pb)) ||
            (epi && (a.T < 32 || !(a.m_scale > 0.0f) || !a.ln2_w || !a.ln2_b)) || (a.pro_k == 64 && a.pro_res)) return hipErrorInvalidValue;
        if (a.pro_k == 64) {
            if (epi) hipLaunchKernelGGL((ffn_x3_kernel<9, true, 4, false, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((ffn_x3_kernel<9, true, 4, false, false>), grid, dim3(256), 0, s, a);
        } else if (a.pro_k == 144) {
            if (a.pro_res) {
                if (epi) hipLaunchKernelGGL((ffn_x3_kernel<9, true, 9, true, true>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((ffn_x3_kernel<9, true, 9, true, false>), grid, dim3(256), 0, s, a);
            } else return hipErrorInvalidValue;
        } else {
            hipLaunchKernelGGL((ffn_x3_kernel<9, true, 0, false, true>), grid, dim3(256), 0, s, a);
        }
        return hipGetLastError();
    }
#define FFN_GO(D16V)                                                                                               \
    {                                                                                                              \
        if (a.h2_x > 0.0f) hipLaunchKernelGGL((ffn_x3_kernel<D16V, true>), grid, dim3(256), 0, s, a);              \
        else hipLaunchKernelGGL((ffn_x3_kernel<D16V, false>), grid, dim3(256), 0, s, a);                           \
    }
#define FFN_GO_H2(D16V)                                                                                            \
    {                                                                                                              \
        if (!(a.h2_x > 0.0f)) return hipErrorInvalidValue;                                                         \
        hipLaunchKernelGGL((ffn_x3_kernel<D16V, true>), grid, dim3(256), 0, s, a);                                 \
    }
    switch (D) {
        case 32: FFN_GO(2) break;
        case 64: FFN_GO(4) break;
        case 96: FFN_GO(6) break;
        case 128: FFN_GO(8) break;
        case 144: FFN_GO(9) break;
        case 192: FFN_GO_H2(12) break;
        case 256: FFN_GO_H2(16) break;
        default: return hipErrorInvalidValue;
    }
#undef FFN_GO_H2
#undef FFN_GO
    return hipGetLastError();
}
