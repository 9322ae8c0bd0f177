// attn_x3.hip - the Conformer's attention module in ONE launch on the binary16 matrix cores (two-term operand splitting, "f16x3"):
//     h <- h + out_proj( softmax(q k^T / sqrt(dh)) v ),   q, k, v = in_proj(h)      (architectures.py:471-493; no pre-LN, :512-513)
// As three launches (lin_x3 in_proj -> mha_h2 -> lin_x3 out_proj + residual) the head-major q / k / v make a round trip through HBM -
// 357 MB written and read back per block at the BASELINE batch (2048 clips x 101 rows x 432) - and every launch re-stages rows it
// does not keep: 0.155 + 0.171 + 0.070 ms.  Here a workgroup owns one clip (58 KB of rows): its four waves hold 32 rows each as
// B / A fragments in registers for the whole clip, q / k / v of ONE head at a time are formed from them, and only h is read and written.
//
// Per clip (wave w = rows 32 w .. 32 w + 31 = query tile w; lane (n, half) of a 32x32x16 C tile holds rows 8 g + 4 half + q of column n):
//   x      rows -> registers, times the CLIP's power of two (largest |x| into [2^14, 2^15)), two binary16 terms: 9 x 2 fragments.  The same
//          registers serve as B operand (column = row of x) of the transposed products and as A operand (row = row of x) of the plain ones.
//   q      Qt [32 dims x 32 rows] = Wq_head . Xt (transposed): a lane holds its OWN query's dims - scaled by the row's power of two, split,
//          they are the B operand of the score product without leaving the registers (as mha_h2.hip).
//   k      Kt likewise: the lane holds its own key's dims, eight consecutive k slots at a time -> 16-byte LDS stores, K[key][slot].
//          The k bias is dropped: q . b_k is the same for every key of a query and cancels in the softmax.
//   v      V [32 rows x 32 dims] = X . Wv_head^T (NOT transposed: A = x fragments, B = weight fragments): the lane holds ONE dim of 16
//          rows, eight consecutive key positions at a time -> 16-byte LDS stores into Vt[dim][key position] (mha_h2's permuted key order).
//          The v bias is deferred: softmax rows sum to 1, so it passes through the average; Wo . b_v is folded into the output bias.
//          k and v need no data-driven scale: |x| <= 2^15 after the clip scale, so the raw accumulators are bounded by the L1 norms of
//          the packed weight rows - plan-time powers of two cK, cV bring them into the binary16 range (typical values land ~3 bits lower
//          than a data-driven maximum would put them: two terms still hold 22 bits of every value that matters).
//   St, softmax, Ot = Vt . Pt exactly as mha_h2.hip (scores' accumulators ARE the B operand of the second product).
//   out    Yt [D x 32 rows] += Wo[:, head dims] . Ot: the normalised Ot tile, split in place, is the B operand (k slots permuted the way the
//          plan-time packed Wo follows - ffn_x3.hip's device); five accumulators live across the heads; h + Yt + bias is stored at the end.
// Head dim 36 = one 32-dim tile + 4 left-over dims: the left-over q / k rows of all heads share ONE transposed tile and the left-over v
// columns ONE plain tile, computed at the top of the clip (14 projection tiles per wave instead of 24).
// Weights: 2 + 4 x heads chunks (<= 32 KB: 18 fragments of 1 KB + biases; out_proj 30) streamed through four LDS slots by LDS-DMA two
// chunks ahead of their use, one barrier per chunk; the chunk sequence is periodic in the clip, so the stream never drains.
// One workgroup per CU (158 KB of LDS, ~390 registers per lane).
// Measured and not kept (tools/ubench/attn_trace, profiles/r06_attn_trace*.txt): (a) a head as TWO phases (k, v, q tiles | scores .. out_proj: 11
// barriers per clip instead of 19, every fetch a full phase ahead) - 85 k clocks per clip against 81.5 k: the phases are chains of LDS
// latency -> dependent MFMAs -> epilogue VALU in a lone in-order wave, not barrier waits; (b) every residual block requested up front
// in the epilogue - the same: all 256 CUs reach their epilogue together and move 3 x 58 KB each at HBM speed whatever the order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "layers.h"
#include "attn_x3.h"
#include "split_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef NWW_TRACE      // tools/ubench/attn_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of its second clip
__device__ unsigned long long g_attn_trace[4 * 32];
#define ATT_STAMP(k) if (blockIdx.x == 0 && att_it == 1 && lane == 0) g_attn_trace[wave * 32 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define ATT_STAMP(k)
#endif

namespace {

constexpr int ATT_CH_IN = 20480;     // an in_proj chunk: D/16 x 2 fragments of 1 KB, 32 biases; padded to whole 4 KB copy steps
constexpr int ATT_CH_O = 32768;      // a head's out_proj chunk: ceil(D/32) x 3 x 2 fragments
__host__ __device__ constexpr size_t att_head_base(int j) { return (size_t)2 * ATT_CH_IN + (size_t)j * (3 * ATT_CH_IN + ATT_CH_O); }

__device__ __forceinline__ float att_pow2_to_2p14(float m) {     // largest power of two s with m s <= 2^14 (mha_h2.hip)
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(m, &e);
    return ldexpf(1.0f, min(14 - e, 100));
}
__device__ __forceinline__ f16x8 att_ld(const unsigned char* p) { return *reinterpret_cast<const f16x8*>(p); }
// three products of two-term operands, small terms first: lo*hi, hi*lo, hi*hi
__device__ __forceinline__ void att_mfma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}
__device__ __forceinline__ void att_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        uint32_t hh, ll;
        nww_split2h(v[2 * e2], v[2 * e2 + 1], hh, ll);
        hi[e2] = hh; lo[e2] = ll;
    }
}

// ---- plan-time packing: one thread per (chunk, fragment, lane); chunk order = the kernel's: q/k left-overs, v left-overs, then per
// head q, k, v, out_proj.  A fragment is 64 lanes x 16 bytes (hi) + the same (lo); lane (i, hh) holds k slots 8 hh + e of row / column i.
__global__ void __launch_bounds__(256) attn_pack_kernel(const float* __restrict__ in_w, const float* __restrict__ in_b,
                                                        const float* __restrict__ out_w, unsigned char* __restrict__ out,
                                                        int D, int NH, float ws_in, float ws_out) {
    const int D16 = D / 16, DH = D / NH, L = DH - 32, NOB = (D + 31) / 32;
    const int nch = 2 + 4 * NH;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nch * 32 * 64) return;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) & 31), c = (int)(idx >> 11);
    const int i = lane & 31, hh = lane >> 5;
    const int j = c >= 2 ? (c - 2) >> 2 : 0, kind = c >= 2 ? (c - 2) & 3 : 4 + c;      // 0 q, 1 k, 2 v, 3 out, 4 q/k left, 5 v left
    unsigned char* base = out + (c == 0 ? 0 : c == 1 ? (size_t)ATT_CH_IN : att_head_base(j) + (size_t)kind * ATT_CH_IN);
    float v[8];
    float sc = ws_in;
    if (kind == 3) {
        if (f >= NOB * 3) return;
        const int ob = f / 3, kbo = f - 3 * ob, m = 32 * ob + i;
        sc = ws_out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int d = -1;
            if (kbo < 2) d = 8 * (2 * kbo + (e >> 2)) + 4 * hh + (e & 3);
            else if (hh == 0 && e < L) d = 32 + e;
            v[e] = (m < D && d >= 0) ? out_w[(size_t)m * D + j * DH + d] : 0.0f;
        }
    } else {
        if (f >= D16) return;
        int src = -1;                                          // in_proj row of this tile row / column i
        float bias = 0.0f;
        if (kind == 0) { src = j * DH + i; bias = in_b[src]; }
        else if (kind == 1) src = D + j * DH + i;
        else if (kind == 2) src = 2 * D + j * DH + i;
        else if (kind == 4) {                                  // row 8 g + 4 half + q: half 0 = q dim 32 + q of head g, half 1 = k dim 32 + q
            const int g = i >> 3, half = (i >> 2) & 1, q = i & 3;
            if (g < NH && q < L) { src = half * D + g * DH + 32 + q; if (!half) bias = in_b[src]; }
        } else {                                               // column 4 head + jj = v dim 32 + jj of that head
            const int head = i >> 2, jj = i & 3;
            if (head < NH && jj < L) src = 2 * D + head * DH + 32 + jj;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src >= 0 ? in_w[(size_t)src * D + 16 * f + 8 * hh + e] : 0.0f;
        if (f == 0 && hh == 0) reinterpret_cast<float*>(base + (size_t)D16 * 2048)[i] = bias;
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) nww_split2h(v[2 * e] * sc, v[2 * e + 1] * sc, hi[e], lo[e]);
    unsigned char* dst = base + ((size_t)(f * 2) * 64 + lane) * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
// bc[m] = out_b[m] + sum_k out_w[m][k] v_bias[k]  (one fmaf chain per output, ascending k)
__global__ void __launch_bounds__(256) attn_bias_kernel(const float* __restrict__ in_b, const float* __restrict__ out_w,
                                                        const float* __restrict__ out_b, float* __restrict__ bc, int D) {
    const int m = (int)blockIdx.x * 256 + threadIdx.x;
    if (m >= D) return;
    float s = 0.0f;
    for (int k = 0; k < D; ++k) s = fmaf(out_w[(size_t)m * D + k], in_b[2 * D + k], s);
    bc[m] = out_b[m] + s;
}

// NT16C: number of 16-key blocks, ceil(T / 16), as a compile-time constant (branch-free score / P V loops: the scheduler can then put a
// block's probability split under the previous block's MFMAs); 0 = taken from T at run time
template <int D16, int NH, int DH, int NT16C>
__global__ void __launch_bounds__(256) attn_x3_kernel(AttnArgs a) {
    constexpr int D = 16 * D16, L = DH - 32, NOB = (D + 31) / 32;
    static_assert(DH >= 32 && DH <= 36 && L % 4 == 0 && NH * 8 <= 32 && NH * DH == D, "attn_x3: one 32-dim tile + at most 4 left-over dims per head");
    constexpr int KROW = 80, VROW = 272;                       // bytes per K row (2 k-blocks + 16: odd number of 16-byte slots) / Vt row (128 keys + 16)
    constexpr int K_BYTES = 128 * KROW, V_BYTES = 32 * VROW, VL_ROWS = 4 * NH + 1, VL_BYTES = VL_ROWS * VROW, KL_BYTES = NH * 128 * 8;
    constexpr int W_FR = D16 * 2048;                           // fragments of an in_proj chunk; its 32 biases follow
    constexpr int S_IN = ATT_CH_IN / 4096, S_O = ATT_CH_O / 4096;
    static_assert(W_FR + 128 <= ATT_CH_IN && NOB * 3 * 2048 <= ATT_CH_O, "chunk sizes");
    // separate LDS objects: the compiler then knows which in-flight LDS-DMA a read may alias (ffn_x3.hip)
    __shared__ __attribute__((aligned(16))) unsigned char slot0[ATT_CH_IN];      // q chunks
    __shared__ __attribute__((aligned(16))) unsigned char slot1[ATT_CH_IN];      // k chunks, q/k left-overs
    __shared__ __attribute__((aligned(16))) unsigned char slot2[ATT_CH_IN];      // v chunks, v left-overs
    __shared__ __attribute__((aligned(16))) unsigned char slot3[ATT_CH_O];       // out_proj chunks
    __shared__ __attribute__((aligned(16))) unsigned char Kh[K_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Kl[K_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vh[V_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vl[V_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char VLh[L > 0 ? VL_BYTES : 16];     // v left-overs [4 head + jj][key position]; last row zeros
    __shared__ __attribute__((aligned(16))) unsigned char VLl[L > 0 ? VL_BYTES : 16];
    __shared__ __attribute__((aligned(16))) unsigned char KLh[L > 0 ? KL_BYTES : 16];     // k left-overs [head][key][4]
    __shared__ __attribute__((aligned(16))) unsigned char KLl[L > 0 ? KL_BYTES : 16];
    __shared__ __attribute__((aligned(16))) float QL[L > 0 ? NH * 128 * 4 : 4];           // q left-overs [head][query][4], true scale
    __shared__ float red[4];
    const int T = a.T;
    // thread coordinates are re-derived from an OPAQUE copy of threadIdx at the top of every clip (convmod_x3.hip: derived from the plain
    // value, the lane addresses of the whole clip are loop-invariant and get hoisted out of the clip loop into registers the phases need)
    int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n = lane & 31, h = lane >> 5;
    int tok = 32 * wave + n, tokc = min(tok, T - 1);
    auto rederive = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        tid = t; lane = t & 63; wave = __builtin_amdgcn_readfirstlane(t >> 6);
        n = lane & 31; h = lane >> 5;
        tok = 32 * wave + n; tokc = min(tok, T - 1);
    };
    const int NT16 = NT16C > 0 ? NT16C : (T + 15) / 16, NTk = (NT16 + 1) / 2;      // 16-key blocks, 32-key tiles with keys below T

    auto fetch = [&](const unsigned char* src, unsigned char* slot, auto steps) {
        const unsigned char* sp = src + tid * 16;
        unsigned char* dst = slot + wave * 1024;               // wave-uniform
#pragma unroll
        for (int s = 0; s < decltype(steps)::value; ++s)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sp + s * 4096),
                                             (void __attribute__((address_space(3)))*)(dst + s * 4096), 16, 0, 0);
    };
    using StepsIn = std::integral_constant<int, S_IN>;
    using StepsO = std::integral_constant<int, S_O>;
    // Start-up stagger (many clips per workgroup only): every workgroup runs the same phases on the same amount of data, so all CUs reach
    // their memory phase - the epilogue's residual rows in, updated rows out, the next clip's rows in: 3 x 58 KB - TOGETHER and share the HBM
    // for it; started in eight phases 2000 clocks apart they stay apart for the whole launch.  Measured at B = 2048 (tools/ubench/attn_trace):
    // epilogue 16.6-17.4 k -> 11.7-12.1 k clocks, clip 81.7 k -> 77.3 k, launch 0.340 -> 0.326 ms; 1000 / 4000 / 8000 clocks per phase:
    // 0.335 / 0.336 / 0.351 (the delay itself: seven steps at most, once).
    if (a.stagger > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long d = (unsigned long long)(((int)blockIdx.x >> 3) & 7) * (unsigned)a.stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    if ((int)blockIdx.x < a.B) {
        fetch(a.packed, slot1, StepsIn{});
        fetch(a.packed + ATT_CH_IN, slot2, StepsIn{});
    }
    if (L > 0)
        for (int i = tid; i < VROW / 4; i += 256) {
            reinterpret_cast<uint32_t*>(VLh + 4 * NH * VROW)[i] = 0u;
            reinterpret_cast<uint32_t*>(VLl + 4 * NH * VROW)[i] = 0u;
        }

    // one projection tile over K = D: W fragments from the slot; TR: Zt = W . Xt (lane = row of x, registers = outputs), else Z = X . Wt
    auto proj_tile = [&](const unsigned char* slot, const f16x8 (&xf)[D16][2], auto tr) {
        constexpr bool TR = decltype(tr)::value;
        const unsigned char* wp = slot + lane * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        f16x8 nh = att_ld(wp), nl = att_ld(wp + 1024);
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            const f16x8 ch = nh, cl = nl;
            if (kb + 1 < D16) { nh = att_ld(wp + (kb + 1) * 2048); nl = att_ld(wp + (kb + 1) * 2048 + 1024); }
            if (TR) att_mfma3(ch, cl, xf[kb][0], xf[kb][1], acc);
            else att_mfma3(xf[kb][0], xf[kb][1], ch, cl, acc);
        }
        return acc;
    };

    // ---- a clip's rows: lane (n, half) takes features 16 kb + 8 half + e of row 32 wave + n.  Requested one clip ahead (at the top of the
    // previous clip's epilogue, when the fragment registers are dead), so the trip to HBM runs under that epilogue
    float v[D16][8];
    auto load_rows = [&](int clip) {
        const float* xrow = a.h + ((size_t)clip * T + tokc) * D;
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h);
            const float4 p1 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h + 4);
            v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
            v[kb][4] = p1.x; v[kb][5] = p1.y; v[kb][6] = p1.z; v[kb][7] = p1.w;
        }
    };
    if ((int)blockIdx.x < a.B) load_rows((int)blockIdx.x);
    [[maybe_unused]] int att_it = 0;
    for (int clip = (int)blockIdx.x; clip < a.B; clip += (int)gridDim.x) {
        ATT_STAMP(0)
        rederive();
        f16x8 xf[D16][2];
        float isx;                                             // 1 / the clip's scale
        {
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < D16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[kb][e]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if (lane == 0) red[wave] = m;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the left-over chunks have landed (and the previous clip's stores left)
            __syncthreads();
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
            const float sX = __uint_as_float((268u - eb) << 23);       // m sX in [2^14, 2^15)
            isx = __uint_as_float((eb - 14u) << 23);
#pragma unroll
            for (int kb = 0; kb < D16; ++kb) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = v[kb][e] * sX;
                u32x4 hi, lo;
                att_split8(y, hi, lo);
                xf[kb][0] = __builtin_bit_cast(f16x8, hi); xf[kb][1] = __builtin_bit_cast(f16x8, lo);
            }
        }
        const float un = isx * a.w_un;                         // raw in_proj accumulator -> true value
        ATT_STAMP(1)

        // ---- left-over q / k rows of all heads (slot 1); q chunk of head 0 -> slot 0
        fetch(a.packed + att_head_base(0), slot0, StepsIn{});
        if (L > 0) {
            const f32x16 acc = proj_tile(slot1, xf, std::true_type{});
            const float* bp = reinterpret_cast<const float*>(slot1 + W_FR) + 4 * h;
            const float m1 = h == 0 ? un : a.cK, m2 = h == 0 ? a.qscale : 1.0f;
#pragma unroll
            for (int g = 0; g < NH; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 8 * g);
                const float v0 = fmaf(acc[4 * g], m1, b.x) * m2, v1 = fmaf(acc[4 * g + 1], m1, b.y) * m2;
                const float v2 = fmaf(acc[4 * g + 2], m1, b.z) * m2, v3 = fmaf(acc[4 * g + 3], m1, b.w) * m2;
                if (h == 0) {
                    *reinterpret_cast<float4*>(QL + (g * 128 + tok) * 4) = make_float4(v0, v1, v2, v3);
                } else {
                    uint32_t h0, l0, h1, l1;
                    nww_split2h(v0, v1, h0, l0);
                    nww_split2h(v2, v3, h1, l1);
                    *reinterpret_cast<uint2*>(KLh + (g * 128 + tok) * 8) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(KLl + (g * 128 + tok) * 8) = make_uint2(l0, l1);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_IN) : "memory");
        __syncthreads();
        ATT_STAMP(2)
        // ---- left-over v columns of all heads (slot 2); k chunk of head 0 -> slot 1
        fetch(a.packed + att_head_base(0) + ATT_CH_IN, slot1, StepsIn{});
        if (L > 0) {
            const f32x16 acc = proj_tile(slot2, xf, std::false_type{});
            if (n < 4 * NH) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = acc[8 * jb + e] * a.cV;
                    u32x4 hi, lo;
                    att_split8(y, hi, lo);
                    const int off = n * VROW + (16 * (2 * wave + jb) + 8 * h) * 2;
                    *reinterpret_cast<u32x4*>(VLh + off) = hi;
                    *reinterpret_cast<u32x4*>(VLl + off) = lo;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_IN) : "memory");
        __syncthreads();
        ATT_STAMP(3)

        f32x16 yacc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.0f;

#pragma unroll 1
        for (int j = 0; j < NH; ++j) {
            const unsigned char* hb = a.packed + att_head_base(j);
            // ---- q (slot 0): the lane's query, scaled by its own power of two, split -> B fragments; v chunk -> slot 2
            fetch(hb + 2 * ATT_CH_IN, slot2, StepsIn{});
            u32x4 qh[3], ql[3];
            float sQ;
            {
                const f32x16 acc = proj_tile(slot0, xf, std::true_type{});
                const float* bp = reinterpret_cast<const float*>(slot0 + W_FR) + 4 * h;
                float qv[16];
                float mq = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(bp + 8 * g);
                    qv[4 * g] = fmaf(acc[4 * g], un, b.x) * a.qscale; qv[4 * g + 1] = fmaf(acc[4 * g + 1], un, b.y) * a.qscale;
                    qv[4 * g + 2] = fmaf(acc[4 * g + 2], un, b.z) * a.qscale; qv[4 * g + 3] = fmaf(acc[4 * g + 3], un, b.w) * a.qscale;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mq = fmaxf(mq, fabsf(qv[r]));
                float4 qe = make_float4(0.f, 0.f, 0.f, 0.f);
                if (L > 0 && h == 0) qe = *reinterpret_cast<const float4*>(QL + (j * 128 + tok) * 4);
                mq = fmaxf(mq, fmaxf(fmaxf(fabsf(qe.x), fabsf(qe.y)), fmaxf(fabsf(qe.z), fabsf(qe.w))));
                mq = fmaxf(mq, __shfl_xor(mq, 32, 64));
                sQ = att_pow2_to_2p14(mq);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = qv[8 * kb + e] * sQ;
                    att_split8(y, qh[kb], ql[kb]);
                }
                uint32_t h0, l0, h1, l1;
                nww_split2h(qe.x * sQ, qe.y * sQ, h0, l0);
                nww_split2h(qe.z * sQ, qe.w * sQ, h1, l1);
                qh[2] = u32x4{h0, h1, 0u, 0u}; ql[2] = u32x4{l0, l1, 0u, 0u};
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_IN) : "memory");
            __syncthreads();
            ATT_STAMP(4 + 4 * j)
            // ---- k (slot 1): the lane's key, eight consecutive k slots per 16-byte store; out_proj chunk -> slot 3
            fetch(hb + 3 * ATT_CH_IN, slot3, StepsO{});
            {
                const f32x16 acc = proj_tile(slot1, xf, std::true_type{});
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = acc[8 * kb + e] * a.cK;
                    u32x4 hi, lo;
                    att_split8(y, hi, lo);
                    const int off = tok * KROW + 32 * kb + 16 * h;
                    *reinterpret_cast<u32x4*>(Kh + off) = hi;
                    *reinterpret_cast<u32x4*>(Kl + off) = lo;
                }
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_O) : "memory");
            __syncthreads();
            ATT_STAMP(5 + 4 * j)
            // ---- v (slot 2), not transposed: the lane holds dim n of 16 rows; the next head's q chunk -> slot 0 (last head: the next clip's left-overs -> slot 1)
            if (j + 1 < NH) fetch(a.packed + att_head_base(j + 1), slot0, StepsIn{});
            else fetch(a.packed, slot1, StepsIn{});
            {
                const f32x16 acc = proj_tile(slot2, xf, std::false_type{});
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = acc[8 * jb + e] * a.cV;
                    u32x4 hi, lo;
                    att_split8(y, hi, lo);
                    const int off = n * VROW + (16 * (2 * wave + jb) + 8 * h) * 2;
                    *reinterpret_cast<u32x4*>(Vh + off) = hi;
                    *reinterpret_cast<u32x4*>(Vl + off) = lo;
                }
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_IN) : "memory");
            __syncthreads();
            ATT_STAMP(6 + 4 * j)
            // ---- scores, softmax, Ot, out_proj partial (slot 3); the next head's k chunk -> slot 1 (last head: the next clip's v left-overs -> slot 2)
            if (j + 1 < NH) fetch(a.packed + att_head_base(j + 1) + ATT_CH_IN, slot1, StepsIn{});
            else fetch(a.packed + ATT_CH_IN, slot2, StepsIn{});
            {
                f32x16 st[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] = 0.0f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        if (kt < NTk) {
                            const int off = (32 * kt + n) * KROW + 32 * kb + 16 * h;
                            att_mfma3(att_ld(Kh + off), att_ld(Kl + off), __builtin_bit_cast(f16x8, qh[kb]), __builtin_bit_cast(f16x8, ql[kb]), st[kt]);
                        }
                if (L > 0) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        if (kt < NTk) {
                            // (both halves read the key's four left-over dims - a broadcast; the upper half's k slots are zeros)
                            const uint2 kh2 = *reinterpret_cast<const uint2*>(KLh + (j * 128 + 32 * kt + n) * 8);
                            const uint2 kl2 = *reinterpret_cast<const uint2*>(KLl + (j * 128 + 32 * kt + n) * 8);
                            const u32x4 ah = {h == 0 ? kh2.x : 0u, h == 0 ? kh2.y : 0u, 0u, 0u}, al = {h == 0 ? kl2.x : 0u, h == 0 ? kl2.y : 0u, 0u, 0u};
                            att_mfma3(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, qh[2]),
                                      __builtin_bit_cast(f16x8, ql[2]), st[kt]);
                        }
                }
                // softmax over the keys in the exp2 domain (mha_h2.hip): raw accumulators are sK sQ times the scores, sK = cK / un
                const float unS = 1.4426950408889634f * un / (a.cK * sQ);
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if (32 * kt + 32 > T) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3);
                            st[kt][r] = key < T ? st[kt][r] : -INFINITY;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float off = 14.0f - mx * unS;
                float den = 0.0f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        st[kt][r] = __builtin_amdgcn_exp2f(fmaf(st[kt][r], unS, off));
                        den += st[kt][r];
                    }
                den += __shfl_xor(den, 32, 64);
                // Ot tiles: dims 0..31 from Vt, the left-over dims from the left-over rows (rows >= L of that tile read the zero row)
                f32x16 ot0, ot1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ot0[r] = 0.0f; ot1[r] = 0.0f; }
                const int lrow = (n < L ? 4 * j + n : 4 * NH) * VROW;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
                        if (2 * kt + jb < NT16) {
                            float y[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = st[kt][8 * jb + e];
                            u32x4 ph, pl;
                            att_split8(y, ph, pl);
                            const int col = (32 * kt + 16 * jb + 8 * h) * 2;
                            att_mfma3(att_ld(Vh + n * VROW + col), att_ld(Vl + n * VROW + col), __builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, pl), ot0);
                            if (L > 0)
                                att_mfma3(att_ld(VLh + lrow + col), att_ld(VLl + lrow + col), __builtin_bit_cast(f16x8, ph), __builtin_bit_cast(f16x8, pl), ot1);
                        }
                // normalised (den and the products both carry the probabilities' 2^14), split: B fragments of the out_proj partial product
                const float inv = 1.0f / den;
                u32x4 oh[3], ol[3];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = ot0[8 * kb + e] * inv;
                    att_split8(y, oh[kb], ol[kb]);
                }
                {
                    uint32_t h0, l0, h1, l1;
                    nww_split2h(ot1[0] * inv, ot1[1] * inv, h0, l0);
                    nww_split2h(ot1[2] * inv, ot1[3] * inv, h1, l1);
                    oh[2] = u32x4{h0, h1, 0u, 0u}; ol[2] = u32x4{l0, l1, 0u, 0u};
                }
                const unsigned char* wo = slot3 + lane * 16;
#pragma unroll
                for (int kbo = 0; kbo < (L > 0 ? 3 : 2); ++kbo)
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        const unsigned char* fp = wo + ((ob * 3 + kbo) * 2) * 1024;
                        att_mfma3(att_ld(fp), att_ld(fp + 1024), __builtin_bit_cast(f16x8, oh[kbo]), __builtin_bit_cast(f16x8, ol[kbo]), yacc[ob]);
                    }
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_IN) : "memory");
            __syncthreads();
            ATT_STAMP(7 + 4 * j)
        }

        // ---- out = h + Yt / (scales) + (b_o + Wo b_v).  As the accumulators hold them a store touches 16-byte pieces of 64 rows; through a
        // per-wave transpose tile (slot 0 is idle from the last head's q tile to the next clip's first fetch; lin_x3.hip's epilogue) lane l
        // of store jj owns 16 bytes of row 8 jj + l / 8 and eight lanes cover 128 contiguous bytes; the residual is loaded the same way -
        // all of it first (out may be h: a load cannot move above a store that may alias it), then the next clip's rows.
        {
            constexpr int TP = 36;
            float* tb = reinterpret_cast<float*>(slot0) + wave * 32 * TP;
            const int tq = lane & 7;
            const float yun = a.o_un * isx;
            const float* rrow[4];
            float* orow[4];
            bool rok[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rt = 32 * wave + 8 * jj + (lane >> 3);
                rok[jj] = rt < T;
                rrow[jj] = a.h + ((size_t)clip * T + min(rt, T - 1)) * D + 4 * tq;
                orow[jj] = a.out + ((size_t)clip * T + min(rt, T - 1)) * D + 4 * tq;
            }
            // Residual one block ahead, the next clip's rows behind the first block's residual request.  (All CUs reach this point together: the
            // phase moves 3 x 58 KB per CU at HBM speed - 14 k of a clip's 86 k clocks - whatever the order of its requests; requesting every
            // residual block up front measured the same, tools/ubench/attn_trace.)
            float4 rcur[4], rnxt[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) rcur[jj] = *reinterpret_cast<const float4*>(rrow[jj]);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                if (ob + 1 < NOB && 32 * (ob + 1) + 4 * tq < D) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) rnxt[jj] = *reinterpret_cast<const float4*>(rrow[jj] + 32 * (ob + 1));
                }
                if (ob == 0) load_rows(min(clip + (int)gridDim.x, a.B - 1));     // (unconditional: a conditionally assigned array keeps its old value live)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(tb + n * TP + 8 * g + 4 * h) = make_float4(yacc[ob][4 * g], yacc[ob][4 * g + 1], yacc[ob][4 * g + 2], yacc[ob][4 * g + 3]);
                __builtin_amdgcn_wave_barrier();               // (one wave: its LDS operations execute in order)
                const int col = 32 * ob + 4 * tq;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col < D) b = *reinterpret_cast<const float4*>(a.bc + col);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 y = *reinterpret_cast<const float4*>(tb + (8 * jj + (lane >> 3)) * TP + 4 * tq);
                    if (rok[jj] && col < D) {
                        const float4 r = rcur[jj];
                        float4 o;
                        o.x = r.x + fmaf(y.x, yun, b.x); o.y = r.y + fmaf(y.y, yun, b.y);
                        o.z = r.z + fmaf(y.z, yun, b.z); o.w = r.w + fmaf(y.w, yun, b.w);
                        *reinterpret_cast<float4*>(orow[jj] + 32 * ob) = o;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) rcur[jj] = rnxt[jj];
            }
        }
        ATT_STAMP(20)
        ++att_it;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup
}

}  // namespace

// (T <= 64: a workgroup would do four query tiles of work for two - the three-launch path is faster there)
bool attn_x3_supported(int T, int D, int n_head) { return D == 144 && n_head == 4 && T > 64 && T <= 128; }

size_t attn_x3_packed_bytes(int D, int n_head) { (void)D; return att_head_base(n_head); }

hipError_t launch_attn_x3_pack(const float* in_w, const float* in_b, const float* out_w, const float* out_b, void* packed, float* bc,
                               int D, int n_head, float ws_in, float ws_out, hipStream_t s) {
    if (!attn_x3_supported(128, D, n_head)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(packed, 0, attn_x3_packed_bytes(D, n_head), s);
    if (e != hipSuccess) return e;
    const size_t total = (size_t)(2 + 4 * n_head) * 32 * 64;
    hipLaunchKernelGGL(attn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in_w, in_b, out_w,
                       reinterpret_cast<unsigned char*>(packed), D, n_head, ws_in, ws_out);
    hipLaunchKernelGGL(attn_bias_kernel, dim3((D + 255) / 256), dim3(256), 0, s, in_b, out_w, out_b, bc, D);
    return hipGetLastError();
}

hipError_t launch_attn_x3(const AttnArgs& a0, int D, int n_head, hipStream_t s) {
    if (a0.B <= 0) return hipSuccess;
    if (!attn_x3_supported(a0.T, D, n_head)) return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(a0.h) | reinterpret_cast<uintptr_t>(a0.out) | reinterpret_cast<uintptr_t>(a0.bc)) & 15) != 0) return hipErrorInvalidValue;
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const dim3 grid(a0.B < cus ? a0.B : cus);
    AttnArgs a = a0;
    a.stagger = a0.B >= 4 * cus ? 2000 : 0;                    // (fewer clips per workgroup: the delay would not pay back)
    const int nt16 = (a.T + 15) / 16;
    if (nt16 == 7) hipLaunchKernelGGL((attn_x3_kernel<9, 4, 36, 7>), grid, dim3(256), 0, s, a);          // T = 97 .. 112: 1 s clips at the 10 ms hop
    else if (nt16 == 8) hipLaunchKernelGGL((attn_x3_kernel<9, 4, 36, 8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_x3_kernel<9, 4, 36, 0>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
