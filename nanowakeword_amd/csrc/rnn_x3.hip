// rnn_x3.hip - GRU / LSTM recurrences with the recurrent product h W_hh^T on the bf16 matrix cores (gfx950).
//
// nn.GRU / nn.LSTM cell semantics as layers.hip states them (CRNNModel / GRU head: nanowakeword/modules/architectures.py:
// 238-254, 731-760): xg = x W_ih^T + b_ih is precomputed for every frame, the kernel walks the T steps of one direction.
//
// One workgroup = 16 clips x all H hidden units for all T steps; wave w owns 16 (H = 128: 32) hidden units of every gate.
// float32 operands are split exactly into three bf16 terms (gemm_x3.hip): W_hh once per launch, into MFMA B fragments that
// stay in registers for all steps (G gates x H/32 k-blocks x 3 terms x 4 registers = 144 for the GRU at H = 128); h once
// per step by the lane that produced it, into three bf16 planes in LDS that every wave reads back as A fragments (three
// 16-byte reads per k-block, shared by the gates).  The 6 (or 9) partial products of a k-block go to
// v_mfma_f32_16x16x32_bf16 - 16 matrix-pipe clocks for K = 32 against 8 x 32 for the same K on v_mfma_f32_16x16x4_f32,
// which is what the float32 instances in layers.hip spend: 6144 of their ~11 000 clocks per step at H = 128.
// C layout = column: hidden unit, rows 4g .. 4g + 3: clips, so xg loads and h stores are coalesced along the hidden dimension.
// xg rows are requested two steps ahead; gate functions on the hardware exp2 / reciprocal (layers.hip: rnn_sigmoid).
// Clips are independent rows of every product: results do not depend on batch size or position.
#include <hip/hip_runtime.h>
#include <stdint.h>
// NP = 3 ("f16x3"): W_hh times a plan-time power of two and h times 2^14 (|h| <= 1: the recurrent operand needs no bound from the
// plan) as TWO binary16 terms each (split_h2.h), three partial products per k-block on v_mfma_f32_16x16x32_f16 - 72 instead of
// 144 MFMAs per step at H = 128, two term planes of h instead of three, and the scaling back is the fma that adds b_hh.
#include "layers.h"
#include "split_h2.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef NWW_TRACE      // tools/ubench/rnn_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of the first 128 steps
__device__ unsigned long long g_rnn_trace[8 * 128 * 8];
#define RNN_STAMP(k) if (blockIdx.x == 0 && step < 128 && lane == 0) g_rnn_trace[((threadIdx.x >> 6) * 128 + step) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define RNN_STAMP(k)
#endif

namespace {
__device__ __forceinline__ void split3r(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack_hi16r(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ float sigmoid_r(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_r(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * v)); }

// G = 3: GRU (r, z, n), G = 4: LSTM (i, f, g, o); NP = 6 or 9 partial products per operand pair; NB = 16-wide column blocks per
// wave (2 at H = 128: four waves, one per SIMD, so that each has the whole 512-register file - 288 of them weight fragments)
// FIN = 32 / 64 (GRU, NP = 3): the input projection of the step fused in (GruArgs::fin); 0: xg precomputed
// PAD (NP = 3): the layer's real width a.H < H, a multiple of 4 (layer_dim = 48, 96, 100 ...): the instance of the next width with the rows
// and columns beyond a.H read as zeros - a padded unit has zero weights, biases and input pre-activations, so its gates are 1/2, 1/2,
// tanh(0) and its state stays 0 for ever (GRU: h' = h / 2; LSTM: c' = c / 2, h' = tanh(c') / 2), which adds nothing to the real units'
// products; all global addressing uses the real width and nothing is stored for the padded units
template <int G, int H, int NP, int NB, int FIN = 0, bool PAD = false>
__global__ void __launch_bounds__(64 * (H / (16 * NB))) rnn_x3_kernel(GruArgs a) {
    static_assert(FIN == 0 || (G == 3 && NP == 3 && FIN % 32 == 0), "fused input projection: GRU, two-term form");
    static_assert(!PAD || (FIN == 0 && NP == 3), "padded widths: two-term form, xg precomputed");
    const int HR = PAD ? a.H : H;                             // real width (global addressing)
    constexpr int KSI = FIN / 32;                             // k-blocks of the input product
    constexpr int KS = H / 32;                                // MFMA k-blocks
    // 16-bit values per LDS row: + 32 bytes.  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) over 64
    // banks (MI355X_MICROARCH.md, LDS); lane (clip n, k-group g) reads 16 bytes at row n, piece g: with a row of 18 pieces every group's sixteen
    // pieces differ (2 n + g mod 16); the + 16 bytes of rounds 3-5 (17 pieces: n + g) were made for groups of consecutive lanes and cost every
    // fragment read a second LDS cycle (tools/lds_conflicts.py's model: 32 cycles per step and wave instead of 16)
    constexpr int LDP = H + 16;
    constexpr bool H2 = NP == 3;
    constexpr int NTM = H2 ? 2 : 3;                           // terms per value
    const float s_h = 16384.0f, s_w = H2 ? a.w_scale : 1.0f, un = H2 ? 1.0f / (16384.0f * a.w_scale) : 1.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    uint16_t* hp = reinterpret_cast<uint16_t*>(smem_r);       // [2 sets][terms][16 clips][LDP]: step t reads set t & 1 and writes the other -
                                                              // ONE barrier per step (a wave writes set s again only two barriers after the last read of it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int j0 = 16 * NB * wave + n;                        // this lane's hidden units j0 + 16 bl (B-operand column, C-layout column)
    for (int idx = threadIdx.x; idx < 2 * NTM * 16 * LDP / 2; idx += blockDim.x) reinterpret_cast<uint32_t*>(hp)[idx] = 0u;

    // W_hh rows q*H + j, k = 32 ks + 8 g .. + 7 -> B fragments, split once
    uint4 wf[G][NB][KS][NTM];
    float bh[G][NB];
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int bl = 0; bl < NB; ++bl) {
            const bool jok = !PAD || j0 + 16 * bl < HR;
            const float* src = a.w_hh + (size_t)(q * HR + (jok ? j0 + 16 * bl : 0)) * HR + 8 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float4 v0, v1;
                if (PAD) {
                    // (in-bounds addresses for every lane, then a select: a conditional 16-byte load puts its result on the stack)
                    const bool ok0 = jok && 32 * ks + 8 * g < HR, ok1 = jok && 32 * ks + 8 * g + 4 < HR;
                    v0 = *reinterpret_cast<const float4*>(ok0 ? src + 32 * ks : a.w_hh);
                    v1 = *reinterpret_cast<const float4*>(ok1 ? src + 32 * ks + 4 : a.w_hh);
                    v0.x = ok0 ? v0.x : 0.0f; v0.y = ok0 ? v0.y : 0.0f; v0.z = ok0 ? v0.z : 0.0f; v0.w = ok0 ? v0.w : 0.0f;
                    v1.x = ok1 ? v1.x : 0.0f; v1.y = ok1 ? v1.y : 0.0f; v1.z = ok1 ? v1.z : 0.0f; v1.w = ok1 ? v1.w : 0.0f;
                } else {
                    v0 = *reinterpret_cast<const float4*>(src + 32 * ks); v1 = *reinterpret_cast<const float4*>(src + 32 * ks + 4);
                }
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (H2) {
                    uint32_t hh[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) nww_split2h(x[2 * e] * s_w, x[2 * e + 1] * s_w, hh[e], ll[e]);
                    wf[q][bl][ks][0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                    wf[q][bl][ks][NTM - 1] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                } else {
                    uint32_t hi[8], mid[8], lo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) split3r(x[e], hi[e], mid[e], lo[e]);
                    wf[q][bl][ks][0] = make_uint4(pack_hi16r(hi[0], hi[1]), pack_hi16r(hi[2], hi[3]), pack_hi16r(hi[4], hi[5]), pack_hi16r(hi[6], hi[7]));
                    wf[q][bl][ks][1] = make_uint4(pack_hi16r(mid[0], mid[1]), pack_hi16r(mid[2], mid[3]), pack_hi16r(mid[4], mid[5]), pack_hi16r(mid[6], mid[7]));
                    wf[q][bl][ks][NTM - 1] = make_uint4(pack_hi16r(lo[0], lo[1]), pack_hi16r(lo[2], lo[3]), pack_hi16r(lo[4], lo[5]), pack_hi16r(lo[6], lo[7]));
                }
            }
            bh[q][bl] = jok ? a.b_hh[q * HR + j0 + 16 * bl] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);                // one row at a time: all rows' raw loads in flight at once would not fit beside the fragments
        }
    // fused input projection: W_ih rows q*H + j, k = 32 ks + 8 g .. + 7 -> B fragments (two binary16 terms of weight x wi_scale)
    uint4 wi[G][NB][KSI > 0 ? KSI : 1][2];
    float bi[G][NB];
    const float unx = FIN ? 1.0f / (a.x_scale * a.wi_scale) : 0.0f;
    if constexpr (FIN > 0) {
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) {
                const float* src = a.w_ih + (size_t)(q * H + j0 + 16 * bl) * FIN + 8 * g;
#pragma unroll
                for (int ks = 0; ks < KSI; ++ks) {
                    const float4 v0 = *reinterpret_cast<const float4*>(src + 32 * ks), v1 = *reinterpret_cast<const float4*>(src + 32 * ks + 4);
                    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    uint32_t hh[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) nww_split2h(x[2 * e] * a.wi_scale, x[2 * e + 1] * a.wi_scale, hh[e], ll[e]);
                    wi[q][bl][ks][0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                    wi[q][bl][ks][1] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                }
                bi[q][bl] = a.b_ih[q * H + j0 + 16 * bl];
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float hprev[NB][4], cprev[NB][4];
#pragma unroll
    for (int bl = 0; bl < NB; ++bl)
#pragma unroll
        for (int r = 0; r < 4; ++r) { hprev[bl][r] = 0.0f; cprev[bl][r] = 0.0f; }
    const unsigned char* arow0 = smem_r + (size_t)(n * LDP + 8 * g) * 2;         // A fragment: clip n, k = 32 ks + 8 g .. + 7
    constexpr int PLANE = 16 * LDP * 2;                                           // bytes per term plane
    constexpr int SET = NTM * PLANE;                                              // bytes per set of planes
    // input-side pre-activations (independent of h): the NEXT step's are requested right behind the last use of this step's, into the same
    // registers.  hipcc waits for loads carried round the loop with s_waitcnt vmcnt(0): a request issued at the step's top, AHEAD of this
    // step's first use (rounds 2-5: "two steps ahead"), made that wait cover the new loads as well - a trip to HBM on every step's critical
    // path, 2000-3300 of the GRU head's 6400 clocks per step (tools/ubench/rnn_trace).  Issued behind the use, the wait finds them a step old.
    // (the four-wave LSTM at H = 128 - 384 registers of weight fragments - has no room to carry a step's rows round the loop: it requests them at
    // the step's top and uses them behind the products)
    constexpr bool CARRY = !(G == 4 && H == 128 && NB == 2);
    float xpf[1][FIN ? 1 : G][FIN ? 1 : NB][4];                // this step's (precomputed xg form)
    // (the lane's four row pointers walk the frames: forming them from (clip, frame) cost ~70 VALU instructions per step - tools/ubench/rnn_trace)
    const float* xrow[4];
    const ptrdiff_t xstep = (ptrdiff_t)(a.reverse ? -1 : 1) * G * HR;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        xrow[r] = FIN ? nullptr : a.xg + ((size_t)min(b0 + 4 * g + r, a.B - 1) * a.T + (a.reverse ? a.T - 1 : 0)) * G * HR + j0;
    auto fetch = [&](int step, float (&x)[FIN ? 1 : G][FIN ? 1 : NB][4]) {      // steps in order: 0, 1, 2, ...
        (void)step;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* xg = xrow[r];
#pragma unroll
            for (int q = 0; q < (FIN ? 1 : G); ++q)
#pragma unroll
                for (int bl = 0; bl < (FIN ? 1 : NB); ++bl) x[q][bl][r] = (!PAD || j0 + 16 * bl < HR) ? xg[q * HR + 16 * bl] : 0.0f;
            xrow[r] += xstep;
        }
    };
    // fused form: the lane's 8 features per k-block of clip n's row of the step (the A fragment of the input product), raw
    float4 xraw[1][KSI > 0 ? KSI : 1][2];
    const float* xin_row = FIN ? a.x_in + ((size_t)min(b0 + n, a.B - 1) * a.T + (a.reverse ? a.T - 1 : 0)) * FIN + 8 * g : nullptr;
    auto fetch_x = [&](int step, float4 (&x)[KSI > 0 ? KSI : 1][2]) {           // steps in order
        (void)step;
        const float* row = xin_row;
#pragma unroll
        for (int ks = 0; ks < KSI; ++ks) {
            x[ks][0] = *reinterpret_cast<const float4*>(row + 32 * ks);
            x[ks][1] = *reinterpret_cast<const float4*>(row + 32 * ks + 4);
        }
        xin_row += (a.reverse ? -1 : 1) * FIN;
    };
    if (a.steps > 0) { if constexpr (FIN > 0) fetch_x(0, xraw[0]); else if constexpr (CARRY) fetch(0, xpf[0]); }
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        RNN_STAMP(0)
        f32x4 acc[G][NB];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) acc[q][bl] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FIN == 0 && !CARRY) fetch(step, xpf[0]);
        // fused input projection of this step: independent of h - on the matrix pipe ahead of the recurrent product
        f32x4 accx[G][NB];
        if constexpr (FIN > 0) {
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int bl = 0; bl < NB; ++bl) accx[q][bl] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float cl = a.x_clamp, sc = a.x_scale;
#pragma unroll
            for (int ks = 0; ks < KSI; ++ks) {
                const float4 p0 = xraw[0][ks][0], p1 = xraw[0][ks][1];
                const float xv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                uint32_t hh[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    nww_split2h(__builtin_amdgcn_fmed3f(xv[2 * e], -cl, cl) * sc, __builtin_amdgcn_fmed3f(xv[2 * e + 1], -cl, cl) * sc, hh[e], ll[e]);
                const f16x8 xh = __builtin_bit_cast(f16x8, make_uint4(hh[0], hh[1], hh[2], hh[3])), xl = __builtin_bit_cast(f16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
                // term-major, like the recurrent product below: consecutive MFMAs go to DIFFERENT accumulators (three back-to-back products into
                // one accumulator are a dependent chain of full MFMA latencies - tools/ubench/rnn_trace: this phase took 2000-3300 of a step's 6400 clocks)
#define RNN_PROD_X(AF, WT)                                                                                            \
    _Pragma("unroll") for (int q = 0; q < G; ++q)                                                                     \
        _Pragma("unroll") for (int bl = 0; bl < NB; ++bl)                                                             \
            accx[q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF, __builtin_bit_cast(f16x8, wi[q][bl][ks][WT]), accx[q][bl], 0, 0, 0);
                RNN_PROD_X(xl, 0) RNN_PROD_X(xh, 1) RNN_PROD_X(xh, 0)
#undef RNN_PROD_X
            }
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < a.steps) fetch_x(step + 1, xraw[0]);      // (behind the split of this step's row)
        }
        const unsigned char* arow = arow0 + (step & 1) * SET;
        uint16_t* hpw = hp + ((step & 1) ^ 1) * (SET / 2);
        RNN_STAMP(1)
        {                                                     // (first step: the planes hold zeros)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arow + 64 * ks), am = *reinterpret_cast<const bf16x8*>(arow + PLANE + 64 * ks),
                             al = *reinterpret_cast<const bf16x8*>(arow + (NTM - 1) * PLANE + 64 * ks);
                // small terms first, the dominant hi*hi last (gemm_x3.hip's order); every product feeds all of the wave's accumulators
#define RNN_PROD(AF, WT)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < G; ++q)                                                                     \
        _Pragma("unroll") for (int bl = 0; bl < NB; ++bl)                                                             \
            acc[q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF, __builtin_bit_cast(bf16x8, wf[q][bl][ks][WT]), acc[q][bl], 0, 0, 0);
#define RNN_PROD_H(AF, WT)                                                                                            \
    _Pragma("unroll") for (int q = 0; q < G; ++q)                                                                     \
        _Pragma("unroll") for (int bl = 0; bl < NB; ++bl)                                                             \
            acc[q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, AF), __builtin_bit_cast(f16x8, wf[q][bl][ks][WT]), acc[q][bl], 0, 0, 0);
                if (H2) { RNN_PROD_H(al, 0) RNN_PROD_H(ah, 1) RNN_PROD_H(ah, 0) }       // lo hi, hi lo, hi hi (al = the second plane here)
                else {
                    if (NP == 9) { RNN_PROD(al, 2) RNN_PROD(al, 1) RNN_PROD(am, 2) }
                    RNN_PROD(am, 1) RNN_PROD(ah, 2) RNN_PROD(al, 0) RNN_PROD(ah, 1) RNN_PROD(am, 0) RNN_PROD(ah, 0)
                }
#undef RNN_PROD_H
#undef RNN_PROD
            }
        }
        RNN_STAMP(2)
#pragma unroll
        for (int bl = 0; bl < NB; ++bl) {
            const int j = j0 + 16 * bl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * g + r, b = b0 + c;
                // rows beyond B repeat the last clip (clamped xg row): straight-line gate arithmetic, only the stores are predicated
                float hn, cn = 0.0f;
                if constexpr (G == 3) {
                    // input-side pre-activations: precomputed rows, or the fused product back at the true scale plus b_ih
                    float xq[3];
                    if constexpr (FIN > 0) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) xq[q] = fmaf(accx[q][bl][r], unx, bi[q][bl]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 3; ++q) xq[q] = xpf[0][q][bl][r];
                    }
                    // (two-term form: the accumulators come back to the true scale inside the fma that adds b_hh)
                    const float a0 = H2 ? fmaf(acc[0][bl][r], un, bh[0][bl]) : acc[0][bl][r] + bh[0][bl];
                    const float a1 = H2 ? fmaf(acc[1][bl][r], un, bh[1][bl]) : acc[1][bl][r] + bh[1][bl];
                    const float a2 = H2 ? fmaf(acc[2][bl][r], un, bh[2][bl]) : acc[2][bl][r] + bh[2][bl];
                    const float rg = sigmoid_r(H2 ? xq[0] + a0 : xq[0] + acc[0][bl][r] + bh[0][bl]);
                    const float zg = sigmoid_r(H2 ? xq[1] + a1 : xq[1] + acc[1][bl][r] + bh[1][bl]);
                    const float ng = tanh_r(xq[2] + rg * a2);
                    hn = (1.0f - zg) * ng + zg * hprev[bl][r];
                } else {
                    const float ig = sigmoid_r(H2 ? xpf[0][0][bl][r] + fmaf(acc[0][bl][r], un, bh[0][bl]) : xpf[0][0][bl][r] + acc[0][bl][r] + bh[0][bl]);
                    const float fg = sigmoid_r(H2 ? xpf[0][1][bl][r] + fmaf(acc[1][bl][r], un, bh[1][bl]) : xpf[0][1][bl][r] + acc[1][bl][r] + bh[1][bl]);
                    const float gg = tanh_r(H2 ? xpf[0][2][bl][r] + fmaf(acc[2][bl][r], un, bh[2][bl]) : xpf[0][2][bl][r] + acc[2][bl][r] + bh[2][bl]);
                    const float og = sigmoid_r(H2 ? xpf[0][G - 1][bl][r] + fmaf(acc[G - 1][bl][r], un, bh[G - 1][bl]) : xpf[0][G - 1][bl][r] + acc[G - 1][bl][r] + bh[G - 1][bl]);
                    cn = fg * cprev[bl][r] + ig * gg;
                    hn = og * tanh_r(cn);
                }
                if (b < a.B && (!PAD || j < HR)) {
                    if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                    if (a.last_out && step == a.steps - 1) {
                        a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
                        if (a.xg2) {          // the opposite direction's first step: h = c = 0, the recurrent product vanishes
                            const float* x2 = a.xg2 + (size_t)b * a.xg2_bstride + j;
                            float h2;
                            if (G == 3) {
                                const float rg = sigmoid_r(x2[0] + a.b_hh2[j]);
                                const float zg = sigmoid_r(x2[HR] + a.b_hh2[HR + j]);
                                const float ng = tanh_r(x2[2 * HR] + rg * a.b_hh2[2 * HR + j]);
                                h2 = (1.0f - zg) * ng;
                            } else {
                                const float ig = sigmoid_r(x2[0] + a.b_hh2[j]);
                                const float gg = tanh_r(x2[2 * HR] + a.b_hh2[2 * HR + j]);
                                const float og = sigmoid_r(x2[(G - 1) * HR] + a.b_hh2[(G - 1) * HR + j]);
                                h2 = og * tanh_r(ig * gg);
                            }
                            a.last_out[(size_t)b * a.ld_last + a.col_off2 + j] = h2;
                        }
                    }
                }
                hprev[bl][r] = hn; cprev[bl][r] = cn;
                uint16_t* d = hpw + c * LDP + j;
                if (H2) {
                    uint32_t hh, ll;
                    nww_split2h(hn * s_h, 0.0f, hh, ll);
                    d[0] = (uint16_t)hh; d[16 * LDP] = (uint16_t)ll;
                } else {
                    uint32_t hi, mid, lo;
                    split3r(hn, hi, mid, lo);
                    d[0] = (uint16_t)(hi >> 16); d[16 * LDP] = (uint16_t)(mid >> 16); d[(NTM - 1) * 16 * LDP] = (uint16_t)(lo >> 16);
                }
            }
        }
        RNN_STAMP(3)
        if constexpr (FIN == 0 && CARRY) {
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < a.steps) fetch(step + 1, xpf[0]);          // (behind the gates that used this step's)
        }
        RNN_STAMP(4)
        __syncthreads();
        RNN_STAMP(5)
    }
}
}  // namespace

// widths 32 / 64 / 128 in every arithmetic; any other multiple of 4 below 128 as a zero-padded instance of the next width (two-term form)
static bool rnn_x3_padded(const GruArgs& a) {
    static const int on = 1;
    return on && a.products == 3 && a.H >= 4 && a.H < 128 && a.H % 4 == 0 && a.H != 32 && a.H != 64 && a.fin == 0;
}
bool rnn_x3_usable(const GruArgs& a) {
    return (a.products == 3 || a.products == 6 || a.products == 9) && (a.H == 32 || a.H == 64 || a.H == 128 || rnn_x3_padded(a)) &&
           (reinterpret_cast<uintptr_t>(a.w_hh) & 15) == 0;
}

hipError_t launch_rnn_x3(const GruArgs& a, int gates, hipStream_t s) {
    if (!rnn_x3_usable(a) || (gates != 3 && gates != 4)) return hipErrorInvalidValue;
    const bool pad = rnn_x3_padded(a);
    const int HP = a.H <= 32 ? 32 : a.H <= 64 ? 64 : 128;     // the instance's width
    const dim3 grid((a.B + 15) / 16), block(HP == 128 ? 256 : 64 * (HP / 16));
    const size_t lds = (size_t)2 * (a.products == 3 ? 2 : 3) * 16 * (HP + 16) * sizeof(uint16_t);     // two sets of h planes
    // H = 128 in the two-term form: eight waves of 16 hidden units (two per SIMD, 144 fragment registers each) instead of four of 32 - a step's
    // products and gate arithmetic per wave halve, and the step is a latency chain: 0.280 -> 0.243 ms (GRU head, B = 2048), 32 -> 23 us (CRNN, B = 16)
    static const int nb1 = 1;
    if (pad) {
#define RNN_PAD(GV, HV) hipLaunchKernelGGL((rnn_x3_kernel<GV, HV, 3, (HV == 128 ? 2 : 1), 0, true>), grid, block, lds, s, a)
#define RNN_PAD8(GV) hipLaunchKernelGGL((rnn_x3_kernel<GV, 128, 3, 1, 0, true>), grid, dim3(512), lds, s, a)
        if (gates == 3) { if (HP == 32) RNN_PAD(3, 32); else if (HP == 64) RNN_PAD(3, 64); else if (nb1) RNN_PAD8(3); else RNN_PAD(3, 128); }
        else { if (HP == 32) RNN_PAD(4, 32); else if (HP == 64) RNN_PAD(4, 64); else RNN_PAD(4, 128); }      // (the padded eight-wave LSTM would need scratch)
#undef RNN_PAD8
#undef RNN_PAD
        return hipGetLastError();
    }
#define RNN_GO(GV, HV)                                                                                                \
    if (a.products == 9) hipLaunchKernelGGL((rnn_x3_kernel<GV, HV, 9, (HV == 128 ? 2 : 1)>), grid, block, lds, s, a);  \
    else if (a.products == 3) hipLaunchKernelGGL((rnn_x3_kernel<GV, HV, 3, (HV == 128 ? 2 : 1)>), grid, block, lds, s, a); \
    else hipLaunchKernelGGL((rnn_x3_kernel<GV, HV, 6, (HV == 128 ? 2 : 1)>), grid, block, lds, s, a);
#define RNN_H(GV)                                                                                                     \
    switch (a.H) {                                                                                                    \
        case 32: RNN_GO(GV, 32) break;                                                                                \
        case 64: RNN_GO(GV, 64) break;                                                                                \
        default: RNN_GO(GV, 128) break;                                                                               \
    }
    if (nb1 && a.H == 128 && a.products == 3 && !pad) {
        const dim3 block8(512);
        if (gates == 3 && a.fin > 0) {
            if ((a.fin != 32 && a.fin != 64) || !a.x_in || !a.w_ih || !a.b_ih || a.reverse) return hipErrorInvalidValue;
            if (a.fin == 32) hipLaunchKernelGGL((rnn_x3_kernel<3, 128, 3, 1, 32>), grid, block8, lds, s, a);
            else hipLaunchKernelGGL((rnn_x3_kernel<3, 128, 3, 1, 64>), grid, block8, lds, s, a);
        } else if (gates == 3) hipLaunchKernelGGL((rnn_x3_kernel<3, 128, 3, 1>), grid, block8, lds, s, a);
        else hipLaunchKernelGGL((rnn_x3_kernel<4, 128, 3, 1>), grid, block8, lds, s, a);
        return hipGetLastError();
    }
    if (gates == 3 && a.fin > 0) {                            // fused input projection (GRU, two-term form)
        if (a.products != 3 || (a.fin != 32 && a.fin != 64) || !a.x_in || !a.w_ih || !a.b_ih || a.reverse) return hipErrorInvalidValue;
#define RNN_FIN(HV, FV) hipLaunchKernelGGL((rnn_x3_kernel<3, HV, 3, (HV == 128 ? 2 : 1), FV>), grid, block, lds, s, a)
        switch (a.H * 100 + a.fin) {
            case 3232: RNN_FIN(32, 32); break;
            case 3264: RNN_FIN(32, 64); break;
            case 6432: RNN_FIN(64, 32); break;
            case 6464: RNN_FIN(64, 64); break;
            case 12832: RNN_FIN(128, 32); break;
            default: RNN_FIN(128, 64); break;
        }
#undef RNN_FIN
        return hipGetLastError();
    }
    if (gates == 3) { RNN_H(3) } else { RNN_H(4) }
#undef RNN_H
#undef RNN_GO
    return hipGetLastError();
}
