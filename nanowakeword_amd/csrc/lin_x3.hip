// lin_x3.hip - short-K Linear layers of the Conformer on the bf16 matrix cores (exact operand splitting), input-stationary:
//     out[M][N] = epilogue( (LayerNorm?)(x)[M][K] . W[N][K]^T + bias ),   K = 16 .. 144 (the model width), any N
// for attention.in_proj (plain), attention.out_proj and conv_module.conv2 (+ residual), conv_module.layer_norm + conv1 +
// GLU (N = 2K, out = a * sigmoid(b)) and input_proj (architectures.py:471-543).  The general split-operand GEMM (gemm_x3.hip)
// walks k-tiles per output tile; with K = 144 that is five tiles of prologue and epilogue per 128 x 160 outputs and A re-read
// per column tile: 0.17 - 0.25 ms per layer at M = 206 848 where the matrix pipe needs 0.02 - 0.06.
//
// Same organisation as the fused feed-forward (ffn_x3.hip), minus the second product: a wave owns 32 rows for the whole
// kernel, their (normalised) values live in registers as 3 x K/16 B fragments; the products are computed transposed,
//     Yt [32 outputs x 32 rows] = W block [32 x K] . Xt,
// so a lane of the accumulator holds 4 consecutive output features of ONE row per register group: 16-byte stores; the
// workgroup's four waves share each 32-output block of W through LDS (plan-time packed fragments + biases, fetched one
// block ahead by global_load_lds_dwordx4 into the other of two buffers, one barrier per block).  Two workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "layers.h"
#include "lin_x3.h"
#include "split_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef NWW_TRACE      // tools/ubench/lin_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of its first 16 output blocks
__device__ unsigned long long g_lin_trace[8 * 16 * 8];
#define LIN_STAMP(blk, k) if (blockIdx.x == 0 && (blk) < 16 && lane == 0) g_lin_trace[(wave * 16 + (blk)) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LIN_STAMP(blk, k)
#endif

namespace {

__device__ __forceinline__ void split3l(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack16l(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ void split_frag_l(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3l(v[e], h[e], m[e], l[e]);
    union { uint4 u; bf16x8 b; } ch, cm, cl;
    ch.u = make_uint4(pack16l(h[0], h[1]), pack16l(h[2], h[3]), pack16l(h[4], h[5]), pack16l(h[6], h[7]));
    cm.u = make_uint4(pack16l(m[0], m[1]), pack16l(m[2], m[3]), pack16l(m[4], m[5]), pack16l(m[6], m[7]));
    cl.u = make_uint4(pack16l(l[0], l[1]), pack16l(l[2], l[3]), pack16l(l[4], l[5]), pack16l(l[6], l[7]));
    fh = ch.b; fm = cm.b; fl = cl.b;
}

__device__ __forceinline__ float lin_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }

// six products, small terms first (the order of gemm_x3.hip): w = weight fragments (A operand), x = activation fragments (B)
__device__ __forceinline__ void mfma6l(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
}

// three products of two-term operands (binary16), small terms first: lo*hi, hi*lo, hi*hi
__device__ __forceinline__ void mfma3hl(const bf16x8 (&w)[3], const bf16x8* x, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, x[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[0]), acc, 0, 0, 0);
}

// ---- plan-time packing: one thread per (output block, fragment, lane).  Block = [part (1 | 2: GLU a, b)][kb][term][lane] 16-byte
// fragments, then 32 biases per part, padded to whole 4 KB copy steps.  Part p of block blk is W rows p * gate_off + 32 blk + i.
__global__ void __launch_bounds__(256) lin_pack_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                       unsigned char* __restrict__ out, int K, int n_out, int parts, int gate_off,
                                                       int terms, float ws) {
    const int K16 = K / 16, nblk = (n_out + 31) / 32;
    const size_t blk_bytes = lin_x3_block_bytes(K, parts, terms);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nblk * parts * K16 * 64) return;
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int kb = (int)(r % K16); r /= K16;
    const int part = (int)(r % parts), blk = (int)(r / parts);
    const int i = lane & 31, h = lane >> 5;
    const int col = 32 * blk + i;                              // output feature inside the part
    unsigned char* base = out + (size_t)blk * blk_bytes;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col < n_out ? W[(size_t)(part * gate_off + col) * K + 16 * kb + 8 * h + e] : 0.0f;
    unsigned char* dst = base + ((size_t)((part * K16 + kb) * terms) * 64 + lane) * 16;
    if (terms == 2) {                                          // two binary16 terms of weight x scale (split_h2.h)
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) nww_split2h(v[2 * e] * ws, v[2 * e + 1] * ws, hi[e], lo[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else {
        uint32_t hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3l(v[e], hh[e], mm[e], ll[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack16l(hh[0], hh[1]), pack16l(hh[2], hh[3]), pack16l(hh[4], hh[5]), pack16l(hh[6], hh[7]));
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(pack16l(mm[0], mm[1]), pack16l(mm[2], mm[3]), pack16l(mm[4], mm[5]), pack16l(mm[6], mm[7]));
        *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(pack16l(ll[0], ll[1]), pack16l(ll[2], ll[3]), pack16l(ll[4], ll[5]), pack16l(ll[6], ll[7]));
    }
    if (kb == 0 && lane < 32)
        reinterpret_cast<float*>(base + (size_t)parts * K16 * terms * 1024)[part * 32 + lane] = (bias && col < n_out) ? bias[part * gate_off + col] : 0.0f;
}

// EPI: 0 = out = y + bias;  1 = out = res + rscale * (y + bias);  2 = GLU, out = (ya + bias_a) * sigmoid(yb + bias_b)
// NWV waves per workgroup (32 rows each) share every weight block streamed through LDS
// H2: two binary16 terms per operand with a per-row scale (LinArgs::h2); fragments are carried as 128-bit bags typed bf16x8 either way
template <int K16, int EPI, bool LN, int NWV, bool H2 = false>
__global__ void __launch_bounds__(64 * NWV, 2) lin_x3_kernel(LinArgs a) {
    constexpr int K = 16 * K16, PARTS = EPI == 2 ? 2 : 1, NTM = H2 ? 2 : 3;
    constexpr int FRAG_BYTES = PARTS * K16 * NTM * 1024, BLK = (FRAG_BYTES + PARTS * 128 + 4095) & ~4095;
    // two separate LDS objects: reads of one cannot alias the LDS-DMA writes into the other (no s_waitcnt vmcnt in mid-block)
    __shared__ __attribute__((aligned(16))) unsigned char wb0[BLK];
    __shared__ __attribute__((aligned(16))) unsigned char wb1[BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int row = (int)blockIdx.x * (32 * NWV) + wave * 32 + n;
    const bool row_ok = row < a.M;
    const size_t rr = (size_t)(row_ok ? row : a.M - 1);
    const float* xrow = a.x + rr * a.ldx;

    auto fetch = [&](int blk, unsigned char* buf) {
        if (wave >= 4) return;                                           // the first four waves issue the copy (4 KB per step)
        const unsigned char* sp = a.packed + (size_t)blk * BLK + tid * 16;
        unsigned char* dst = buf + wave * 1024;                          // wave-uniform
#pragma unroll
        for (int j = 0; j < BLK / 4096; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sp + j * 4096),
                                             (void __attribute__((address_space(3)))*)(dst + j * 4096), 16, 0, 0);
    };
    fetch(0, wb0);

    // ---- the lane's half row (features 16kb + 8h + e), LayerNorm-ed if asked -> X fragments
    bf16x8 xf[K16][NTM];
    float pin = 1.0f;                                          // H2: 1 / (the row's scale x the weight scale)
    {
        float v[K16][8];
        float s = 0.0f;
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h);
            const float4 p1 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h + 4);
            v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
            v[kb][4] = p1.x; v[kb][5] = p1.y; v[kb][6] = p1.z; v[kb][7] = p1.w;
            if (LN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[kb][e];
            }
        }
        float mu = 0.0f, rstd = 1.0f;
        if (LN) {
            s += __shfl_xor(s, 32, 64);
            mu = s / (float)K;
            float q = 0.0f;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[kb][e] - mu; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 32, 64);
            rstd = 1.0f / sqrtf(q / (float)K + 1e-5f);
        }
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            if (LN) {
                const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + 16 * kb + 8 * h), w1 = *reinterpret_cast<const float4*>(a.ln_w + 16 * kb + 8 * h + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(a.ln_b + 16 * kb + 8 * h), c1 = *reinterpret_cast<const float4*>(a.ln_b + 16 * kb + 8 * h + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[kb][e] = (v[kb][e] - mu) * rstd * w[e] + c[e];
            }
            if constexpr (!H2) split_frag_l(v[kb], xf[kb][0], xf[kb][1], xf[kb][2]);
        }
        if constexpr (H2) {
            // the row's largest magnitude (both half rows) -> its power-of-two scale: max * s in [2^14, 2^15)
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[kb][e]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
            const float sc = __uint_as_float((268u - eb) << 23);
            pin = __uint_as_float((eb - 14u) << 23) * a.w_un;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) nww_split2h(v[kb][2 * e] * sc, v[kb][2 * e + 1] * sc, hi[e], lo[e]);
                xf[kb][0] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
                xf[kb][1] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
            }
        }
    }

    float* orow = a.out + rr * a.ldc;
    // head-major qkv store (epilogue 0): column part of the destination from a table made once per workgroup, row part here
    // (the q / k / v plane offset is added in size_t at the store: M x N may exceed 2^31 elements)
    __shared__ int qkv_lut[EPI == 0 ? 256 : 1];
    __shared__ unsigned char qkv_which[EPI == 0 ? 256 : 1];
    size_t per_which = 0;
    if (EPI == 0 && a.qkv_T > 0) {
        const int D = a.N / 3, dh = a.qkv_dh, NH = D / dh, T = a.qkv_T;
        per_which = (size_t)(a.M / T) * NH * T * dh;
        for (int c4 = tid; c4 < a.N / 4; c4 += 64 * NWV) {
            const int col = 4 * c4, which = col / D, rem = col - which * D, head = rem / dh, c = rem - head * dh;
            qkv_lut[c4] = head * T * dh + c;
            qkv_which[c4] = (unsigned char)which;
        }
        const int b = (int)(rr / T), t = (int)(rr - (size_t)b * T);
        orow = a.out + ((size_t)b * NH * T + t) * dh;
    }
    const float* rrow = EPI == 1 ? a.res + rr * a.ldres : nullptr;
    // Plain / residual epilogues store through a per-wave LDS transpose: as the accumulator holds them, a store instruction writes 16-byte
    // pieces of 64 different rows (64 cache lines per instruction - tools/ubench/lin_trace: issuing a block's four stores took 2450 of its 7700
    // clocks, the residual's loads likewise); transposed, lane l of store j owns 16 bytes of row 8 j + l / 8 and eight lanes cover a row's 128
    // contiguous bytes.  Same arithmetic per element: bit-identical results.
    // (GLU, measured in round 6: its product a sigmoid(b) formed in the accumulator layout, then through the same transpose - 0.1014 ms against
    // 0.101-0.104 direct, 230 registers instead of 174; four-wave workgroups of 128 rows for a finer tail - 0.120: not kept)
    constexpr bool TR = EPI != 2;
    constexpr int TP = 36;                                     // floats per row of the transpose tile (16-byte writes of 16 lanes: conflict-free)
    __shared__ __attribute__((aligned(16))) float tbuf[TR ? NWV * 32 * TP : 4];
    float* tb = tbuf + (TR ? wave * 32 * TP : 0);
    const int tq = lane & 7;                                   // the lane's 16-byte chunk of a 32-feature block
    bool t_ok[4];
    float* t_orow[4];
    const float* t_rrow[4];
    if constexpr (TR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rj = (int)blockIdx.x * (32 * NWV) + wave * 32 + 8 * j + (lane >> 3);
            t_ok[j] = rj < a.M;
            const size_t rrj = (size_t)(t_ok[j] ? rj : a.M - 1);
            t_orow[j] = a.out + rrj * a.ldc;
            if (EPI == 0 && a.qkv_T > 0) {
                const int T = a.qkv_T, dh = a.qkv_dh, NH = (a.N / 3) / dh;
                const int b = (int)(rrj / T), t = (int)(rrj - (size_t)b * T);
                t_orow[j] = a.out + ((size_t)b * NH * T + t) * dh;
            }
            t_rrow[j] = EPI == 1 ? a.res + rrj * a.ldres : nullptr;
        }
    }
    auto block = [&](int blk, const unsigned char* wbuf) {
        LIN_STAMP(blk, 1)
        const unsigned char* wp = wbuf + lane * 16;
        f32x16 acc[PARTS];
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
        // the residual values of this block's outputs travel during its products
        float4 rres[4];
        if (EPI == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 32 * blk + 4 * tq;
                rres[j] = (t_ok[j] && col < a.N) ? *reinterpret_cast<const float4*>(t_rrow[j] + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        bf16x8 nw[PARTS][3];
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int t = 0; t < NTM; ++t) nw[p][t] = *reinterpret_cast<const bf16x8*>(wp + ((p * K16) * NTM + t) * 1024);
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            bf16x8 cw[PARTS][3];
#pragma unroll
            for (int p = 0; p < PARTS; ++p)
#pragma unroll
                for (int t = 0; t < NTM; ++t) cw[p][t] = nw[p][t];
            if (kb + 1 < K16) {
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
#pragma unroll
                    for (int t = 0; t < NTM; ++t) nw[p][t] = *reinterpret_cast<const bf16x8*>(wp + ((p * K16 + kb + 1) * NTM + t) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < PARTS; ++p) {
                if constexpr (H2) mfma3hl(cw[p], xf[kb], acc[p]);
                else mfma6l(cw[p], reinterpret_cast<const bf16x8(&)[3]>(xf[kb]), acc[p]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // The next weight block's LDS-DMA (issued before this block's products) and the residual loads must have landed before
        // the barrier behind this block - waited for HERE, before the stores: vmcnt counts stores too, and waiting for it after
        // them made every block sit out the write latency of its own outputs (out_proj 0.117 -> see docs/DESIGN_rounds1-5.md 7)
        LIN_STAMP(blk, 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LIN_STAMP(blk, 3)
        if constexpr (H2) {
#pragma unroll
            for (int p = 0; p < PARTS; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] *= pin;
        }
        // lane (row n, half h), register 4g + q = output feature 32 blk + 8g + 4h + q
        if constexpr (TR) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(tb + n * TP + 8 * g + 4 * h) = make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
            __builtin_amdgcn_wave_barrier();                   // (one wave: its LDS operations execute in order)
            const int col = 32 * blk + 4 * tq;
            const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wbuf + FRAG_BYTES) + 4 * tq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 y = *reinterpret_cast<const float4*>(tb + (8 * j + (lane >> 3)) * TP + 4 * tq);
                float4 o = make_float4(y.x + b0.x, y.y + b0.y, y.z + b0.z, y.w + b0.w);
                if (EPI == 1) {
                    const float4 r4 = rres[j];
                    o.x = r4.x + a.rscale * o.x; o.y = r4.y + a.rscale * o.y; o.z = r4.z + a.rscale * o.z; o.w = r4.w + a.rscale * o.w;
                }
                if (t_ok[j] && col < a.N) {                    // N % 4 == 0: the four features are in or out together
                    if (EPI == 0 && a.qkv_T > 0) *reinterpret_cast<float4*>(t_orow[j] + (size_t)qkv_which[col >> 2] * per_which + qkv_lut[col >> 2]) = o;
                    else *reinterpret_cast<float4*>(t_orow[j] + col) = o;
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        if (!row_ok) return;
        const float* bp = reinterpret_cast<const float*>(wbuf + FRAG_BYTES) + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * blk + 8 * g + 4 * h;
            if (col < a.N) {                                   // N % 4 == 0: the four features are in or out together
                const float4 b0 = *reinterpret_cast<const float4*>(bp + 8 * g);
                float4 o = make_float4(acc[0][4 * g] + b0.x, acc[0][4 * g + 1] + b0.y, acc[0][4 * g + 2] + b0.z, acc[0][4 * g + 3] + b0.w);
                if (EPI == 2) {
                    const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 + 8 * g);
                    o.x *= lin_sigmoid(acc[PARTS - 1][4 * g] + b1.x);
                    o.y *= lin_sigmoid(acc[PARTS - 1][4 * g + 1] + b1.y);
                    o.z *= lin_sigmoid(acc[PARTS - 1][4 * g + 2] + b1.z);
                    o.w *= lin_sigmoid(acc[PARTS - 1][4 * g + 3] + b1.w);
                }
                if (EPI == 1) {
                    const float4 r4 = rres[g];
                    o.x = r4.x + a.rscale * o.x; o.y = r4.y + a.rscale * o.y; o.z = r4.z + a.rscale * o.z; o.w = r4.w + a.rscale * o.w;
                }
                if (EPI == 0 && a.qkv_T > 0) *reinterpret_cast<float4*>(orow + (size_t)qkv_which[col >> 2] * per_which + qkv_lut[col >> 2]) = o;
                else *reinterpret_cast<float4*>(orow + col) = o;
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nblk = a.nblk;
    for (int blk = 0; blk < nblk; blk += 2) {
        LIN_STAMP(blk, 0)
        if (blk + 1 < nblk) fetch(blk + 1, wb1);               // buffer 1 was last read in block blk - 1, behind a barrier
        block(blk, wb0);
        LIN_STAMP(blk, 4)
        __syncthreads();
        LIN_STAMP(blk, 5)
        if (blk + 1 < nblk) {
            LIN_STAMP(blk + 1, 0)
            if (blk + 2 < nblk) fetch(blk + 2, wb0);
            block(blk + 1, wb1);
            LIN_STAMP(blk + 1, 4)
            __syncthreads();
            LIN_STAMP(blk + 1, 5)
        }
    }
}

}  // namespace

// K = 192 / 256 (round 6): two-term instances only (the three-term GLU blocks exceed the LDS); one workgroup per CU there
bool lin_x3_supported(int K, int N, bool h2) { return (K == 32 || K == 64 || K == 96 || K == 128 || K == 144 || (h2 && (K == 192 || K == 256))) && N % 4 == 0 && N >= 4; }

size_t lin_x3_packed_bytes(int K, int n_out, int parts, int terms) { return (size_t)((n_out + 31) / 32) * lin_x3_block_bytes(K, parts, terms); }

hipError_t launch_lin_x3_pack(const float* W, const float* bias, void* out, int K, int n_out, int parts, int gate_off, hipStream_t s,
                              int terms, float ws) {
    if (terms != 2 && terms != 3) return hipErrorInvalidValue;
    const size_t total = (size_t)((n_out + 31) / 32) * parts * (K / 16) * 64;
    hipLaunchKernelGGL(lin_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, bias,
                       reinterpret_cast<unsigned char*>(out), K, n_out, parts, gate_off, terms, ws);
    return hipGetLastError();
}

hipError_t launch_lin_x3(const LinArgs& a0, int K, int epi, bool ln, hipStream_t s) {
    if (a0.M <= 0) return hipSuccess;
    if (!lin_x3_supported(K, a0.N, a0.h2 != 0) || (ln && epi != 2) || (a0.ldx % 4) || (a0.ldc % 4)) return hipErrorInvalidValue;
    if (a0.qkv_T > 0 && (epi != 0 || a0.N > 1024 || a0.N % 3 || a0.qkv_dh <= 0 || (a0.N / 3) % a0.qkv_dh || a0.qkv_dh % 4 || a0.M % a0.qkv_T))
        return hipErrorInvalidValue;
    // 16-byte row loads / stores (as the general GEMM's loaders): refuse a misaligned buffer loudly instead of faulting
    if (((reinterpret_cast<uintptr_t>(a0.x) | reinterpret_cast<uintptr_t>(a0.out) | reinterpret_cast<uintptr_t>(a0.res)) & 15) != 0) return hipErrorInvalidValue;
    LinArgs a = a0;
    a.nblk = (a.N + 31) / 32;
    // eight waves per workgroup for the GLU epilogue only: its four-wave instances need 212 - 264 registers and do not reach two
    // workgroups per CU (LayerNorm + conv1 + GLU of the Conformer 0.155 -> 0.138 ms); the other epilogues measured slower with
    // eight (in_proj 0.170 -> 0.199, out_proj / conv2 0.117 -> 0.125-0.129).  K >= 128 has no four-wave GLU instance at all (one
    // workgroup per CU, "final occupancy 1"); the row arithmetic is the same in both shapes, so results do not depend on the choice.
    // (two-term instances, 138 - 148 registers at K = 144: eight waves measured again - in_proj 0.148 -> 0.161, out_proj unchanged)
    const bool w8 = epi == 2 && K <= 144 && (K >= 128 || a.M >= 256 * 256);      // (K > 144: four waves - the eight-wave form would spill at 256 registers)
    const dim3 grid(w8 ? (a.M + 255) / 256 : (a.M + 127) / 128);
#define LIN_GO2(K16V, EPIV, LNV, H2V)                                                                              \
    if constexpr (EPIV == 2 && K16V >= 8 && K16V <= 9) hipLaunchKernelGGL((lin_x3_kernel<K16V, EPIV, LNV, 8, H2V>), grid, dim3(512), 0, s, a); \
    else if constexpr (K16V > 9) hipLaunchKernelGGL((lin_x3_kernel<K16V, EPIV, LNV, 4, H2V>), grid, dim3(256), 0, s, a);                        \
    else if (EPIV == 2 && w8) hipLaunchKernelGGL((lin_x3_kernel<K16V, EPIV, LNV, (EPIV == 2 ? 8 : 4), H2V>), grid, dim3(512), 0, s, a); \
    else hipLaunchKernelGGL((lin_x3_kernel<K16V, EPIV, LNV, 4, H2V>), grid, dim3(256), 0, s, a);
#define LIN_GO(K16V, EPIV, LNV)                                                                                    \
    if (a.h2) { LIN_GO2(K16V, EPIV, LNV, true) } else { LIN_GO2(K16V, EPIV, LNV, false) }
#define LIN_GO_H2ONLY(K16V, EPIV, LNV)                                                                             \
    if (a.h2) { LIN_GO2(K16V, EPIV, LNV, true) } else return hipErrorInvalidValue;
#define LIN_EPI(K16V)                                                                                              \
    if (epi == 0) { LIN_GO(K16V, 0, false) } else if (epi == 1) { LIN_GO(K16V, 1, false) } else if (ln) { LIN_GO(K16V, 2, true) } else { LIN_GO(K16V, 2, false) }
#define LIN_EPI_H2(K16V)                                                                                           \
    if (epi == 0) { LIN_GO_H2ONLY(K16V, 0, false) } else if (epi == 1) { LIN_GO_H2ONLY(K16V, 1, false) } else if (ln) { LIN_GO_H2ONLY(K16V, 2, true) } else { LIN_GO_H2ONLY(K16V, 2, false) }
    switch (K) {
        case 32: LIN_EPI(2) break;
        case 64: LIN_EPI(4) break;
        case 96: LIN_EPI(6) break;
        case 128: LIN_EPI(8) break;
        case 144: LIN_EPI(9) break;
        case 192: LIN_EPI_H2(12) break;
        case 256: LIN_EPI_H2(16) break;
        default: return hipErrorInvalidValue;
    }
#undef LIN_EPI_H2
#undef LIN_GO_H2ONLY
#undef LIN_EPI
#undef LIN_GO
#undef LIN_GO2
    return hipGetLastError();
}
