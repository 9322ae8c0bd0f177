// dual_x3.hip - the pointwise half of a BcResNet block (architectures.py:632-647) on the bf16 matrix cores by exact operand
// splitting, input-stationary:
//     out[m][n] = BN_s( xs[m] . Wsc[n] )  +  act( BN_1( d[m] . Wpw[n] ) )          m = pixel (B x Ho x Wo), K = C_in = 32 / 64 / 128
// (d = depthwise output, xs = the block input at the strided centres; activation BEFORE the residual add).  The float32-MFMA
// dual GEMM this replaces (layers.hip: gemm_lds_kernel<ACT, 1>) is matrix-pipe-bound for the wider blocks (0.41 / 0.69 ms
// for blocks 2 / 3 at 8192 clips, where the bf16 pipe needs a sixth of the float32 pipe's time even with six products).
//
// Organisation of lin_x3.hip with two inputs: a wave owns 32 pixels for the whole kernel, BOTH of their rows live in
// registers as 3 x K/16 B fragments each; per 32-output block the products are computed transposed,
//     Pt [32 outputs x 32 pixels] = Wpw block . Dt,     St = Wsc block . Xst,
// so a lane holds 4 consecutive output channels of ONE pixel per register group (16-byte stores); the workgroup's four
// waves share each block's weights through LDS (plan-time packed fragments + the four folded-BN vectors, fetched one block
// ahead by global_load_lds_dwordx4 into the other of two buffers, one barrier per block).
// Tried and dropped: computing d and xs here from the block input (depthwise 3x3 once per pixel, no d / xs round trip) -
// correct, but hipcc needs 256 VGPRs + 256 AGPRs + scratch for the nine-tap loader next to the resident fragments (K = 64:
// 0.86 ms against 0.30 + 0.24 apart; K = 128: 1.73 against 0.32 + 0.45).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "dual_x3.h"
#include "split_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4d __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void split3d(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack16d(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ void split_frag_d(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3d(v[e], h[e], m[e], l[e]);
    union { uint4 u; bf16x8 b; } ch, cm, cl;
    ch.u = make_uint4(pack16d(h[0], h[1]), pack16d(h[2], h[3]), pack16d(h[4], h[5]), pack16d(h[6], h[7]));
    cm.u = make_uint4(pack16d(m[0], m[1]), pack16d(m[2], m[3]), pack16d(m[4], m[5]), pack16d(m[6], m[7]));
    cl.u = make_uint4(pack16d(l[0], l[1]), pack16d(l[2], l[3]), pack16d(l[4], l[5]), pack16d(l[6], l[7]));
    fh = ch.b; fm = cm.b; fl = cl.b;
}

template <int ACT>
__device__ __forceinline__ float dual_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return nww_silu(v);
    return v;
}

// six products, small terms first (the order of gemm_x3.hip): w = weight fragments (A operand), x = activation fragments (B)
// SW (the fused-mean instances): the activation term is the A operand and the weight term the B operand - the accumulator is then
// [pixels x output channels], a lane holds 16 pixels of ONE channel (same fragments, the other orientation of the same product)
template <bool SW>
__device__ __forceinline__ f32x16 mm_bf16(bf16x8 w, bf16x8 x, f32x16 acc) {
    return SW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0);
}
template <bool SW>
__device__ __forceinline__ f32x16 mm_f16(bf16x8 w, bf16x8 x, f32x16 acc) {
    return SW ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), acc, 0, 0, 0)
              : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
template <bool SW = false>
__device__ __forceinline__ void mfma6d(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x16& acc) {
    acc = mm_bf16<SW>(w[1], x[1], acc);
    acc = mm_bf16<SW>(w[2], x[0], acc);
    acc = mm_bf16<SW>(w[0], x[2], acc);
    acc = mm_bf16<SW>(w[1], x[0], acc);
    acc = mm_bf16<SW>(w[0], x[1], acc);
    acc = mm_bf16<SW>(w[0], x[0], acc);
}

// plan-time packing: block = [part 0 = Wpw | 1 = Wsc][kb][term][lane] 16-byte fragments, then a1[32], b1[32], as[32], bs[32]
__global__ void __launch_bounds__(256) dual_pack_kernel(const float* __restrict__ Wpw, const float* __restrict__ Wsc,
                                                        const float* __restrict__ a1, const float* __restrict__ b1,
                                                        const float* __restrict__ as, const float* __restrict__ bs,
                                                        unsigned char* __restrict__ out, int K, int N, int terms, DualPackScales sc) {
    const int K16 = K / 16, nblk = (N + 31) / 32;
    const size_t blk_bytes = dual_x3_block_bytes(K, terms);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nblk * 2 * K16 * 64) return;
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int kb = (int)(r % K16); r /= K16;
    const int part = (int)(r & 1), blk = (int)(r >> 1);
    const int i = lane & 31, h = lane >> 5;
    const int col = 32 * blk + i;
    const float* W = part ? Wsc : Wpw;
    unsigned char* base = out + (size_t)blk * blk_bytes;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col < N ? W[(size_t)col * K + 16 * kb + 8 * h + e] : 0.0f;
    unsigned char* dst = base + ((size_t)((part * K16 + kb) * terms) * 64 + lane) * 16;
    if (terms == 2) {                                          // two binary16 terms of weight x scale (split_h2.h)
        const float ws = part ? sc.sc_ws : sc.pw_ws;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) nww_split2h(v[2 * e] * ws, v[2 * e + 1] * ws, hi[e], lo[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else {
        uint32_t hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3d(v[e], hh[e], mm[e], ll[e]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack16d(hh[0], hh[1]), pack16d(hh[2], hh[3]), pack16d(hh[4], hh[5]), pack16d(hh[6], hh[7]));
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(pack16d(mm[0], mm[1]), pack16d(mm[2], mm[3]), pack16d(mm[4], mm[5]), pack16d(mm[6], mm[7]));
        *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(pack16d(ll[0], ll[1]), pack16d(ll[2], ll[3]), pack16d(ll[4], ll[5]), pack16d(ll[6], ll[7]));
    }
    if (kb == 0 && part == 0 && lane < 32) {
        float* aff = reinterpret_cast<float*>(base + (size_t)2 * K16 * terms * 1024);
        const bool ok = col < N;
        aff[lane] = ok ? (a1 ? a1[col] : 1.0f) * sc.pw_un : 0.0f;
        aff[32 + lane] = ok && b1 ? b1[col] : 0.0f;
        aff[64 + lane] = ok ? (as ? as[col] : 1.0f) * sc.sc_un : 0.0f;
        aff[96 + lane] = ok && bs ? bs[col] : 0.0f;
    }
}

// three products of a float32 weight (three bf16 terms) with a bf16 activation, small terms first
template <bool SW = false>
__device__ __forceinline__ void mfma3d(const bf16x8 (&w)[3], const bf16x8& x, f32x16& acc) {
    acc = mm_bf16<SW>(w[2], x, acc);
    acc = mm_bf16<SW>(w[1], x, acc);
    acc = mm_bf16<SW>(w[0], x, acc);
}
// two products of a scaled float32 weight (two binary16 terms) with a binary16 activation, small term first
template <bool SW = false>
__device__ __forceinline__ void mfma2h(const bf16x8 (&w)[3], const bf16x8& x, f32x16& acc) {
    acc = mm_f16<SW>(w[1], x, acc);
    acc = mm_f16<SW>(w[0], x, acc);
}

// three products of two-term operands (binary16), small terms first: lo*hi, hi*lo, hi*hi
template <bool SW = false>
__device__ __forceinline__ void mfma3h(const bf16x8 (&w)[3], const bf16x8* x, f32x16& acc) {
    acc = mm_f16<SW>(w[1], x[0], acc);
    acc = mm_f16<SW>(w[0], x[1], acc);
    acc = mm_f16<SW>(w[0], x[0], acc);
}

// AT: 0 float32 activations on three bf16 terms (six products), 1 bf16 in and out, 2 scaled binary16 in and out (DualArgs::act16),
// 3 float32 activations on two binary16 terms with a per-pixel scale (DualArgs::h2; three products).  Fragments are carried as
// 128-bit bags typed bf16x8 either way.  NWV waves per workgroup (32 pixels each) share every weight block: a
// workgroup streams ALL packed weights (K = 128: 8 x 52 KB) through LDS once per 32 NWV pixels, which at four waves was the
// kernel's bound for the wide blocks (1.4 GB of L2 -> LDS traffic for block 3 at 8192 clips) - eight waves halve it.
// MEAN: the global average pool fused behind the last block (DualArgs::mean_out): waves are (clip, 32-pixel group) pairs and the
// products run in the OTHER orientation (activations as the A operand): a lane of the accumulator holds 16 pixels of ONE output
// channel, so the per-channel pixel sum is 16 register adds and one cross-half exchange - no LDS tile.  Pixels behind a clip's last
// one carry zero fragments (their constant bs + act(b1) is subtracted); with float32 activations the power-of-two scale is the
// wave's (32 pixels of ONE clip: still a function of the clip alone), a scalar folded into the block's BN factors.  The groups of a
// clip are added in group order by the clip's first wave.
template <int K16, int ACT, int AT, int NWV, bool MEAN = false>
__global__ void __launch_bounds__(64 * NWV) dual_x3_kernel(DualArgs a) {
    constexpr int K = 16 * K16;
    constexpr bool BF = AT == 1 || AT == 2;                              // 16-bit activations in HBM
    constexpr int NTM = AT >= 2 ? 2 : 3;                                 // terms per weight
    constexpr int NXT = BF ? 1 : AT == 3 ? 2 : 3;                        // terms per activation
    constexpr int FRAG_BYTES = 2 * K16 * NTM * 1024, BLK = (FRAG_BYTES + 512 + 4095) & ~4095;
    // two separate LDS objects: reads of one cannot alias the LDS-DMA writes into the other (no s_waitcnt vmcnt in mid-block)
    __shared__ __attribute__((aligned(16))) unsigned char wb0[BLK];
    __shared__ __attribute__((aligned(16))) unsigned char wb1[BLK];
    __shared__ float mpart[MEAN ? 2 * NWV * 32 : 4];                     // per block parity: the waves' channel sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    int row;
    bool row_ok;
    int gpc = 1, clip = 0, grp = 0;                                      // MEAN: groups per clip, this wave's clip and group
    float n_inv = 0.0f;                                                  // MEAN: the wave's pixels behind its clip's last one
    if constexpr (MEAN) {
        gpc = (a.mean_P + 31) / 32;
        const int wg = (int)blockIdx.x * NWV + wave;
        clip = wg / gpc; grp = wg - clip * gpc;
        n_inv = (float)(32 - min(max(a.mean_P - 32 * grp, 0), 32));
        const int pix = 32 * grp + n;
        row = clip * a.mean_P + pix;
        row_ok = pix < a.mean_P && row < a.M;
    } else {
        row = (int)blockIdx.x * (32 * NWV) + wave * 32 + n;
        row_ok = row < a.M;
    }
    const size_t rr = (size_t)(row_ok ? row : a.M - 1);

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

    // ---- the lane's half rows (features 16kb + 8h + e) of d and xs -> fragments
    bf16x8 xf[2][K16][NXT];
    float pin[2] = {1.0f, 1.0f};                               // AT == 3: 1 / the pixel's scale, per input
    const float* xs_row = nullptr;
    size_t xs_off = 0;                                         // element offset of the pixel's row in x (BF: 2-byte elements)
    if (a.x) {
        const int per = a.Ho * a.Wo;
        const int b = (int)(rr / per), r2 = (int)(rr - (size_t)b * per), oy = r2 / a.Wo, ox = r2 - oy * a.Wo;
        xs_off = (((size_t)b * a.H + (size_t)oy * a.sh) * a.W + (size_t)ox * a.sw) * K;
        xs_row = a.x + xs_off;
    }
    if constexpr (BF) {
        // a bf16 row IS the B fragment: eight channels = one 16-byte load
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const __bf16* xrow = p ? (a.x ? reinterpret_cast<const __bf16*>(a.x) + xs_off : reinterpret_cast<const __bf16*>(a.xs) + rr * K)
                                   : reinterpret_cast<const __bf16*>(a.d) + rr * K;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                u32x4d q = *reinterpret_cast<const u32x4d*>(xrow + 16 * kb + 8 * h);
                if (MEAN && !row_ok) q = u32x4d{0u, 0u, 0u, 0u};
                xf[p][kb][0] = __builtin_bit_cast(bf16x8, q);
            }
        }
    } else if constexpr (AT == 3) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float* xrow = (p ? (a.x ? xs_row : a.xs + rr * K) : a.d + rr * K);
            float v[K16][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                const float4 p0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h);
                const float4 p1 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h + 4);
                v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
                v[kb][4] = p1.x; v[kb][5] = p1.y; v[kb][6] = p1.z; v[kb][7] = p1.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[kb][e]));
            }
            // the row's largest magnitude (both half rows) -> its power-of-two scale: max * s in [2^14, 2^15)
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if constexpr (MEAN) {                              // ... of the wave's 32 pixels (one clip's), invalid pixels left out and zeroed
                if (!row_ok) m = 0.0f;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            }
            const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
            const float sc = (MEAN && !row_ok) ? 0.0f : __uint_as_float((268u - eb) << 23);
            pin[p] = __uint_as_float((eb - 14u) << 23);
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) nww_split2h(v[kb][2 * e] * sc, v[kb][2 * e + 1] * sc, hi[e], lo[e]);
                xf[p][kb][0] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
                xf[p][kb][1] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // xs: its own [M][K] rows, or (a.x) the block input x [B][H][W][K] read at the strided centre of the pixel - the
            // depthwise kernel then does not write a copy of those rows
            const float* xrow = (p ? (a.x ? xs_row : a.xs + rr * K) : a.d + rr * K);
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                const float4 p0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h);
                const float4 p1 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h + 4);
                const float z = (MEAN && !row_ok) ? 0.0f : 1.0f;
                const float v[8] = {p0.x * z, p0.y * z, p0.z * z, p0.w * z, p1.x * z, p1.y * z, p1.z * z, p1.w * z};
                split_frag_d(v, xf[p][kb][0], xf[p][kb][1], xf[p][kb][2]);
            }
        }
    }

    float* orow = a.out + rr * a.N;
    auto block = [&](int blk, const unsigned char* wbuf) {
        const unsigned char* wp = wbuf + lane * 16;
        f32x16 acc[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
        bf16x8 nw[2][3];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < NTM; ++t) nw[p][t] = *reinterpret_cast<const bf16x8*>(wp + ((p * K16) * NTM + t) * 1024);
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            bf16x8 cw[2][3];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int t = 0; t < NTM; ++t) cw[p][t] = nw[p][t];
            if (kb + 1 < K16) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int t = 0; t < NTM; ++t) nw[p][t] = *reinterpret_cast<const bf16x8*>(wp + ((p * K16 + kb + 1) * NTM + t) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AT == 3) {
                mfma3h<MEAN>(cw[0], xf[0][kb], acc[0]);
                mfma3h<MEAN>(cw[1], xf[1][kb], acc[1]);
            } else if constexpr (AT == 2) {
                mfma2h<MEAN>(cw[0], xf[0][kb][0], acc[0]);
                mfma2h<MEAN>(cw[1], xf[1][kb][0], acc[1]);
            } else if constexpr (AT == 1) {
                mfma3d<MEAN>(cw[0], xf[0][kb][0], acc[0]);
                mfma3d<MEAN>(cw[1], xf[1][kb][0], acc[1]);
            } else {
                mfma6d<MEAN>(cw[0], xf[0][kb], acc[0]);
                mfma6d<MEAN>(cw[1], xf[1][kb], acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next weight block's LDS-DMA must have landed before the barrier behind this block: waited for here, BEFORE the
        // stores (vmcnt counts them too - waiting after them made every block sit out the write latency of its own outputs)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MEAN) {
            // lane (channel n, half h), register 4g + q = pixel 8g + 4h + q of the wave's 32: the channel's folded BN factors are four
            // scalars, the pixel sum 16 adds in register order + the other half - the same order for every clip and slot
            const float* affn = reinterpret_cast<const float*>(wbuf + FRAG_BYTES) + n;
            float a1c = affn[0], asc = affn[64];
            const float b1c = affn[32], bsc = affn[96];
            if constexpr (AT == 3) { a1c *= pin[0]; asc *= pin[1]; }
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += (acc[1][r] * asc + bsc) + dual_act<ACT>(acc[0][r] * a1c + b1c);
            sum += __shfl_xor(sum, 32, 64);
            sum -= n_inv * (bsc + dual_act<ACT>(b1c));         // the zero-fragment pixels behind the clip's last one
            if (h == 0) mpart[((blk & 1) * NWV + wave) * 32 + n] = sum;
            return;
        }
        if constexpr (AT == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] *= pin[0]; acc[1][r] *= pin[1]; }
        }
        // lane (pixel n, half h), register 4g + q = output channel 32 blk + 8g + 4h + q
        if (!row_ok) return;
        const float* aff = reinterpret_cast<const float*>(wbuf + FRAG_BYTES) + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * blk + 8 * g + 4 * h;
            if (col < a.N) {                                   // N % 4 == 0: the four channels are in or out together
                const float4 a1 = *reinterpret_cast<const float4*>(aff + 8 * g), b1 = *reinterpret_cast<const float4*>(aff + 32 + 8 * g);
                const float4 as = *reinterpret_cast<const float4*>(aff + 64 + 8 * g), bs = *reinterpret_cast<const float4*>(aff + 96 + 8 * g);
                float4 o;
                o.x = (acc[1][4 * g + 0] * as.x + bs.x) + dual_act<ACT>(acc[0][4 * g + 0] * a1.x + b1.x);
                o.y = (acc[1][4 * g + 1] * as.y + bs.y) + dual_act<ACT>(acc[0][4 * g + 1] * a1.y + b1.y);
                o.z = (acc[1][4 * g + 2] * as.z + bs.z) + dual_act<ACT>(acc[0][4 * g + 2] * a1.z + b1.z);
                o.w = (acc[1][4 * g + 3] * as.w + bs.w) + dual_act<ACT>(acc[0][4 * g + 3] * a1.w + b1.w);
                if (BF) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.out) + rr * a.N + col) =
                        make_uint2(nww_pk_act16(AT, o.x, o.y, a.out_mul), nww_pk_act16(AT, o.z, o.w, a.out_mul));
                else *reinterpret_cast<float4*>(orow + col) = o;
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nblk = a.nblk;
    // MEAN: behind the barrier that ends block blk, the first wave of every clip adds the groups' sums (group order) and stores the
    // clip's 32 channel means; mpart alternates with the block parity, so the next block's sums do not disturb the reads
    auto finish_mean = [&](int blk) {
        if constexpr (MEAN) {
            if (grp == 0 && h == 0 && (size_t)clip * a.mean_P < (size_t)a.M && 32 * blk + n < a.N) {
                float sum = mpart[((blk & 1) * NWV + wave) * 32 + n];
                for (int j = 1; j < gpc; ++j) sum += mpart[((blk & 1) * NWV + wave + j) * 32 + n];
                a.mean_out[(size_t)clip * a.N + 32 * blk + n] = sum * (1.0f / (float)a.mean_P);
            }
        }
    };
    for (int blk = 0; blk < nblk; blk += 2) {
        if (blk + 1 < nblk) fetch(blk + 1, wb1);               // buffer 1 was last read in block blk - 1, behind a barrier
        block(blk, wb0);
        __syncthreads();
        finish_mean(blk);
        if (blk + 1 < nblk) {
            if (blk + 2 < nblk) fetch(blk + 2, wb0);
            block(blk + 1, wb1);
            __syncthreads();
            finish_mean(blk + 1);
        }
    }
}

}  // namespace

bool dual_x3_supported(int K, int N) { return (K == 32 || K == 64 || K == 128) && N % 4 == 0 && N >= 4; }
// the fused mean needs every clip's 32-pixel groups inside one four-wave workgroup
bool dual_x3_mean_supported(int pixels_per_clip) { return pixels_per_clip >= 1 && pixels_per_clip <= 128 && 4 % ((pixels_per_clip + 31) / 32) == 0; }

size_t dual_x3_packed_bytes(int K, int N, int terms) { return (size_t)((N + 31) / 32) * dual_x3_block_bytes(K, terms); }

hipError_t launch_dual_x3_pack(const float* Wpw, const float* Wsc, const float* a1, const float* b1, const float* as,
                               const float* bs, void* out, int K, int N, hipStream_t s, int terms, DualPackScales sc) {
    if (terms != 2 && terms != 3) return hipErrorInvalidValue;
    const size_t total = (size_t)((N + 31) / 32) * 2 * (K / 16) * 64;
    hipLaunchKernelGGL(dual_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Wpw, Wsc, a1, b1, as, bs,
                       reinterpret_cast<unsigned char*>(out), K, N, terms, sc);
    return hipGetLastError();
}

hipError_t launch_dual_x3(const DualArgs& a0, int K, int act, hipStream_t s) {
    if (a0.M <= 0) return hipSuccess;
    if (!dual_x3_supported(K, a0.N)) return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(a0.d) | reinterpret_cast<uintptr_t>(a0.xs) | reinterpret_cast<uintptr_t>(a0.out) | reinterpret_cast<uintptr_t>(a0.x)) & 15) != 0)
        return hipErrorInvalidValue;
    if (a0.x && (a0.Ho <= 0 || a0.Wo <= 0 || a0.M % (a0.Ho * a0.Wo) != 0)) return hipErrorInvalidValue;
    DualArgs a = a0;
    if (a.act16) a.h2 = 0;
    a.nblk = (a.N + 31) / 32;
    if (a.mean_out) {
        if (!dual_x3_mean_supported(a.mean_P) || a.M % a.mean_P != 0 || K != 128) return hipErrorInvalidValue;      // (the last block: K = 128)
        const int gpc = (a.mean_P + 31) / 32, clips = a.M / a.mean_P;
        // bf16 activations: eight waves per workgroup as in the unfused launch (half the weight traffic through LDS; 8 % gpc == 0 too)
        // 16-bit activations: eight waves (one workgroup per CU, half the weight traffic through LDS).  Two-term float32 form: FOUR -
        // without the mean tile two four-wave workgroups share a CU (72 KB of weight buffers each), one loads its rows (262 KB per 256
        // pixels: as long as its products at the CU's share of HBM) while the other multiplies: 0.305 ms at eight waves, 0.266 at four
        const bool w8m = a.act16 && a.M >= 256 * 256;
        const dim3 gridm((unsigned)(((size_t)clips * gpc + (w8m ? 7 : 3)) / (w8m ? 8 : 4)));
#define DUAL_MEAN(ACTV)                                                                                            \
        if (a.act16 == 2) {                                                                                        \
            if (w8m) hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 2, 8, true>), gridm, dim3(512), 0, s, a);         \
            else hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 2, 4, true>), gridm, dim3(256), 0, s, a);             \
        } else if (w8m) hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 1, 8, true>), gridm, dim3(512), 0, s, a);        \
        else if (a.act16) hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 1, 4, true>), gridm, dim3(256), 0, s, a);    \
        else if (a.h2) hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 3, 4, true>), gridm, dim3(256), 0, s, a);       \
        else hipLaunchKernelGGL((dual_x3_kernel<8, ACTV, 0, 4, true>), gridm, dim3(256), 0, s, a);
        switch (act) {
            case ACT_RELU: DUAL_MEAN(ACT_RELU) break;
            case ACT_GELU: DUAL_MEAN(ACT_GELU) break;
            case ACT_SILU: DUAL_MEAN(ACT_SILU) break;
            default: return hipErrorInvalidValue;
        }
#undef DUAL_MEAN
        return hipGetLastError();
    }
    // eight waves per workgroup where it measured faster at 8192 clips: bf16 activations at K = 32 (0.255 -> 0.211 ms) and
    // K = 128 (0.293 -> 0.229); K = 64 (0.161 -> 0.172) and every float32 shape (0.363 -> 0.395, 0.235 -> 0.248; K = 128
    // needs 284 registers) stay at four
    const bool w8 = a.M >= 256 * 256 && a.act16 && K != 64;
    const dim3 grid(w8 ? (a.M + 255) / 256 : (a.M + 127) / 128);
#define DUAL_GO(K16V, ACTV)                                                                                        \
    if (a.act16 == 2) {                                                                                            \
        if (w8) hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 2, 8>), grid, dim3(512), 0, s, a);                  \
        else hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 2, 4>), grid, dim3(256), 0, s, a);                     \
    } else if (w8) {                                                                                               \
        hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 1, 8>), grid, dim3(512), 0, s, a);                          \
    } else {                                                                                                       \
        if (a.act16) hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 1, 4>), grid, dim3(256), 0, s, a);             \
        else if (a.h2) hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 3, 4>), grid, dim3(256), 0, s, a);           \
        else hipLaunchKernelGGL((dual_x3_kernel<K16V, ACTV, 0, 4>), grid, dim3(256), 0, s, a);                     \
    }
#define DUAL_ACT(K16V)                                                                                             \
    switch (act) {                                                                                                 \
        case ACT_RELU: DUAL_GO(K16V, ACT_RELU) break;                                                              \
        case ACT_GELU: DUAL_GO(K16V, ACT_GELU) break;                                                              \
        case ACT_SILU: DUAL_GO(K16V, ACT_SILU) break;                                                              \
        default: return hipErrorInvalidValue;                                                                      \
    }
    switch (K) {
        case 32: DUAL_ACT(2) break;
        case 64: DUAL_ACT(4) break;
        case 128: DUAL_ACT(8) break;
        default: return hipErrorInvalidValue;
    }
#undef DUAL_ACT
#undef DUAL_GO
    return hipGetLastError();
}
