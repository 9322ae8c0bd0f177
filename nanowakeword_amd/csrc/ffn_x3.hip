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
#include "layers.h"
#include "ffn_x3.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// ---- plan-time packing: one thread per (hidden block, fragment, lane)
__global__ void __launch_bounds__(256) ffn_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, unsigned char* __restrict__ out, int D) {
    const int D16 = D / 16, NOB = (D + 31) / 32, NHB = D / 8, H4 = 4 * D;
    const int frags = D16 + 2 * NOB;
    const size_t blk = ffn_x3_block_bytes(D);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)NHB * frags * 64) return;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) % frags), hb = (int)((idx >> 6) / frags);
    const int i = lane & 31, h = lane >> 5;
    unsigned char* base = out + (size_t)hb * blk;
    float v[8];
    unsigned char* dst;
    if (f < D16) {                                             // W1 fragment kb = f: row = hidden 32hb + i, k = 16kb + 8h + e
        const int kb = f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W1[(size_t)(32 * hb + i) * D + 16 * kb + 8 * h + e];
        dst = base + ((size_t)(kb * 3) * 64 + lane) * 16;
    } else {                                                   // W2 fragment (ob, kb2): row = out feature 32ob + i, slot e <-> hidden
        const int ob = (f - D16) >> 1, kb2 = (f - D16) & 1;    //   32hb + 8 (2 kb2 + (e >> 2)) + 4h + (e & 3)
        const int m = 32 * ob + i;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = m < D ? W2[(size_t)m * H4 + 32 * hb + 8 * (2 * kb2 + (e >> 2)) + 4 * h + (e & 3)] : 0.0f;
        dst = base + (size_t)D16 * 3072 + ((size_t)((ob * 2 + kb2) * 3) * 64 + lane) * 16;
    }
    uint32_t hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3f(v[e], hh[e], mm[e], ll[e]);
    *reinterpret_cast<uint4*>(dst) = make_uint4(pack16(hh[0], hh[1]), pack16(hh[2], hh[3]), pack16(hh[4], hh[5]), pack16(hh[6], hh[7]));
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(pack16(mm[0], mm[1]), pack16(mm[2], mm[3]), pack16(mm[4], mm[5]), pack16(mm[6], mm[7]));
    *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(pack16(ll[0], ll[1]), pack16(ll[2], ll[3]), pack16(ll[4], ll[5]), pack16(ll[6], ll[7]));
    if (f == 0 && lane < 32) reinterpret_cast<float*>(base + (size_t)D16 * 3072 + (size_t)NOB * 6144)[lane] = b1[32 * hb + lane];
}

template <int D16>
__global__ void __launch_bounds__(256) ffn_x3_kernel(FfnArgs a) {
    constexpr int D = 16 * D16, NOB = (D + 31) / 32, NHB = D / 8;
    constexpr int W1_BYTES = D16 * 3072, W2_BYTES = NOB * 6144, BLK = (W1_BYTES + W2_BYTES + 128 + 4095) & ~4095;
    constexpr int NLD = BLK / 4096;                            // 16-byte pieces per thread and block
    // two separate LDS objects, not two halves of one: hipcc then knows that the reads of one buffer cannot alias the
    // LDS-DMA writes into the other and does not wait for the next block's fetch (s_waitcnt vmcnt(0)) in mid-block
    __shared__ __attribute__((aligned(16))) unsigned char lds_buf0[BLK];
    __shared__ __attribute__((aligned(16))) unsigned char lds_buf1[BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int row = (int)blockIdx.x * 128 + wave * 32 + n;
    const bool row_ok = row < a.M;
    float* hrow = a.h + (size_t)(row_ok ? row : a.M - 1) * D;

    // ---- weight blocks go global -> LDS directly (global_load_lds_dwordx4: lane i of a wave lands at base + 16 i), no
    // staging registers; the first one is on its way while the rows are normalised
    auto fetch_block = [&](int hb, unsigned char* buf) {
        const unsigned char* src = a.packed + (size_t)hb * BLK + tid * 16;
        unsigned char* dst = buf + wave * 1024;                           // wave-uniform
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + j * 4096),
                                             (void __attribute__((address_space(3)))*)(dst + j * 4096), 16, 0, 0);
    };
    fetch_block(0, lds_buf0);

    // ---- LayerNorm of the lane's half row (features 16kb + 8h + e) -> X fragments
    bf16x8 xf[D16][3];
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
            split_frag(y, xf[kb][0], xf[kb][1], xf[kb][2]);
        }
    }

    f32x16 yacc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.0f;

    auto block = [&](const unsigned char* blk) {
        const unsigned char* w1p = blk + lane * 16;
        const unsigned char* w2p = blk + W1_BYTES + lane * 16;
        // ---- Ht = W1 block . Xt
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
        bf16x8 nw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(w1p + t * 1024);
#pragma unroll
        for (int kb = 0; kb < D16; ++kb) {
            bf16x8 cw[3] = {nw[0], nw[1], nw[2]};
            if (kb + 1 < D16) {
#pragma unroll
                for (int t = 0; t < 3; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(w1p + ((kb + 1) * 3 + t) * 1024);
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(w2p + t * 1024);   // first W2 fragment
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma6(cw, xf[kb], acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- swish(Ht + b1), re-split in place: registers 8 kb2 .. 8 kb2 + 7 are the eight k slots of 16-block kb2
        bf16x8 hf[2][3];
        {
            const float* b1p = reinterpret_cast<const float*>(blk + W1_BYTES + W2_BYTES);
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2) {
                const float4 ba = *reinterpret_cast<const float4*>(b1p + 8 * (2 * kb2) + 4 * h);
                const float4 bb = *reinterpret_cast<const float4*>(b1p + 8 * (2 * kb2 + 1) + 4 * h);
                const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = ffn_swish(acc1[8 * kb2 + e] + bias[e]);
                split_frag(y, hf[kb2][0], hf[kb2][1], hf[kb2][2]);
            }
        }
        // ---- Yt += W2 block . swish(Ht)
#pragma unroll
        for (int s = 0; s < 2 * NOB; ++s) {                    // s = 2 ob + kb2
            bf16x8 cw[3] = {nw[0], nw[1], nw[2]};
            if (s + 1 < 2 * NOB) {
#pragma unroll
                for (int t = 0; t < 3; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(w2p + ((s + 1) * 3 + t) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma6(cw, hf[s & 1], yacc[s >> 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int hb = 0; hb < NHB; hb += 2) {                      // NHB = D / 8 is even
        fetch_block(hb + 1, lds_buf1);                         // buffer 1 was last read in block hb - 1, behind a barrier
        block(lds_buf0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (hb + 2 < NHB) fetch_block(hb + 2, lds_buf0);
        block(lds_buf1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- h <- h + rscale * (Yt + b2): lane (row n, half h) holds out features 32 ob + 8 g + 4 h + 0..3
    if (!row_ok) return;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = 32 * ob + 8 * g + 4 * h;
            if (m < D) {                                       // D % 16 == 0, so the four features are in or out together
                const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + m);
                float4 r = *reinterpret_cast<const float4*>(hrow + m);
                r.x += a.rscale * (yacc[ob][4 * g + 0] + b2.x);
                r.y += a.rscale * (yacc[ob][4 * g + 1] + b2.y);
                r.z += a.rscale * (yacc[ob][4 * g + 2] + b2.z);
                r.w += a.rscale * (yacc[ob][4 * g + 3] + b2.w);
                *reinterpret_cast<float4*>(hrow + m) = r;
            }
        }
}

}  // namespace

size_t ffn_x3_packed_bytes(int D) { return (size_t)(D / 8) * ffn_x3_block_bytes(D); }

bool ffn_x3_supported(int D) { return D == 32 || D == 64 || D == 96 || D == 128 || D == 144; }

hipError_t launch_ffn_x3_pack(const float* W1, const float* b1, const float* W2, void* out, int D, hipStream_t s) {
    const size_t total = (size_t)(D / 8) * (D / 16 + 2 * ((D + 31) / 32)) * 64;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W1, b1, W2,
                       reinterpret_cast<unsigned char*>(out), D);
    return hipGetLastError();
}

hipError_t launch_ffn_x3(const FfnArgs& a, int D, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    const dim3 grid((a.M + 127) / 128);
#define FFN_GO(D16V)                                                                                               \
    {                                                                                                              \
                hipLaunchKernelGGL((ffn_x3_kernel<D16V>), grid, dim3(256), 0, s, a);                                       \
    }
    switch (D) {
        case 32: FFN_GO(2) break;
        case 64: FFN_GO(4) break;
        case 96: FFN_GO(6) break;
        case 128: FFN_GO(8) break;
        case 144: FFN_GO(9) break;
        default: return hipErrorInvalidValue;
    }
#undef FFN_GO
    return hipGetLastError();
}
