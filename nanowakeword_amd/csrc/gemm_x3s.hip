// gemm_x3s.hip - the split-operand GEMM of gemm_x3.hip with the work divided BY WAVE ROLE instead of by phase.
//
// gemm_x3_kernel runs every wave through  wait-loads -> split + LDS store -> barrier -> MFMAs -> barrier  per 32-k tile;
// its phases add up (measured by skipping them one at a time: loads + split + MFMA + epilogue + skeleton = the whole
// time), because the two workgroups of a CU execute identical work and stay in the same phase.  On gfx950 a wave of
// bf16 MFMAs with AGPR accumulators and a wave of VALU work on the same SIMD run truly concurrently
// (tools/ubench/mfma_valu_overlap.hip: max(a, b), not a + b), so here the overlap is built into the workgroup:
//   waves 4-7 (producers)  global loads (A float32, W pre-split bf16), exact 3-way split of A, LDS stores into stage j & 1
//   waves 0-3 (consumers)  fragment reads from stage (j - 1) & 1 one column block ahead of their MFMAs, accumulators in AGPRs
//   one workgroup barrier per k-tile; every SIMD hosts one wave of each kind (waves are dealt round-robin over the SIMDs).
// Per-output arithmetic is IDENTICAL to gemm_x3_kernel: same k-tile sequence, same six products in the same order, same
// split-K chunks - the two kernels give bit-identical results (the small-M instance of gemm_x3.hip keeps serving M <= 64).
// LDS: two stages x (128 + BN) rows x 208 B (BN = 128: 106 KB, one workgroup per CU).  A 512-thread workgroup has 256
// registers per lane, which hipcc splits 128 VGPRs + 128 AGPRs once AGPRs are in play: consumers need ~100 + 16 CB,
// producers ~110 + 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int XS_ROW = 208;                      // bytes per LDS row: 3 terms x 32 k x bf16 + 16 pad (conflict-free b128 reads)
constexpr int XS_BM = 128;

__device__ __forceinline__ float xs_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
template <int ACT>
__device__ __forceinline__ float xs_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == ACT_SILU) return v * xs_sigmoid(v);
    if (ACT == ACT_SIGMOID) return xs_sigmoid(v);
    return v;
}
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

template <int CB, int ACT>
__global__ void __launch_bounds__(512) gemm_x3s_kernel(GemmArgs g) {
    { float agpr_hint = 0.0f; asm volatile("; mfma accumulators in AGPRs" : "+a"(agpr_hint)); }
    constexpr int BN = 32 * CB;
    constexpr int STAGE = (XS_BM + BN) * XS_ROW;
    constexpr int WPIECES = BN * 12;                           // 16-byte pieces of a W tile (32 k x 3 terms per row)
    constexpr int WLD = (WPIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int bm = blockIdx.x * XS_BM, bn = blockIdx.y * BN;
    const int KB = (g.K + 15) >> 4, KT = (g.K + 31) >> 5;
    int kt_begin = 0, kt_end = KT;
    if (g.splitk > 1) {
        const int kc = (KT + g.splitk - 1) / g.splitk;
        kt_begin = blockIdx.z * kc;
        kt_end = min(KT, kt_begin + kc);
    }
    const int nt = kt_end - kt_begin;

    if (producer) {
        const int pt = tid - 256;                              // 0..255 over the four producer waves
        const uint4* Wx = reinterpret_cast<const uint4*>(g.Wx3);
        const int lr = pt >> 3, lq = pt & 7;                   // rows lr + 32 q, floats 4 lq .. +3 of the 32-k tile
        const float* arow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            arow[q] = g.a_blocked ? g.A + ((size_t)(bm >> 7) * g.a_blocked * 128 + lr + 32 * q) * 32 + 4 * lq
                                  : g.A + (size_t)min(bm + lr + 32 * q, g.M - 1) * g.lda + 4 * lq;
        const int a_kstep = g.a_blocked ? 128 * 32 : 32;
        // W piece -> (row, 16-byte column) of the tile and its source, fixed for the whole k loop
        int wrow[WLD], wdst[WLD];
        size_t wsrc[WLD];
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int p = pt + 256 * j;
            const int row = (WPIECES % 256 == 0 || p < WPIECES) ? p / 12 : 0, c = p - (p / 12) * 12;
            const int c6 = c % 6;
            wrow[j] = c / 6;                                   // which 16-k block of the tile
            wdst[j] = row * XS_ROW + (c6 >> 1) * 64 + (c / 6) * 32 + (c6 & 1) * 16;
            wsrc[j] = (size_t)min(bn + row, g.N - 1) * KB * 6 + c6;
        }
        float4 a0[4], a1[4];                                   // two register stages of A (HBM latency), one of W (L2)
        uint4 w0[WLD];
        auto load_a = [&](int kt, float4 (&a)[4]) {
            const int k = kt * 32 + 4 * lq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* p = arow[q] + (size_t)kt * a_kstep;
                if (k + 4 <= g.K) {
                    a[q] = *reinterpret_cast<const float4*>(p);
                } else {
                    a[q].x = k + 0 < g.K ? p[0] : 0.0f; a[q].y = k + 1 < g.K ? p[1] : 0.0f;
                    a[q].z = k + 2 < g.K ? p[2] : 0.0f; a[q].w = k + 3 < g.K ? p[3] : 0.0f;
                }
            }
        };
        auto load_w = [&](int kt) {
#pragma unroll
            for (int j = 0; j < WLD; ++j) {
                const int p = pt + 256 * j;
                if (WPIECES % 256 == 0 || p < WPIECES) {
                    const int kb = 2 * kt + wrow[j];
                    w0[j] = kb < KB ? Wx[wsrc[j] + (size_t)kb * 6] : make_uint4(0, 0, 0, 0);
                }
            }
        };
        auto store = [&](int stage, const float4 (&a)[4]) {
            unsigned char* As = smem + stage * STAGE;
            unsigned char* Ws = As + XS_BM * XS_ROW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t hi[4], mid[4], lo[4];
                split3(a[q].x, hi[0], mid[0], lo[0]); split3(a[q].y, hi[1], mid[1], lo[1]);
                split3(a[q].z, hi[2], mid[2], lo[2]); split3(a[q].w, hi[3], mid[3], lo[3]);
                unsigned char* d = As + (lr + 32 * q) * XS_ROW + 8 * lq;
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]));
                *reinterpret_cast<uint2*>(d + 64) = make_uint2(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]));
                *reinterpret_cast<uint2*>(d + 128) = make_uint2(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]));
            }
#pragma unroll
            for (int j = 0; j < WLD; ++j) {
                const int p = pt + 256 * j;
                if (WPIECES % 256 == 0 || p < WPIECES) *reinterpret_cast<uint4*>(Ws + wdst[j]) = w0[j];
            }
        };
        if (nt > 0) { load_a(kt_begin, a0); load_w(kt_begin); }
        if (nt > 1) load_a(kt_begin + 1, a1);
        for (int j = 0; j < nt; j += 2) {
            // tile j from a0 -> stage 0; then tile j + 1 from a1 -> stage 1 (stage index = tile parity)
            store(0, a0);
            if (j + 1 < nt) load_w(kt_begin + j + 1);
            if (j + 2 < nt) load_a(kt_begin + j + 2, a0);
            __syncthreads();                                   // barrier j: stage 0 holds tile j; consumers are done with tile j - 1
            if (j + 1 < nt) {
                store(1, a1);
                if (j + 2 < nt) load_w(kt_begin + j + 2);
                if (j + 3 < nt) load_a(kt_begin + j + 3, a1);
                __syncthreads();                               // barrier j + 1
            }
        }
        return;                                                // consumers own the epilogue; no barrier follows the last one above
    }

    // ---------------------------------------------------------------- consumers
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    const int a_off = (wave * 32 + i) * XS_ROW + 16 * h, w_off = XS_BM * XS_ROW + i * XS_ROW + 16 * h;
    const bool has_rows = bm + wave * 32 < g.M;
    auto multiply = [&](const unsigned char* st) {
        bf16x8 af[3], wf[3], an[3], wn[3];
        auto read_a = [&](int kk, bf16x8 (&d)[3]) {
            const unsigned char* ap = st + a_off + 32 * kk;
            d[0] = *reinterpret_cast<const bf16x8*>(ap); d[1] = *reinterpret_cast<const bf16x8*>(ap + 64);
            d[2] = *reinterpret_cast<const bf16x8*>(ap + 128);
        };
        auto read_w = [&](int kk, int c, bf16x8 (&d)[3]) {
            const unsigned char* wp = st + w_off + c * 32 * XS_ROW + 32 * kk;
            d[0] = *reinterpret_cast<const bf16x8*>(wp); d[1] = *reinterpret_cast<const bf16x8*>(wp + 64);
            d[2] = *reinterpret_cast<const bf16x8*>(wp + 128);
        };
        read_a(0, af);
        read_w(0, 0, wf);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                // the next column block's fragments (at the end of a k-half: the next half's) are requested BEFORE this
                // block's six MFMAs and pinned there: the matrix pipe never waits for an LDS round trip
                if (c + 1 < CB) read_w(kk, c + 1, wn);
                else if (kk == 0) { read_a(1, an); read_w(1, 0, wn); }
                __builtin_amdgcn_sched_barrier(0);
                // terms: 0 = hi, 1 = mid, 2 = lo; small products first, hi*hi last (the order of gemm_x3_kernel)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], wf[1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wf[2], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], wf[0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wf[1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], wf[0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wf[0], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < CB || kk == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) wf[q] = wn[q];
                    if (c + 1 == CB) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) af[q] = an[q];
                    }
                }
            }
        }
    };
    for (int j = 0; j < nt; ++j) {
        __syncthreads();                                       // barrier j: stage j & 1 holds tile j
        if (has_rows) multiply(smem + (j & 1) * STAGE);
    }

    const int m0 = bm + wave * 32;
    if (m0 >= g.M) return;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int n = bn + c * 32 + i;
        if (n >= g.N) continue;
        if (g.splitk > 1) {
            float* part = g.splitk_ws + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) part[(size_t)m * g.N + n] = acc[c][r];
            }
            continue;
        }
        const float bias = g.bias ? g.bias[n] : 0.0f;
        const float al2 = g.alpha ? g.alpha[n] : 1.0f, be2 = g.alpha ? g.beta[n] : 0.0f;
        float* cp = g.C + (size_t)(m0 + 4 * h) * g.ldc + n;
        const float* rp = g.res ? g.res + (size_t)(m0 + 4 * h) * g.ldres + n : nullptr;
        const bool full = m0 + 32 <= g.M;
        auto finish = [&](float a) {
            float v = a + bias;
            v = v * al2 + be2;
            return xs_act<ACT>(v);
        };
        if (full && !rp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = finish(acc[c][r]);
        } else if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t ro = (size_t)((r & 3) + 8 * (r >> 2));
                cp[ro * g.ldc] = rp[ro * g.ldres] + g.rscale * finish(acc[c][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = (r & 3) + 8 * (r >> 2);
                if (m0 + 4 * h + mo < g.M) {
                    float v = finish(acc[c][r]);
                    if (rp) v = rp[(size_t)mo * g.ldres] + g.rscale * v;
                    cp[(size_t)mo * g.ldc] = v;
                }
            }
        }
    }
}
}  // namespace

// column blocks per workgroup (<= 4: 64 accumulator AGPRs): least padded width first, then the widest tile
static int xs_pick_cb(int N) {
    int best = 2;
    long best_cost = -1;
    for (int cb = 2; cb <= 4; ++cb) {
        const long tiles = (N + 32 * cb - 1) / (32 * cb);
        const long cost = tiles * 32 * cb * 8 - cb;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cb; }
    }
    return best;
}

hipError_t launch_gemm_x3s(const GemmArgs& g, hipStream_t s) {
    const int sk = (g.splitk > 1 && g.splitk_ws) ? g.splitk : 1;
    GemmArgs a = g;
    a.splitk = sk;
    const int cb = xs_pick_cb(g.N);
    const int bn = 32 * cb;
    dim3 grid((g.M + XS_BM - 1) / XS_BM, (g.N + bn - 1) / bn, sk);
    const size_t lds = (size_t)2 * (XS_BM + bn) * XS_ROW;
#define XS_GO(CBV, ACTV)                                                                                           \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(gemm_x3s_kernel<CBV, ACTV>), lds);              \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((gemm_x3s_kernel<CBV, ACTV>), grid, dim3(512), lds, s, a);                              \
    }
#define XS_ACT(CBV)                                                                                                \
    switch (sk > 1 ? (int)ACT_NONE : a.act) {                                                                      \
        case ACT_RELU: XS_GO(CBV, ACT_RELU) break;                                                                 \
        case ACT_GELU: XS_GO(CBV, ACT_GELU) break;                                                                 \
        case ACT_SILU: XS_GO(CBV, ACT_SILU) break;                                                                 \
        case ACT_SIGMOID: XS_GO(CBV, ACT_SIGMOID) break;                                                           \
        default: XS_GO(CBV, ACT_NONE) break;                                                                       \
    }
    switch (cb) {
        case 2: XS_ACT(2) break;
        case 3: XS_ACT(3) break;
        default: XS_ACT(4) break;
    }
#undef XS_ACT
#undef XS_GO
    return hipGetLastError();
}
