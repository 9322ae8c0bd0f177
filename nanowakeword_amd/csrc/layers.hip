// layers.hip - gfx950 kernels of the classifier heads (float32 end to end, MFMA f32 for dense contractions).
//
// Dense contractions (Linear layers, 1x1 convs, GRU input projections) run on v_mfma_f32_32x32x2_f32: exact float32
// products/accumulation (an fmaf chain per output), 157 TF peak; tiles are staged through LDS with coalesced 16-byte
// loads.  Lane l = (i = l&31, h = l>>5) feeds A[m0+i][k0+4h .. +3]; MFMA step s of a group of four consumes element s,
// i.e. k = k0 + 4h + s - the k <-> (step, half-wave) assignment is a bijection shared by both operands, which is all
// the contraction needs.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Long-K layers go to the
// split-operand kernel in gemm_x3.hip (bf16 matrix cores, float32-grade results).
//
// Convolutions are register-blocked direct convs on the VALU: one lane = one output position x COB
// output channels, weights wave-uniform (scalar loads), so the FMA:load ratio is 36:1 (pooled 3x3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "split_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_GELU: return nww_gelu(v);
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// compile-time variant: a run-time switch costs several scalar branches per element (instruction-fetch bubbles)
template <int ACT>
__device__ __forceinline__ float act_ct(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return v / (1.0f + expf(-v));
    if (ACT == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
#define NWW_DISPATCH_ACT(act, CALL)                 \
    switch (act) {                                  \
        case ACT_RELU: { CALL(ACT_RELU); break; }   \
        case ACT_GELU: { CALL(ACT_GELU); break; }   \
        case ACT_SILU: { CALL(ACT_SILU); break; }   \
        default: { CALL(ACT_NONE); break; }         \
    }

// ------------------------------------------------------------------------------------------ GEMM
// block = 256 threads = 4 waves as 2(M) x 2(N), each wave one 32x32 tile -> 64x64x32 tiles; coalesced 16-byte global
// loads (8 rows x 128 B per wave instruction - a first version that fetched one row per lane straight from L2 was bound
// by the texture-address unit: fc1 at 37% of peak), next K-tile prefetched into registers during the multiplication, operand
// fragments read back with conflict-free ds_read_b128 (row stride 36 floats).  ONE LDS stage (18 KB): the kernel is bound
// by per-workgroup load latency, and twice the resident workgroups hide it better than a second stage did (+2-4 %).
#define GT_LD 36
// ACT is a template parameter: the run-time switch costs several scalar branches per output element, and for short-K
// layers (Conformer: 4.5 K-tiles) the epilogue is a third of a workgroup's time.
// KIND 0: plain GEMM; 1: dual product (BcResNet block)
template <int ACT, int KIND = 0>
__global__ void __launch_bounds__(256) gemm_lds_kernel(GemmArgs g) {
    constexpr bool DUAL = KIND != 0;
    __shared__ __attribute__((aligned(16))) float As[1][64 * GT_LD];
    __shared__ __attribute__((aligned(16))) float Ws[1][64 * GT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = blockIdx.x * 64, bn = blockIdx.y * 64;
    const int i = lane & 31, h = lane >> 5;
    // loader mapping: thread -> (row r = tid/8 + 32*q, 16-byte column kq = tid%8)
    const int lr = tid >> 3, lq = tid & 7;
    const int arow_l = ((wave >> 1) * 32 + i) * GT_LD + 4 * h;
    const int wrow_l = ((wave & 1) * 32 + i) * GT_LD + 4 * h;
    // acc += A[bm.., k_begin..k_end) . W[bn.., k_begin..k_end)^T, K-tiles of 32 through the two LDS stages
    auto contract = [&](const float* A, int lda, const float* W, int K, int k_begin, int k_end, f32x16& acc) {
        const float* arow[2];
        const float* wrow[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            arow[q] = A + (size_t)min(bm + lr + 32 * q, g.M - 1) * lda + 4 * lq;
            wrow[q] = W + (size_t)min(bn + lr + 32 * q, g.N - 1) * K + 4 * lq;
        }
        auto gload = [&](int k0, float4 (&ra)[2], float4 (&rw)[2]) {
            const bool ok = k0 + 4 * lq + 4 <= k_end;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rw[q] = ok ? *reinterpret_cast<const float4*>(wrow[q] + k0) : make_float4(0, 0, 0, 0);
                ra[q] = ok ? *reinterpret_cast<const float4*>(arow[q] + k0) : make_float4(0, 0, 0, 0);
            }
        };
        auto lstore = [&](int buf, const float4 (&ra)[2], const float4 (&rw)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<float4*>(&As[buf][(lr + 32 * q) * GT_LD + 4 * lq]) = ra[q];
                *reinterpret_cast<float4*>(&Ws[buf][(lr + 32 * q) * GT_LD + 4 * lq]) = rw[q];
            }
        };
        float4 ra[2], rw[2];
        constexpr int cur = 0;
        if (k_begin < k_end) gload(k_begin, ra, rw);
        for (int k0 = k_begin; k0 < k_end; k0 += 32) {
            const bool more = k0 + 32 < k_end;
            __syncthreads();
            lstore(0, ra, rw);
            __syncthreads();
            if (more) gload(k0 + 32, ra, rw);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 a = *reinterpret_cast<const float4*>(&As[cur][arow_l + 8 * u]);
                const float4 b = *reinterpret_cast<const float4*>(&Ws[cur][wrow_l + 8 * u]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    };
    int k_begin = 0, k_end = g.K;
    if (g.splitk > 1) {
        const int kc = (((g.K + g.splitk - 1) / g.splitk) + 31) & ~31;
        k_begin = blockIdx.z * kc;
        k_end = min(g.K, k_begin + kc);
    }
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; if (DUAL) acc2[r] = 0.0f; }
    const bool dual = DUAL;
    contract(g.A, g.lda, g.W, g.K, k_begin, k_end, acc);
    if (DUAL) contract(g.A2, g.lda2, g.W2, g.K2, 0, g.K2, acc2);         // second product of a dual GEMM (never split)
    const int m0 = bm + (wave >> 1) * 32, n = bn + (wave & 1) * 32 + i;
    if (m0 >= g.M || n >= g.N) return;
    if (g.splitk > 1) {
        float* part = g.splitk_ws + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < g.M) part[(size_t)m * g.N + n] = acc[r];
        }
        return;
    }
    const float bias = g.bias ? g.bias[n] : 0.0f;
    const float al = g.alpha ? g.alpha[n] : 1.0f, be = g.alpha ? g.beta[n] : 0.0f;
    const float al2 = (dual && g.alpha2) ? g.alpha2[n] : 1.0f, be2 = (dual && g.alpha2) ? g.beta2[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) {
            float v = acc[r] + bias;
            if (g.alpha) v = v * al + be;
            v = act_ct<ACT>(v);
            if (g.res) v = g.res[(size_t)m * g.ldres + n] + g.rscale * v;
            if (dual) v = (acc2[r] * al2 + be2) + g.rscale * v;      // same association as res + rscale * v
            g.C[(size_t)m * g.ldc + n] = v;
        }
    }
}

__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmArgs g) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t MN = (size_t)g.M * g.N;
    if (idx >= MN) return;
    const int m = (int)(idx / g.N), n = (int)(idx - (size_t)m * g.N);
    float acc = g.splitk_ws[idx];
    for (int z = 1; z < g.splitk; ++z) acc += g.splitk_ws[(size_t)z * MN + idx];     // fixed order: deterministic
    float v = acc + (g.bias ? g.bias[n] : 0.0f);
    if (g.alpha) v = v * g.alpha[n] + g.beta[n];
    v = act_apply(v, g.act);
    if (g.res) v = g.res[(size_t)m * g.ldres + n] + g.rscale * v;
    g.C[(size_t)m * g.ldc + n] = v;
}

// The split depends on the layer shape (N, K) only - never on M - so a clip's logit is bit-identical whatever
// batch or shard it is computed in (the K-chunk boundaries and the z-order of the reduction are fixed).
int gemm_recommended_splitk(long long M, int N, int K, int cu_count) {
    (void)M; (void)cu_count;
    if (K < 2048 || N > 256) return 1;
    // K >= 4096 (fc1-like): chunks of ~3072; 2048 <= K < 4096 (DNN layer1 on (98,40): K = 3920, two column tiles
    // only): chunks of ~1024 so that small batches still spread over a few dozen CUs
    int s = K / (K >= 4096 ? 3072 : 1024);
    return s < 1 ? 1 : (s > 16 ? 16 : s);
}

// generic (unaligned / K % 4 != 0) fallback on the VALU: one thread per output, same epilogue.
__global__ void __launch_bounds__(256) gemm_valu_kernel(GemmArgs g) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)g.M * g.N) return;
    const int m = (int)(idx / g.N), n = (int)(idx - (size_t)m * g.N);
    const float* a = g.A + (size_t)m * g.lda;
    const float* w = g.W + (size_t)n * g.K;
    float acc = 0.0f;
    for (int k = 0; k < g.K; ++k) acc = fmaf(a[k], w[k], acc);
    float v = acc + (g.bias ? g.bias[n] : 0.0f);
    if (g.alpha) v = v * g.alpha[n] + g.beta[n];
    v = act_apply(v, g.act);
    if (g.res) v = g.res[(size_t)m * g.ldres + n] + g.rscale * v;
    g.C[(size_t)m * g.ldc + n] = v;
}

// true when launch_gemm will run one of the MFMA kernels (which write split-K partials for splitk > 1) and not the VALU fallback
bool gemm_writes_partials(const GemmArgs& g) {
    const bool aligned = (g.K % 4 == 0) && (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.W) & 15) == 0);
    return aligned || (!g.A2 && gemm_x3_usable(g));
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const bool aligned = (g.K % 4 == 0) && (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.W) & 15) == 0);
    if (g.A2 && !(aligned && g.K2 % 4 == 0 && g.lda2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(g.A2) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(g.W2) & 15) == 0) && g.splitk <= 1))
        return hipErrorInvalidValue;                           // the dual form exists on the LDS kernel only
    if (!g.A2 && gemm_x3_usable(g)) {
        hipError_t e = launch_gemm_x3(g, s);
        if (e != hipSuccess) return e;
        if (g.splitk > 1 && g.splitk_ws && !g.defer_reduce) {
            const size_t total = (size_t)g.M * g.N;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g);
        }
        return hipGetLastError();
    }
    if (aligned) {
        const int sk = (g.splitk > 1 && g.splitk_ws) ? g.splitk : 1;
        GemmArgs a = g;
        a.splitk = sk;
        dim3 grid((g.M + 63) / 64, (g.N + 63) / 64, sk);
#define GL_CALL(A)                                                                                         \
    if (a.A2) hipLaunchKernelGGL((gemm_lds_kernel<A, 1>), grid, dim3(256), 0, s, a);                       \
    else hipLaunchKernelGGL((gemm_lds_kernel<A, 0>), grid, dim3(256), 0, s, a);
        switch (a.act) {
            case ACT_RELU: GL_CALL(ACT_RELU) break;
            case ACT_GELU: GL_CALL(ACT_GELU) break;
            case ACT_SILU: GL_CALL(ACT_SILU) break;
            case ACT_SIGMOID: GL_CALL(ACT_SIGMOID) break;
            default: GL_CALL(ACT_NONE) break;
        }
#undef GL_CALL
        if (sk > 1 && !g.defer_reduce) {
            const size_t total = (size_t)g.M * g.N;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        }
    } else {
        if (g.defer_reduce) return hipErrorInvalidValue;       // this path writes C itself, not split-K partials: the consumer's reduce would read a stale workspace
        const size_t total = (size_t)g.M * g.N;
        hipLaunchKernelGGL(gemm_valu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ conv 3x3
template <int COB, bool POOL, int ACT>
__global__ void __launch_bounds__(256) conv3x3_kernel(Conv3Args a) {
    constexpr int PS = POOL ? 4 : 3;       // input patch edge
    constexpr int NP = POOL ? 4 : 1;       // conv outputs per lane (2x2 quad when pooling)
    const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < Ho * Wo;
    const int pc = valid ? p : 0;
    const int oy = pc / Wo, ox = pc - oy * Wo;
    const int co0 = blockIdx.y * COB, b = blockIdx.z;
    const int y0 = (POOL ? 2 * oy : oy) - 1, x0 = (POOL ? 2 * ox : ox) - 1;
    int off[PS][PS];
    unsigned okmask = 0;
#pragma unroll
    for (int r = 0; r < PS; ++r)
#pragma unroll
        for (int c = 0; c < PS; ++c) {
            const int yy = y0 + r, xx = x0 + c;
            const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            off[r][c] = ok ? yy * a.W + xx : 0;
            okmask |= (ok ? 1u : 0u) << (r * PS + c);
        }
    float acc[NP][COB];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int co = 0; co < COB; ++co) acc[q][co] = 0.0f;
    const size_t plane = (size_t)a.H * a.W;
    const float* inb = a.in + (size_t)b * a.Cin * plane;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* inc = inb + (size_t)ci * plane;
        float patch[PS][PS];
#pragma unroll
        for (int r = 0; r < PS; ++r)
#pragma unroll
            for (int c = 0; c < PS; ++c) {
                const float v = inc[off[r][c]];
                patch[r][c] = ((okmask >> (r * PS + c)) & 1u) ? v : 0.0f;
            }
        const float* wci = a.w + ((size_t)co0 * a.Cin + ci) * 9;      // wave-uniform -> scalar loads
#pragma unroll
        for (int co = 0; co < COB; ++co) {
            const float* wc = wci + (size_t)co * a.Cin * 9;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = wc[dy * 3 + dx];
                    if (POOL) {
                        acc[0][co] = fmaf(patch[dy][dx], wv, acc[0][co]);
                        acc[1][co] = fmaf(patch[dy][dx + 1], wv, acc[1][co]);
                        acc[2][co] = fmaf(patch[dy + 1][dx], wv, acc[2][co]);
                        acc[3][co] = fmaf(patch[dy + 1][dx + 1], wv, acc[3][co]);
                    } else {
                        acc[0][co] = fmaf(patch[dy][dx], wv, acc[0][co]);
                    }
                }
        }
    }
    if (!valid) return;
    const size_t ostride = a.nhwc_out ? 1 : (size_t)Ho * Wo;          // channel stride of the output
    float* ob = a.nhwc_out ? a.out + ((size_t)b * Ho * Wo + p) * a.Cout + co0
                           : a.out + ((size_t)b * a.Cout + co0) * Ho * Wo + p;
#pragma unroll
    for (int co = 0; co < COB; ++co) {
        const float bias = a.bias ? a.bias[co0 + co] : 0.0f;
        const float al = a.alpha ? a.alpha[co0 + co] : 1.0f, be = a.alpha ? a.beta[co0 + co] : 0.0f;
        float best = -INFINITY;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float v = acc[q][co] + bias;
            if (a.alpha) v = v * al + be;
            v = act_ct<ACT>(v);
            best = (q == 0) ? v : fmaxf(best, v);
        }
        ob[(size_t)co * ostride] = best;
    }
}

// Cin = 1, 3x3, pad 1, + folded BN/bias + act + MaxPool2, channels-last output [B][Ho][Wo][Cout].
// lane = (position, channel group of 8): the Cout/8 lanes of a position are adjacent, so a position's Cout floats
// leave as one contiguous run (the generic kernel wrote 32-byte pieces at a Cout*4-byte stride: 3.0 ms -> see
// DESIGN.md).  A thread keeps its 72 weights in registers and walks positions grid-stride.
template <int ACT>
__global__ void __launch_bounds__(256) conv3x3_c1_pool_nhwc_kernel(Conv3Args a) {
    const int Ho = a.H / 2, Wo = a.W / 2, G = a.Cout / 8;
    const int cg = threadIdx.x % G;
    const int co0 = cg * 8;
    float w[8][9], bias[8], al[8], be[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w[co][t] = a.w[(size_t)(co0 + co) * 9 + t];
        bias[co] = a.bias ? a.bias[co0 + co] : 0.0f;
        al[co] = a.alpha ? a.alpha[co0 + co] : 1.0f;
        be[co] = a.alpha ? a.beta[co0 + co] : 0.0f;
    }
    const size_t npos = (size_t)a.B * Ho * Wo;
    const size_t ppb = 256 / G;                                  // positions per block iteration
    for (size_t pos = (size_t)blockIdx.x * ppb + threadIdx.x / G; pos < npos; pos += (size_t)gridDim.x * ppb) {
        const size_t b = pos / ((size_t)Ho * Wo);
        const int p = (int)(pos - b * Ho * Wo);
        const int oy = p / Wo, ox = p - oy * Wo;
        const float* in = a.in + b * (size_t)a.H * a.W;
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = 2 * oy - 1 + r, xx = 2 * ox - 1 + c;
                const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                patch[r][c] = ok ? in[yy * a.W + xx] : 0.0f;
            }
        float outv[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = w[co][dy * 3 + dx];
                    s00 = fmaf(patch[dy][dx], wv, s00);
                    s01 = fmaf(patch[dy][dx + 1], wv, s01);
                    s10 = fmaf(patch[dy + 1][dx], wv, s10);
                    s11 = fmaf(patch[dy + 1][dx + 1], wv, s11);
                }
            float v0 = s00 + bias[co], v1 = s01 + bias[co], v2 = s10 + bias[co], v3 = s11 + bias[co];
            if (a.alpha) { v0 = v0 * al[co] + be[co]; v1 = v1 * al[co] + be[co]; v2 = v2 * al[co] + be[co]; v3 = v3 * al[co] + be[co]; }
            outv[co] = fmaxf(fmaxf(act_ct<ACT>(v0), act_ct<ACT>(v1)), fmaxf(act_ct<ACT>(v2), act_ct<ACT>(v3)));
        }
        float4* dst = reinterpret_cast<float4*>(a.out + pos * a.Cout + co0);
        dst[0] = make_float4(outv[0], outv[1], outv[2], outv[3]);
        dst[1] = make_float4(outv[4], outv[5], outv[6], outv[7]);
    }
}

template <int COB>
static hipError_t launch_conv3x3_cob(const Conv3Args& a, hipStream_t s) {
    const int Ho = a.pool ? a.H / 2 : a.H, Wo = a.pool ? a.W / 2 : a.W;
    if (Ho <= 0 || Wo <= 0) return hipErrorInvalidValue;
    dim3 grid((Ho * Wo + 255) / 256, a.Cout / COB, a.B);
#define CONV_CALL(A)                                                                              \
    if (a.pool) hipLaunchKernelGGL((conv3x3_kernel<COB, true, A>), grid, dim3(256), 0, s, a);     \
    else hipLaunchKernelGGL((conv3x3_kernel<COB, false, A>), grid, dim3(256), 0, s, a)
    NWW_DISPATCH_ACT(a.act, CONV_CALL)
#undef CONV_CALL
    return hipGetLastError();
}

hipError_t launch_conv3x3(const Conv3Args& a, hipStream_t s) {
    if (a.nhwc_out && a.pool && a.Cin == 1 && a.Cout % 8 == 0 && 256 % (a.Cout / 8) == 0 && a.H >= 2 && a.W >= 2) {
        const size_t npos = (size_t)a.B * (a.H / 2) * (a.W / 2);
        const size_t ppb = 256 / (a.Cout / 8);
        size_t grid = (npos + ppb - 1) / ppb;
        if (grid > 256 * 16) grid = 256 * 16;
#define C1_CALL(A) hipLaunchKernelGGL((conv3x3_c1_pool_nhwc_kernel<A>), dim3((unsigned)grid), dim3(256), 0, s, a)
        NWW_DISPATCH_ACT(a.act, C1_CALL)
#undef C1_CALL
        return hipGetLastError();
    }
    // pooled convs: 8 output channels per lane keep the 72 per-channel weights in SGPRs (16 spill them)
    const int want = a.pool ? 8 : 16;
    if (want >= 16 && a.Cout % 16 == 0) return launch_conv3x3_cob<16>(a, s);
    if (a.Cout % 8 == 0) return launch_conv3x3_cob<8>(a, s);
    if (a.Cout % 4 == 0) return launch_conv3x3_cob<4>(a, s);
    return launch_conv3x3_cob<1>(a, s);
}


__global__ void __launch_bounds__(256)
dwconv3x3_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ d_out,
                      float* __restrict__ xs_out, int C, int H, int W, int Ho, int Wo, int sh, int sw, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // ((b*Ho + oy)*Wo + ox)*C + c : lanes = channels
    if (idx >= total) return;
    const int c = (int)(idx % C);
    size_t t = idx / C;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const size_t b = t / Ho;
    const float* ip = in + b * (size_t)H * W * C + c;
    float acc = 0.0f, centre = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = oy * sh - 1 + dy;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = ox * sw - 1 + dx;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float v = ok ? ip[((size_t)yy * W + xx) * C] : 0.0f;
            if (dy == 1 && dx == 1) centre = v;                  // in[b][oy*sh][ox*sw][c]: always in range
            acc = fmaf(v, wt[(dy * 3 + dx) * C + c], acc);
        }
    }
    d_out[idx] = acc;
    if (xs_out) xs_out[idx] = centre;
}

// Four outputs along x and four channels per thread: the 3 x ((4 - 1) sw + 3) input columns are loaded once as 16-byte pieces
// (stride 1: 18 loads for 4 outputs instead of 36 four-byte ones per channel; stride 2: 27), the nine weights of the four
// channels stay in registers.  Same tap order and fmaf chain as dwconv3x3_nhwc_kernel.  xs_out is not supported (the
// split-operand block kernel gathers the shortcut rows itself).
template <int SW>
__global__ void __launch_bounds__(256)
dwconv3x3_nhwc_x4_kernel(const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ d_out, int C, int H, int W,
                         int Ho, int Wo, int sh, size_t total) {
    constexpr int NX = 4, COLS = (NX - 1) * SW + 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // ((b*Ho + oy)*Wg + xg)*C4 + c4
    if (idx >= total) return;
    const int C4 = C / 4, Wg = (Wo + NX - 1) / NX;
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int xg = (int)(t % Wg);
    t /= Wg;
    const int oy = (int)(t % Ho);
    const size_t b = t / Ho;
    const float* ip = in + b * (size_t)H * W * C + 4 * c4;
    float4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(wt + (size_t)k * C + 4 * c4);
    float4 acc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int x_first = xg * NX * SW - 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = oy * sh - 1 + dy;
        const bool oky = yy >= 0 && yy < H;
        float4 v[COLS];
#pragma unroll
        for (int cx = 0; cx < COLS; ++cx) {
            const int xx = x_first + cx;
            v[cx] = (oky && xx >= 0 && xx < W) ? *reinterpret_cast<const float4*>(ip + ((size_t)yy * W + xx) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const float4 p = v[j * SW + dx], q = w[dy * 3 + dx];
                acc[j].x = fmaf(p.x, q.x, acc[j].x); acc[j].y = fmaf(p.y, q.y, acc[j].y);
                acc[j].z = fmaf(p.z, q.z, acc[j].z); acc[j].w = fmaf(p.w, q.w, acc[j].w);
            }
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int ox = xg * NX + j;
        if (ox < Wo) *reinterpret_cast<float4*>(d_out + ((b * Ho + oy) * (size_t)Wo + ox) * C + 4 * c4) = acc[j];
    }
}

// 16-bit activations (nww_config.act_dtype; KIND = ACT16_BF16 / ACT16_F16 of split_h2.h): a thread owns EIGHT channels (one
// 16-byte load per input pixel) of two outputs along x - 7.5 sixteen-byte loads per output where the float32 kernel above needs
// 13.5 per eight channels.  binary16: the input carries its tensor's scale, the sums are multiplied by out_mul = output scale /
// input scale (a power of two) before they are rounded.
template <int SW, int KIND>
__global__ void __launch_bounds__(256)
dwconv3x3_nhwc_bf16_kernel(const uint16_t* __restrict__ in, const float* __restrict__ wt, uint16_t* __restrict__ d_out, int C, int H, int W,
                           int Ho, int Wo, int sh, size_t total, float out_mul) {
    constexpr int NX = 2, COLS = (NX - 1) * SW + 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // ((b*Ho + oy)*Wg + xg)*C8 + c8
    if (idx >= total) return;
    const int C8 = C / 8, Wg = (Wo + NX - 1) / NX;
    const int c8 = (int)(idx % C8);
    size_t t = idx / C8;
    const int xg = (int)(t % Wg);
    t /= Wg;
    const int oy = (int)(t % Ho);
    const size_t b = t / Ho;
    const uint16_t* ip = in + b * (size_t)H * W * C + 8 * c8;
    float acc[NX][8];
#pragma unroll
    for (int j = 0; j < NX; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.0f;
    const int x_first = xg * NX * SW - 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = oy * sh - 1 + dy;
        const bool oky = yy >= 0 && yy < H;
        uint4 v[COLS];
#pragma unroll
        for (int cx = 0; cx < COLS; ++cx) {
            const int xx = x_first + cx;
            v[cx] = (oky && xx >= 0 && xx < W) ? *reinterpret_cast<const uint4*>(ip + ((size_t)yy * W + xx) * C) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float4 w0 = *reinterpret_cast<const float4*>(wt + (size_t)(dy * 3 + dx) * C + 8 * c8);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + (size_t)(dy * 3 + dx) * C + 8 * c8 + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const uint4 p = v[j * SW + dx];
                const uint32_t u[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo, hi;
                    nww_unpk_act16(KIND, u[e], lo, hi);
                    acc[j][2 * e] = fmaf(lo, w[2 * e], acc[j][2 * e]);
                    acc[j][2 * e + 1] = fmaf(hi, w[2 * e + 1], acc[j][2 * e + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int ox = xg * NX + j;
        if (ox < Wo)
            *reinterpret_cast<uint4*>(d_out + ((b * Ho + oy) * (size_t)Wo + ox) * C + 8 * c8) =
                make_uint4(nww_pk_act16(KIND, acc[j][0], acc[j][1], out_mul), nww_pk_act16(KIND, acc[j][2], acc[j][3], out_mul),
                           nww_pk_act16(KIND, acc[j][4], acc[j][5], out_mul), nww_pk_act16(KIND, acc[j][6], acc[j][7], out_mul));
    }
}

hipError_t launch_dwconv3x3_nhwc(const float* in, const float* wt, float* d_out, float* xs_out, int B, int C, int H,
                                 int W, int sh, int sw, hipStream_t s, int act16, float out_mul) {
    const int Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;
    if (act16) {                                               // 16-bit activations: their own kernel, eight channels per thread
        if (xs_out || C % 8 != 0 || (sw != 1 && sw != 2) || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(wt)) & 15) != 0)
            return hipErrorInvalidValue;
        const size_t total8 = (size_t)B * Ho * ((Wo + 1) / 2) * (C / 8);
        const uint16_t* ib = reinterpret_cast<const uint16_t*>(in);
        uint16_t* ob = reinterpret_cast<uint16_t*>(d_out);
        const dim3 grid((unsigned)((total8 + 255) / 256));
        if (act16 == ACT16_F16) {
            if (sw == 1) hipLaunchKernelGGL((dwconv3x3_nhwc_bf16_kernel<1, ACT16_F16>), grid, dim3(256), 0, s, ib, wt, ob, C, H, W, Ho, Wo, sh, total8, out_mul);
            else hipLaunchKernelGGL((dwconv3x3_nhwc_bf16_kernel<2, ACT16_F16>), grid, dim3(256), 0, s, ib, wt, ob, C, H, W, Ho, Wo, sh, total8, out_mul);
        } else {
            if (sw == 1) hipLaunchKernelGGL((dwconv3x3_nhwc_bf16_kernel<1, ACT16_BF16>), grid, dim3(256), 0, s, ib, wt, ob, C, H, W, Ho, Wo, sh, total8, 1.0f);
            else hipLaunchKernelGGL((dwconv3x3_nhwc_bf16_kernel<2, ACT16_BF16>), grid, dim3(256), 0, s, ib, wt, ob, C, H, W, Ho, Wo, sh, total8, 1.0f);
        }
        return hipGetLastError();
    }
    if (!xs_out && C % 4 == 0 && (sw == 1 || sw == 2) &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0) {
        const size_t total4 = (size_t)B * Ho * ((Wo + 3) / 4) * (C / 4);
        if (sw == 1) hipLaunchKernelGGL(dwconv3x3_nhwc_x4_kernel<1>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, wt, d_out, C, H, W, Ho, Wo, sh, total4);
        else hipLaunchKernelGGL(dwconv3x3_nhwc_x4_kernel<2>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, wt, d_out, C, H, W, Ho, Wo, sh, total4);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(dwconv3x3_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, wt, d_out,
                       xs_out, C, H, W, Ho, Wo, sh, sw, total);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------ LayerNorm rows
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                 const float* __restrict__ b, int R, int D, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + (size_t)row * D;
    float* yr = y + (size_t)row * D;
    if (D <= 256) {                       // the row lives in four registers per lane: one HBM read instead of three passes
        float v[4];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = lane + 64 * j; v[j] = i < D ? xr[i] : 0.0f; s += v[j]; }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; if (lane + 64 * j < D) q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            if (i < D) yr[i] = act_apply((v[j] - mu) * rstd * w[i] + b[i], act);
        }
        return;
    }
    float s = 0.0f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mu = wave_sum(s) / (float)D;
    float q = 0.0f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mu; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    for (int i = lane; i < D; i += 64) yr[i] = act_apply((xr[i] - mu) * rstd * w[i] + b[i], act);
}

hipError_t launch_layernorm(const float* x, float* y, const float* w, const float* b, int R, int D, int act,
                            hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(layernorm_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, y, w, b, R, D, act);
    return hipGetLastError();
}

// The same LayerNorm on the split-K partials of the Linear that feeds it: x[row][i] = sum_z parts[z][row][i] + in_bias[i], z
// ascending - gemm_splitk_reduce_kernel's sum, bit for bit, without its launch and its round trip (DNN layer1; D <= 256).
__global__ void __launch_bounds__(256)
layernorm_parts_kernel(const float* __restrict__ parts, int nparts, size_t part_stride, const float* __restrict__ in_bias,
                       float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ b, int R, int D, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        float acc = 0.0f;
        if (i < D) {
            const float* p = parts + (size_t)row * D + i;
            acc = p[0];
            for (int z = 1; z < nparts; ++z) acc += p[(size_t)z * part_stride];
            acc += in_bias ? in_bias[i] : 0.0f;
        }
        v[j] = acc;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j];
    const float mu = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; if (lane + 64 * j < D) q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    float* yr = y + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i < D) yr[i] = act_apply((v[j] - mu) * rstd * w[i] + b[i], act);
    }
}
hipError_t launch_layernorm_parts(const float* parts, int nparts, size_t part_stride, const float* in_bias, float* y, const float* w,
                                  const float* b, int R, int D, int act, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if (D > 256 || nparts < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_parts_kernel, dim3((R + 3) / 4), dim3(256), 0, s, parts, nparts, part_stride, in_bias, y, w, b, R, D, act);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ transposes (E2E head on the transposed plane)
// 3x3 filters with their taps transposed: out[f][kx][ky] = w[f][ky][kx] (a convolution of the transposed plane with these
// gives the transposed result of the original convolution)
__global__ void __launch_bounds__(256) transpose3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int n9) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n9) return;
    const int f = idx / 9, t = idx - f * 9, ky = t / 3, kx = t - ky * 3;
    out[f * 9 + kx * 3 + ky] = w[idx];
}
hipError_t launch_transpose3x3(const float* w, float* out, int nfilters, hipStream_t s) {
    const int n9 = nfilters * 9;
    hipLaunchKernelGGL(transpose3x3_kernel, dim3((n9 + 255) / 256), dim3(256), 0, s, w, out, n9);
    return hipGetLastError();
}
// [B][R][C] -> [B][C][R] through a 32 x 33 LDS tile
__global__ void __launch_bounds__(256) transpose_planes_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int ct = (C + 31) / 32;                             // (clip on grid.x: no 65535 limit on the batch)
    const int b = blockIdx.x, r0 = (blockIdx.y / ct) * 32, c0 = (blockIdx.y % ct) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* ib = in + (size_t)b * R * C;
    float* ob = out + (size_t)b * R * C;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = ib[(size_t)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) ob[(size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}
hipError_t launch_transpose_planes(const float* in, float* out, int B, int R, int C, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_planes_kernel, dim3(B, ((C + 31) / 32) * ((R + 31) / 32)), dim3(256), 0, s, in, out, R, C);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ reductions / pools
__global__ void __launch_bounds__(256)
mean_mid_kernel(const float* __restrict__ in, float* __restrict__ out, int L, int D, size_t total, int bf16, float unscale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // idx = b*D + d
    if (idx >= total) return;
    const size_t b = idx / D, d = idx - b * D;
    float s = 0.0f;
    if (bf16) {                                                     // bf16 activations in, float32 mean out
        const uint16_t* p = reinterpret_cast<const uint16_t*>(in) + b * (size_t)L * D + d;
        if (bf16 == ACT16_F16) {
            for (int l = 0; l < L; ++l) s += (float)__builtin_bit_cast(_Float16, p[(size_t)l * D]);
            s *= unscale;
        } else {
            for (int l = 0; l < L; ++l) s += __uint_as_float((uint32_t)p[(size_t)l * D] << 16);
        }
    } else {
        const float* p = in + b * (size_t)L * D + d;
        for (int l = 0; l < L; ++l) s += p[(size_t)l * D];
    }
    out[idx] = s / (float)L;
}
// LayerNorm of every row of a clip followed by the mean over its rows, in one pass over [B][L][D] (the Conformer's last
// block.layer_norm + x.mean(dim=1), architectures.py:536-541): one workgroup per clip, a wave normalises rows w, w + 4, ...
// with the row in registers (layernorm_kernel's arithmetic) and keeps per-lane sums; the four waves' sums are added in wave
// order.  D <= 256.
__global__ void __launch_bounds__(256)
ln_mean_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ w, const float* __restrict__ bvec,
               int L, int D) {
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (size_t)blockIdx.x * L * D;
    float wv[4], bv[4], acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        wv[j] = i < D ? w[i] : 0.0f; bv[j] = i < D ? bvec[i] : 0.0f; acc[j] = 0.0f;
    }
    for (int row = wave; row < L; row += 4) {
        const float* xr = xb + (size_t)row * D;
        float v[4];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = lane + 64 * j; v[j] = i < D ? xr[i] : 0.0f; s += v[j]; }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; if (lane + 64 * j < D) q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += (v[j] - mu) * rstd * wv[j] + bv[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][lane + 64 * j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) out[(size_t)blockIdx.x * D + i] = (((part[0][i] + part[1][i]) + part[2][i]) + part[3][i]) / (float)L;
}
hipError_t launch_ln_mean(const float* x, float* out, const float* w, const float* b, int B, int L, int D, hipStream_t s) {
    if (D > 256 || B <= 0) return D > 256 ? hipErrorInvalidValue : hipSuccess;
    hipLaunchKernelGGL(ln_mean_kernel, dim3(B), dim3(256), 0, s, x, out, w, b, L, D);
    return hipGetLastError();
}

// 16-bit rows (KIND = ACT16_BF16 / ACT16_F16 of split_h2.h; binary16: times unscale = 1 / the tensor's scale): a thread owns eight
// channels (16-byte loads)
template <int KIND>
__global__ void __launch_bounds__(256)
mean_mid_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int L, int D, size_t total8, float unscale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // idx = b*(D/8) + d8
    if (idx >= total8) return;
    const int D8 = D / 8;
    const size_t b = idx / D8, d8 = idx - b * D8;
    const uint16_t* p = in + b * (size_t)L * D + 8 * d8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_row = [&](const uint4 v) {
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo, hi;
            nww_unpk_act16(KIND, u[e], lo, hi);
            s[2 * e] += lo; s[2 * e + 1] += hi;
        }
    };
    int l = 0;
    for (; l + 8 <= L; l += 8) {                                    // eight rows in flight, added in ascending l
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const uint4*>(p + (size_t)(l + q) * D);
#pragma unroll
        for (int q = 0; q < 8; ++q) add_row(v[q]);
    }
    for (; l < L; ++l) add_row(*reinterpret_cast<const uint4*>(p + (size_t)l * D));
    float* o = out + b * D + 8 * d8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (KIND == ACT16_F16 ? s[e] * unscale : s[e]) / (float)L;
}
// float32 rows: a thread owns four channels (16-byte loads, eight rows in flight); per channel the same ascending-l sum
__global__ void __launch_bounds__(256)
mean_mid_f32x4_kernel(const float* __restrict__ in, float* __restrict__ out, int L, int D, size_t total4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // idx = b*(D/4) + d4
    if (idx >= total4) return;
    const int D4 = D / 4;
    const size_t b = idx / D4, d4 = idx - b * D4;
    const float* p = in + b * (size_t)L * D + 4 * d4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int l = 0;
    for (; l + 8 <= L; l += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(l + u) * D);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; l < L; ++l) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)l * D);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float fl = (float)L;
    *reinterpret_cast<float4*>(out + b * D + 4 * d4) = make_float4(s.x / fl, s.y / fl, s.z / fl, s.w / fl);
}
hipError_t launch_mean_mid(const float* in, float* out, int B, int L, int D, hipStream_t s, int bf16, float unscale) {
    if (bf16 && D % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const size_t total8 = (size_t)B * (D / 8);
        const dim3 grid((unsigned)((total8 + 255) / 256));
        if (bf16 == ACT16_F16) hipLaunchKernelGGL(mean_mid_bf16_kernel<ACT16_F16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), out, L, D, total8, unscale);
        else hipLaunchKernelGGL(mean_mid_bf16_kernel<ACT16_BF16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), out, L, D, total8, 1.0f);
        return hipGetLastError();
    }
    if (!bf16 && D % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const size_t total4 = (size_t)B * (D / 4);
        hipLaunchKernelGGL(mean_mid_f32x4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, out, L, D, total4);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * D;
    hipLaunchKernelGGL(mean_mid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, L, D, total, bf16, unscale);
    return hipGetLastError();
}


__global__ void __launch_bounds__(256)
avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int kh, int kw, int sh, int sw,
               int oh, int ow, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % ow);
    size_t t = idx / ow;
    const int i = (int)(t % oh);
    t /= oh;
    const float* p = in + t * (size_t)H * W + (size_t)(i * sh) * W + j * sw;
    float s = 0.0f;
    for (int y = 0; y < kh; ++y)
        for (int x = 0; x < kw; ++x) s += p[y * W + x];
    out[idx] = s / (float)(kh * kw);
}
// one wave per [H][W] plane, coalesced reads, up to 8 (possibly overlapping) windows accumulated per lane and
// reduced across the wave (the one-thread-per-output kernel above strides through the plane: 0.32 ms for the E2E head)
__global__ void __launch_bounds__(256)
avgpool_plane_kernel(const float* __restrict__ in, float* __restrict__ out, int BC, int H, int W, int kh, int kw, int sh,
                     int sw, int oh, int ow) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= BC) return;
    const float* p = in + (size_t)plane * H * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const int nwin = oh * ow;
    for (int idx = lane; idx < H * W; idx += 64) {
        const int y = idx / W, x = idx - y * W;
        const float v = p[idx];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nwin) {
                const int wy = (j / ow) * sh, wx = (j % ow) * sw;
                if (y >= wy && y < wy + kh && x >= wx && x < wx + kw) acc[j] += v;
            }
        }
    }
    const float inv = 1.0f / (float)(kh * kw);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < nwin) {
            const float s = wave_sum(acc[j]);
            if (lane == 0) out[(size_t)plane * nwin + j] = s * inv;
        }
    }
}

hipError_t launch_avgpool(const float* in, float* out, int BC, int H, int W, int kh, int kw, int sh, int sw, int oh,
                          int ow, hipStream_t s) {
    if (oh * ow <= 8) {
        hipLaunchKernelGGL(avgpool_plane_kernel, dim3((BC + 3) / 4), dim3(256), 0, s, in, out, BC, H, W, kh, kw, sh, sw, oh, ow);
        return hipGetLastError();
    }
    const size_t total = (size_t)BC * oh * ow;
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, H, W, kh, kw,
                       sh, sw, oh, ow, total);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) unary_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) y[idx] = act_apply(x[idx], act);
}
hipError_t launch_unary(const float* x, float* y, size_t n, int act, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(unary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n, act);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) glu_kernel(const float* __restrict__ in, float* __restrict__ out, int D, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t r = idx / D, d = idx - r * D;
    const float a = in[r * 2 * D + d], g = in[r * 2 * D + D + d];
    out[idx] = a * (1.0f / (1.0f + expf(-g)));
}
hipError_t launch_glu(const float* in, float* out, int R, int D, hipStream_t s) {
    const size_t total = (size_t)R * D;
    hipLaunchKernelGGL(glu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, D, total);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
dwconv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ y, int T, int D,
                int K, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // idx = (b*T + t)*D + d
    if (idx >= total) return;
    const int d = (int)(idx % D);
    const size_t bt = idx / D;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    const int left = K / 2;                                          // padding='same', odd K: K/2 each side
    const float* xb = x + b * (size_t)T * D + d;
    const float* wd = w + (size_t)d * K;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int tt = t - left + k;
        if (tt >= 0 && tt < T) acc = fmaf(xb[(size_t)tt * D], wd[k], acc);
    }
    float v = (acc + bias[d]) * alpha[d] + beta[d];
    y[idx] = v / (1.0f + expf(-v));
}
// 1 / (1 + e^-v) on the hardware exp2 and reciprocal (1 ulp each; relative error <= 3e-7, gemm_x3.hip's x3_sigmoid)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
// Register-blocked version for the Conformer's kernel_size 31: a lane owns channel d of TB consecutive frames, loads the
// TB + K - 1 inputs it needs once (coalesced along d) and keeps the K weights in registers: 46 loads per 496 FMAs (TB = 16) instead
// of one load per FMA.  Same fmaf order per output as the generic kernel (a zero-padded tap adds exactly nothing).
template <int K, int TB>
__global__ void __launch_bounds__(256)
dwconv1d_blocked_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                        const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ y, int T,
                        int D, int nTB, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // idx = (b*nTB + tb)*D + d
    if (idx >= total) return;
    const int d = (int)(idx % D);
    const size_t r = idx / D;
    const int tb = (int)(r % nTB);
    const size_t b = r / nTB;
    const int t0 = tb * TB, left = K / 2;
    const float* xb = x + b * (size_t)T * D + d;
    float wv[K], xin[TB + K - 1];
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = w[(size_t)d * K + k];
#pragma unroll
    for (int j = 0; j < TB + K - 1; ++j) {
        const int tt = t0 - left + j;
        xin[j] = (tt >= 0 && tt < T) ? xb[(size_t)tt * D] : 0.0f;
    }
    const float bs = bias[d], al = alpha[d], be = beta[d];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(xin[j + k], wv[k], acc);
        const float v = (acc + bs) * al + be;
        if (t0 + j < T) y[(b * T + t0 + j) * (size_t)D + d] = v * fast_sigmoid(v);      // swish: 4 instructions instead of expf + IEEE division (~35)
    }
}

hipError_t launch_dwconv1d_bn_swish(const float* x, const float* w, const float* bias, const float* alpha,
                                    const float* beta, float* y, int B, int T, int D, int K, hipStream_t s) {
    if (K == 31) {
        constexpr int TB = 26;      // T = 101 = 4 x 26 - 3: 97 % of the slots are frames (TB = 16: 90 %), 56 loads per 26 outputs (46 per 16)
        const int nTB = (T + TB - 1) / TB;
        const size_t lanes = (size_t)B * nTB * D;
        hipLaunchKernelGGL((dwconv1d_blocked_kernel<31, TB>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, x, w,
                           bias, alpha, beta, y, T, D, nTB, lanes);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * T * D;
    hipLaunchKernelGGL(dwconv1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, alpha,
                       beta, y, T, D, K, total);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
crnn_seq_kernel(const float* __restrict__ in, float* __restrict__ out, int CH, int W, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // out index: (b*W + w)*CH + ch
    if (idx >= total) return;
    const int ch = (int)(idx % CH);
    const size_t bw = idx / CH;
    const int w = (int)(bw % W);
    const size_t b = bw / W;
    out[idx] = in[(b * CH + ch) * (size_t)W + w];
}
hipError_t launch_crnn_seq(const float* in, float* out, int B, int C, int H, int W, hipStream_t s) {
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(crnn_seq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, C * H, W, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ attention core
// One workgroup per (clip, head).  K and V of the head live in LDS; lane t owns query row t and walks the
// keys with an online softmax (running max / sum), every lane reading the same K/V element (LDS broadcast).
// DH = the compiled width, dh <= DH the head's real width: the columns beyond dh are zeros (q . k gains + 0 terms, the sums are
// unchanged bit for bit), so any head dim runs on the next compiled instance
template <int DH>
__global__ void __launch_bounds__(128)
mha_core_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int D, int dh, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + (size_t)T * DH;
    const int head = blockIdx.x, b = blockIdx.y;
    const float* base = qkv + (size_t)b * T * 3 * D;
    for (int i = threadIdx.x; i < T * DH; i += blockDim.x) {
        const int t = i / DH, c = i - t * DH;
        Ks[i] = c < dh ? base[(size_t)t * 3 * D + D + head * dh + c] : 0.0f;
        Vs[i] = c < dh ? base[(size_t)t * 3 * D + 2 * D + head * dh + c] : 0.0f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float q[DH], o[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { q[c] = c < dh ? base[(size_t)t * 3 * D + head * dh + c] * scale : 0.0f; o[c] = 0.0f; }
        float mx = -INFINITY, den = 0.0f;
        for (int j = 0; j < T; ++j) {
            const float* kj = Ks + (size_t)j * DH;
            float sc = 0.0f;
#pragma unroll
            for (int c = 0; c < DH; ++c) sc = fmaf(q[c], kj[c], sc);
            const float nm = fmaxf(mx, sc);
            const float corr = expf(mx - nm), pj = expf(sc - nm);
            den = den * corr + pj;
            const float* vj = Vs + (size_t)j * DH;
#pragma unroll
            for (int c = 0; c < DH; ++c) o[c] = fmaf(pj, vj[c], o[c] * corr);
            mx = nm;
        }
        const float inv = 1.0f / den;
        float* op = out + ((size_t)b * T + t) * D + head * dh;
#pragma unroll
        for (int c = 0; c < DH; ++c)
            if (c < dh) op[c] = o[c] * inv;
    }
}

// compiled widths (q and the output row live in registers, so the width is a template parameter): every multiple of 4 up to 72,
// 18 (the reference's default d_model = 144 over 8 heads), then 80 / 96 / 112 / 128; a head dim in between runs zero-padded on the
// next one (nn.MultiheadAttention takes any embed_dim divisible by num_heads, architectures.py:499)
#define NWW_MHA_HEAD_DIMS(X) X(4) X(8) X(12) X(16) X(18) X(20) X(24) X(28) X(32) X(36) X(40) X(44) X(48) X(52) X(56) \
    X(60) X(64) X(68) X(72) X(80) X(96) X(112) X(128)
static int mha_compiled_width(int dh) {
    static const int widths[] = {4, 8, 12, 16, 18, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 80, 96, 112, 128};
    for (int w : widths)
        if (w >= dh) return w;
    return 0;
}
hipError_t launch_mha_core(const float* qkv, float* out, int B, int T, int D, int n_head, hipStream_t s) {
    const int dh = D / n_head, dhw = mha_compiled_width(dh);
    const size_t lds = (size_t)2 * T * dhw * sizeof(float);
    if (dhw == 0 || lds > 150 * 1024 || dh * n_head != D) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)dh);
    dim3 grid(n_head, B);
#define MHA_CASE(DHV)                                                                                              \
    case DHV: {                                                                                                    \
        hipError_t ea = nww_allow_lds(reinterpret_cast<const void*>(mha_core_kernel<DHV>), lds);                   \
        if (ea != hipSuccess) return ea;                                                                           \
        hipLaunchKernelGGL((mha_core_kernel<DHV>), grid, dim3(128), lds, s, qkv, out, T, D, dh, scale);            \
        break;                                                                                                     \
    }
    switch (dhw) {
        NWW_MHA_HEAD_DIMS(MHA_CASE)
        default: return hipErrorInvalidValue;
    }
#undef MHA_CASE
    return hipGetLastError();
}
bool mha_head_dim_supported(int dh) { return dh >= 1 && mha_compiled_width(dh) > 0; }

// ------------------------------------------------------------------------------------------ GRU recurrence
// Gate functions of the register-resident recurrences on the hardware exp2 and reciprocal (1 ulp each): absolute error
// <= 2e-7 against ~30 (sigmoid: expf + IEEE division) and ~40 (tanhf) instructions each - the gate arithmetic of a step was
// as long as its matrix products.  tanh x = 1 - 2 / (1 + e^2x) saturates correctly through exp2 = 0 / inf.
__device__ __forceinline__ float rnn_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float rnn_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * v)); }
// One workgroup = 32 clips x all H hidden units; wave w owns hidden units [32w, 32w+32) and computes, per
// step, the three 32x32 gate tiles (r, z, n columns j, H+j, 2H+j) of  hg = h W_hh^T  on MFMA f32, with h as the
// A operand read from LDS ([32][H+4] floats) and W_hh rows streamed from L2.  Gate math runs in the MFMA C
// layout (lane = hidden unit j, 16 clips per lane), so xg loads and h stores are coalesced along j, and
// h_prev stays in registers across steps.  PyTorch semantics: r,z = sigmoid(xg + hg + b_hh);
// n = tanh(xg_n + r*(hg_n + b_hn)); h' = (1-z) n + z h.
__global__ void __launch_bounds__(512) gru_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);           // [32][H+4]
    const int H = a.H, ldh = H + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b0 = blockIdx.x * 32;
    const int j = wave * 32 + i;                              // hidden unit of this lane (C-layout column)
    const bool jok = j < H;
    const int jc = jok ? j : H - 1;
    for (int idx = threadIdx.x; idx < 32 * ldh; idx += blockDim.x) hs[idx] = 0.0f;
    float hprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hprev[r] = 0.0f;
    const float bhr = a.b_hh[jc], bhz = a.b_hh[H + jc], bhn = a.b_hh[2 * H + jc];
    const float* wr = a.w_hh + (size_t)jc * H + 4 * hh;       // B operand rows (col = lane&31 -> same jc)
    const float* wz = a.w_hh + (size_t)(H + jc) * H + 4 * hh;
    const float* wn = a.w_hh + (size_t)(2 * H + jc) * H + 4 * hh;
    const float* arow = hs + (size_t)i * ldh + 4 * hh;        // A operand row (clip i)
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        f32x16 ar, az, an;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ar[r] = 0.0f; az[r] = 0.0f; an[r] = 0.0f; }
        // the input-side gate pre-activations of this step do not depend on h: fetch them (HBM, one row per clip)
        // before the recurrent product so their latency hides under the MFMA loop
        float xr[16], xz[16], xn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            xr[r] = xz[r] = xn[r] = 0.0f;
            if (b < a.B && jok) {
                const float* xg = a.xg + ((size_t)b * a.T + t) * 3 * H;
                xr[r] = xg[j]; xz[r] = xg[H + j]; xn[r] = xg[2 * H + j];
            }
        }
        if (step > 0) {                                       // h == 0 on the first step
            // W_hh streams from L2 every step; the next 8-k slice is requested before the current one is multiplied
            float4 nbr = make_float4(0, 0, 0, 0), nbz = nbr, nbn = nbr;
            if (4 * hh + 4 <= H) {
                nbr = *reinterpret_cast<const float4*>(wr); nbz = *reinterpret_cast<const float4*>(wz);
                nbn = *reinterpret_cast<const float4*>(wn);
            }
            for (int k = 0; k < H; k += 8) {
                float4 av = make_float4(0, 0, 0, 0);
                const float4 br = nbr, bz = nbz, bn = nbn;
                if (k + 4 * hh + 4 <= H) av = *reinterpret_cast<const float4*>(arow + k);
                nbr = nbz = nbn = make_float4(0, 0, 0, 0);
                if (k + 8 + 4 * hh + 4 <= H) {
                    nbr = *reinterpret_cast<const float4*>(wr + k + 8);
                    nbz = *reinterpret_cast<const float4*>(wz + k + 8);
                    nbn = *reinterpret_cast<const float4*>(wn + k + 8);
                }
                ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, br.x, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bz.x, az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bn.x, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, br.y, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bz.y, az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bn.y, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, br.z, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bz.z, az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bn.z, an, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, br.w, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bz.w, az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bn.w, an, 0, 0, 0);
            }
        }
        __syncthreads();                                      // every wave has finished reading hs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;    // clip row within the block
            const int b = b0 + c;
            float hn = 0.0f;
            if (b < a.B && jok) {
                const float rg = 1.0f / (1.0f + expf(-(xr[r] + ar[r] + bhr)));
                const float zg = 1.0f / (1.0f + expf(-(xz[r] + az[r] + bhz)));
                const float ng = tanhf(xn[r] + rg * (an[r] + bhn));
                hn = (1.0f - zg) * ng + zg * hprev[r];
                if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                if (a.last_out && step == a.steps - 1) a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
            }
            hprev[r] = hn;
            if (jok) hs[(size_t)c * ldh + j] = hn;
        }
        __syncthreads();
    }
}

// Register-resident variant for H in {32, 64, 128} (H = 256 would need 384 weight registers at two waves per SIMD): one workgroup = 16 clips, wave w owns hidden units [32w, 32w+32)
// as two 16-wide column blocks per gate on v_mfma_f32_16x16x4_f32.  Lane (n = l&15, g = l>>4) feeds k = g*H/4 + s at MFMA
// step s, so its slice of every W_hh row it needs is H/4 CONTIGUOUS floats, loaded once and kept in 6*H/4 VGPRs for
// all steps (one wave per SIMD: the 512-register budget is there) - the 32-clip kernel above re-streams W_hh (3*H*H
// floats) from L2 on every step.  Half the clips per workgroup also means twice the workgroups (256 at B = 4096) and half
// the MFMA chain per step.  C layout: column = hidden unit, rows 4g..4g+3 = clips, so xg loads / h stores stay coalesced
// along the hidden dimension.
template <int H>
__global__ void __launch_bounds__(64 * (H / 32), 1) gru16_kernel(GruArgs a) {
    constexpr int KS = H / 4, LDH = H + 4;                    // MFMA steps per product, LDS row stride
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);           // [16][H+4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16;
    for (int idx = threadIdx.x; idx < 16 * LDH; idx += blockDim.x) hs[idx] = 0.0f;
    // W_hh -> registers: gate q (r, z, n), column block bl, k slice g
    float wreg[3][2][KS];
    float bh[3][2];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            const int j = 32 * wave + 16 * bl + n;
            const float4* src = reinterpret_cast<const float4*>(a.w_hh + (size_t)(q * H + j) * H + g * KS);
#pragma unroll
            for (int s4 = 0; s4 < KS / 4; ++s4) {
                const float4 v = src[s4];
                wreg[q][bl][4 * s4] = v.x; wreg[q][bl][4 * s4 + 1] = v.y; wreg[q][bl][4 * s4 + 2] = v.z; wreg[q][bl][4 * s4 + 3] = v.w;
            }
            bh[q][bl] = a.b_hh[q * H + j];
        }
    float hprev[2][4];
#pragma unroll
    for (int bl = 0; bl < 2; ++bl)
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[bl][r] = 0.0f;
    const float* arow = hs + n * LDH + g * KS;                // A operand: clip n, k slice g
    // input-side pre-activations (independent of h) are fetched ONE STEP AHEAD: a row per clip from HBM takes longer than a step's
    // recurrent product
    float xq[3][2][4], xnext[3][2][4];
    auto fetch = [&](int step, float (&x)[3][2][4]) {
        const int t = a.reverse ? a.T - 1 - step : step;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + 4 * g + r;
            const float* xg = a.xg + ((size_t)min(b, a.B - 1) * a.T + t) * 3 * H + 32 * wave + n;
#pragma unroll
            for (int q = 0; q < 3; ++q) { x[q][0][r] = xg[q * H]; x[q][1][r] = xg[q * H + 16]; }
        }
    };
    fetch(0, xq);
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        f32x4 acc[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int bl = 0; bl < 2; ++bl) acc[q][bl] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step + 1 < a.steps) fetch(step + 1, xnext);
        if (step > 0) {                                       // h == 0 on the first step
#pragma unroll
            for (int s4 = 0; s4 < KS / 4; ++s4) {
                const float4 av = *reinterpret_cast<const float4*>(arow + 4 * s4);
                const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int bl = 0; bl < 2; ++bl)
                            acc[q][bl] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], wreg[q][bl][4 * s4 + e], acc[q][bl], 0, 0, 0);
            }
        }
        __syncthreads();                                      // every wave has finished reading hs
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            const int j = 32 * wave + 16 * bl + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * g + r, b = b0 + c;
                float hn = 0.0f;
                if (b < a.B) {
                    const float rg = rnn_sigmoid(xq[0][bl][r] + acc[0][bl][r] + bh[0][bl]);
                    const float zg = rnn_sigmoid(xq[1][bl][r] + acc[1][bl][r] + bh[1][bl]);
                    const float ng = rnn_tanh(xq[2][bl][r] + rg * (acc[2][bl][r] + bh[2][bl]));
                    hn = (1.0f - zg) * ng + zg * hprev[bl][r];
                    if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                    if (a.last_out && step == a.steps - 1) a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
                }
                hprev[bl][r] = hn;
                hs[c * LDH + j] = hn;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                for (int r = 0; r < 4; ++r) xq[q][bl][r] = xnext[q][bl][r];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ LSTM recurrence
// nn.LSTM cell (PyTorch gate order i, f, g, o in the 4H rows of W_ih / W_hh): with xg = x W_ih^T + b_ih precomputed,
//   i, f, o = sigmoid(xg + h W_hh^T + b_hh), g = tanh(.), c' = f c + i g, h' = o tanh(c')
// (CRNNModel's default backend: nanowakeword/modules/architectures.py:247-254, model.py:214).  Same argument block as the
// GRU (xg is [B][T][4H]); same "reverse direction of the last layer runs one step" shortcut in the plan.
//
// Generic kernel (any H % 4 == 0, H <= 512): 32 clips per workgroup, wave w owns hidden units [32w, 32w+32), W_hh
// streamed from L2 every step, four 32x32 accumulators on v_mfma_f32_32x32x2_f32; h and c stay in registers in the
// C layout, h also in LDS as the next step's A operand.
__global__ void __launch_bounds__(512) lstm_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);           // [32][H+4]
    const int H = a.H, ldh = H + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b0 = blockIdx.x * 32;
    const int j = wave * 32 + i;
    const bool jok = j < H;
    const int jc = jok ? j : H - 1;
    for (int idx = threadIdx.x; idx < 32 * ldh; idx += blockDim.x) hs[idx] = 0.0f;
    float cprev[16];                                         // (the previous h is not needed by the LSTM cell update: only c is carried)
#pragma unroll
    for (int r = 0; r < 16; ++r) cprev[r] = 0.0f;
    float bh[4];
    const float* wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bh[q] = a.b_hh[q * H + jc];
        wq[q] = a.w_hh + (size_t)(q * H + jc) * H + 4 * hh;
    }
    const float* arow = hs + (size_t)i * ldh + 4 * hh;
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        // (opaque per step: the 16 gate-row and output addresses of the lane are not hoisted out of the step loop and spilled)
        int hh_o = lane >> 5;
        asm volatile("" : "+v"(hh_o));
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
        float xq[4][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * hh_o;
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[q][r] = 0.0f;
            if (b < a.B && jok) {
                const float* xg = a.xg + ((size_t)b * a.T + t) * 4 * H;
#pragma unroll
                for (int q = 0; q < 4; ++q) xq[q][r] = xg[q * H + j];
            }
        }
        if (step > 0) {
            for (int k = 0; k < H; k += 8) {
                float4 av = make_float4(0, 0, 0, 0), bw[4];
                const bool ok = k + 4 * hh + 4 <= H;
                if (ok) av = *reinterpret_cast<const float4*>(arow + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) bw[q] = ok ? *reinterpret_cast<const float4*>(wq[q] + k) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw[q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw[q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw[q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw[q].w, acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * hh_o;
            const int b = b0 + c;
            float hn = 0.0f, cn = 0.0f;
            if (b < a.B && jok) {
                const float ig = 1.0f / (1.0f + expf(-(xq[0][r] + acc[0][r] + bh[0])));
                const float fg = 1.0f / (1.0f + expf(-(xq[1][r] + acc[1][r] + bh[1])));
                const float gg = tanhf(xq[2][r] + acc[2][r] + bh[2]);
                const float og = 1.0f / (1.0f + expf(-(xq[3][r] + acc[3][r] + bh[3])));
                cn = fg * cprev[r] + ig * gg;
                hn = og * tanhf(cn);
                if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                if (a.last_out && step == a.steps - 1) a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
            }
            cprev[r] = cn;
            if (jok) hs[(size_t)c * ldh + j] = hn;
        }
        __syncthreads();
    }
}

// Register-resident variant for H in {32, 64, 128}: 16 clips per workgroup, wave w owns the 16 hidden units
// [16w, 16w+16) of all four gates on v_mfma_f32_16x16x4_f32; its W_hh slices (4 gates x H/4 contiguous floats per lane)
// are loaded once and stay in 4*H/4 VGPRs for every step (H = 128: 128 registers, eight waves = two per SIMD).
template <int H>
__global__ void __launch_bounds__(64 * (H / 16), 1) lstm16_kernel(GruArgs a) {
    constexpr int KS = H / 4, LDH = H + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hs = reinterpret_cast<float*>(smem_raw);           // [16][H+4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int j = 16 * wave + n;
    for (int idx = threadIdx.x; idx < 16 * LDH; idx += blockDim.x) hs[idx] = 0.0f;
    float wreg[4][KS], bh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4* src = reinterpret_cast<const float4*>(a.w_hh + (size_t)(q * H + j) * H + g * KS);
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
            const float4 v = src[s4];
            wreg[q][4 * s4] = v.x; wreg[q][4 * s4 + 1] = v.y; wreg[q][4 * s4 + 2] = v.z; wreg[q][4 * s4 + 3] = v.w;
        }
        bh[q] = a.b_hh[q * H + j];
    }
    float hprev[4], cprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { hprev[r] = 0.0f; cprev[r] = 0.0f; }
    const float* arow = hs + n * LDH + g * KS;
    float xq[4][4], xnext[4][4];                              // input-side pre-activations, fetched one step ahead (gru16_kernel)
    auto fetch = [&](int step, float (&x)[4][4]) {
        const int t = a.reverse ? a.T - 1 - step : step;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + 4 * g + r;
            const float* xg = a.xg + ((size_t)min(b, a.B - 1) * a.T + t) * 4 * H + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q][r] = xg[q * H];
        }
    };
    fetch(0, xq);
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step + 1 < a.steps) fetch(step + 1, xnext);
        if (step > 0) {
#pragma unroll
            for (int s4 = 0; s4 < KS / 4; ++s4) {
                const float4 av = *reinterpret_cast<const float4*>(arow + 4 * s4);
                const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], wreg[q][4 * s4 + e], acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * g + r, b = b0 + c;
            float hn = 0.0f, cn = 0.0f;
            if (b < a.B) {
                const float ig = rnn_sigmoid(xq[0][r] + acc[0][r] + bh[0]);
                const float fg = rnn_sigmoid(xq[1][r] + acc[1][r] + bh[1]);
                const float gg = rnn_tanh(xq[2][r] + acc[2][r] + bh[2]);
                const float og = rnn_sigmoid(xq[3][r] + acc[3][r] + bh[3]);
                cn = fg * cprev[r] + ig * gg;
                hn = og * rnn_tanh(cn);
                if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                if (a.last_out && step == a.steps - 1) a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
            }
            hprev[r] = hn; cprev[r] = cn;
            hs[c * LDH + j] = hn;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) xq[q][r] = xnext[q][r];
        __syncthreads();
    }
}

// the split-operand recurrence serves this layer (plan time: the plan folds the reverse direction's single step into it)
// ------------------------------------------------------------------------------------------ any-width recurrence
// nn.GRU / nn.LSTM take any hidden_size (architectures.py:132-145,238-254); the kernels above want H % 4 == 0 (16-byte weight rows)
// and H <= 256 (a wave per 32 hidden units).  This one takes any H <= 512: W_hh re-laid at plan time as [G H][ldw] rows padded with
// zeros to ldw = H rounded up to 8 (rnn_pad_rows_kernel), a wave walks tiles wave, wave + 8 (32 hidden units each) one after the
// other within a step, and h lives in TWO LDS buffers (read step t, write step t + 1: one barrier per step).  Gate arithmetic as in
// gru_kernel / lstm_kernel (expf / tanhf).  G = 3: GRU, 4: LSTM.
__global__ void __launch_bounds__(256) rnn_pad_rows_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int H, int ldw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)rows * ldw) return;
    const int r = (int)(idx / ldw), k = (int)(idx - (size_t)r * ldw);
    out[idx] = k < H ? w[(size_t)r * H + k] : 0.0f;
}
size_t rnn_wide_weight_bytes(int gates, int H) { return (size_t)gates * H * ((H + 7) & ~7) * sizeof(float); }
hipError_t launch_rnn_pad_weights(const float* w_hh, float* out, int gates, int H, hipStream_t s) {
    const int ldw = (H + 7) & ~7;
    const size_t total = (size_t)gates * H * ldw;
    hipLaunchKernelGGL(rnn_pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_hh, out, gates * H, H, ldw);
    return hipGetLastError();
}

// gate functions on the hardware exp2 / rcp (abs. error <= 2e-7, as rnn_x3.hip / rnn_stream.hip; the library expf / tanhf cost this kernel
// 22 (GRU) / 52 (LSTM) spilled registers under its 256-register budget: VERDICT r05)
__device__ __forceinline__ float wide_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float wide_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * v)); }
template <int G>
__global__ void __launch_bounds__(512) rnn_wide_kernel(GruArgs a) {
    constexpr int TP = 2;                                     // tiles per wave: 8 waves x 2 x 32 = 512 hidden units
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int H = a.H, ldw = a.ldw, ldh = ldw + 4;
    float* hbuf[2] = {reinterpret_cast<float*>(smem_raw), reinterpret_cast<float*>(smem_raw) + (size_t)32 * ldh};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b0 = blockIdx.x * 32;
    const int ntile = (H + 31) / 32;
    for (int idx = threadIdx.x; idx < 2 * 32 * ldh; idx += blockDim.x) hbuf[0][idx] = 0.0f;
    // the previous h of a (clip, unit) is read back from the LDS plane the products use (zero at step 0); only the LSTM's cell state lives in
    // registers (h in registers as well cost 20 / 50 spilled ones under the 256-register budget of sixteen... eight waves)
    float cprev[G == 4 ? TP : 1][16];
#pragma unroll
    for (int tp = 0; tp < (G == 4 ? TP : 1); ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) cprev[tp][r] = 0.0f;
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        const int t = a.reverse ? a.T - 1 - step : step;
        const float* cur = hbuf[step & 1];
        float* nxt = hbuf[(step & 1) ^ 1];
        // lane coordinates re-derived from an opaque copy every step: from the plain ones the 16 output addresses per tile are loop-invariant,
        // get hoisted out of the step loop and spill (50 registers in the LSTM instance)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int i = lane_o & 31, hh = lane_o >> 5;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
            const int tile = wave + 8 * tp;
            if (tile >= ntile) continue;                      // (wave-uniform)
            const int j = tile * 32 + i;
            const bool jok = j < H;
            const int jc = jok ? j : H - 1;
            f32x16 acc[G];
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
            if (step > 0) {
                const float* arow = cur + (size_t)i * ldh + 4 * hh;
                const float* wq[G];
#pragma unroll
                for (int q = 0; q < G; ++q) wq[q] = a.w_hh + (size_t)(q * H + jc) * ldw + 4 * hh;
                for (int k = 0; k < ldw; k += 8) {
                    const float4 av = *reinterpret_cast<const float4*>(arow + k);
                    float4 bw[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) bw[q] = *reinterpret_cast<const float4*>(wq[q] + k);
#pragma unroll
                    for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw[q].x, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw[q].y, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw[q].z, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw[q].w, acc[q], 0, 0, 0);
                }
            }
            float bh[G];
#pragma unroll
            for (int q = 0; q < G; ++q) bh[q] = a.b_hh[q * H + jc];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int b = b0 + c;
                float hn = 0.0f, cn = 0.0f;
                if (b < a.B && jok) {
                    const float* xg = a.xg + ((size_t)b * a.T + t) * G * H + j;
                    if (G == 3) {
                        const float rg = wide_sigmoid(xg[0] + acc[0][r] + bh[0]);
                        const float zg = wide_sigmoid(xg[H] + acc[1][r] + bh[1]);
                        const float ng = wide_tanh(xg[2 * H] + rg * (acc[2][r] + bh[2]));
                        hn = (1.0f - zg) * ng + zg * cur[(size_t)c * ldh + j];
                    } else {
                        const float ig = wide_sigmoid(xg[0] + acc[0][r] + bh[0]);
                        const float fg = wide_sigmoid(xg[H] + acc[1][r] + bh[1]);
                        const float gg = wide_tanh(xg[2 * H] + acc[2][r] + bh[2]);
                        const float og = wide_sigmoid(xg[3 * H] + acc[G - 1][r] + bh[G - 1]);
                        cn = fg * cprev[G == 4 ? tp : 0][r] + ig * gg;
                        hn = og * wide_tanh(cn);
                    }
                    if (a.seq_out) a.seq_out[((size_t)b * a.T + t) * a.ld_seq + a.col_off + j] = hn;
                    if (a.last_out && step == a.steps - 1) a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hn;
                }
                if (G == 4) cprev[G == 4 ? tp : 0][r] = cn;
                if (jok) nxt[(size_t)c * ldh + j] = hn;
            }
        }
        __syncthreads();
    }
}

static hipError_t launch_rnn_wide(const GruArgs& a, int gates, hipStream_t s) {
    if (a.H < 1 || a.H > 512 || a.ldw != ((a.H + 7) & ~7) || a.xg2) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * 32 * (a.ldw + 4) * sizeof(float);
    const dim3 grid((a.B + 31) / 32);
    if (gates == 3) {
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(rnn_wide_kernel<3>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rnn_wide_kernel<3>, grid, dim3(512), lds, s, a);
    } else {
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(rnn_wide_kernel<4>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rnn_wide_kernel<4>, grid, dim3(512), lds, s, a);
    }
    return hipGetLastError();
}

bool rnn_x3_enabled(const GruArgs& a) {
    static const int use16 = 1;
    return use16 && rnn_x3_usable(a);
}

hipError_t launch_lstm(const GruArgs& a, hipStream_t s) {
    if (a.w_packed) return launch_rnn_stream(a, 4, s);
    if (a.ldw) return launch_rnn_wide(a, 4, s);                      // padded weights: the any-width kernel (planned for H % 4 != 0 or H > 256)
    if (a.H % 4 != 0 || a.H > 256) return hipErrorInvalidValue;      // a wave per 32 hidden units, 512 threads
    static const int use16 = 1;
    if (rnn_x3_enabled(a)) return launch_rnn_x3(a, 4, s);
    if (a.xg2) return hipErrorInvalidValue;                           // only rnn_x3 folds the opposite direction's step
    if (use16 && (a.H == 32 || a.H == 64 || a.H == 128) && (reinterpret_cast<uintptr_t>(a.w_hh) & 15) == 0) {
        const size_t lds16 = (size_t)16 * (a.H + 4) * sizeof(float);
        const dim3 grid((a.B + 15) / 16);
        switch (a.H) {
            case 32: hipLaunchKernelGGL(lstm16_kernel<32>, grid, dim3(128), lds16, s, a); break;
            case 64: hipLaunchKernelGGL(lstm16_kernel<64>, grid, dim3(256), lds16, s, a); break;
            default: hipLaunchKernelGGL(lstm16_kernel<128>, grid, dim3(512), lds16, s, a); break;
        }
        return hipGetLastError();
    }
    const int waves = (a.H + 31) / 32;
    const size_t lds = (size_t)32 * (a.H + 4) * sizeof(float);
    {
        hipError_t ea = nww_allow_lds(reinterpret_cast<const void*>(lstm_kernel), lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(lstm_kernel, dim3((a.B + 31) / 32), dim3(waves * 64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gru(const GruArgs& a, hipStream_t s) {
    if (a.w_packed) return launch_rnn_stream(a, 3, s);
    if (a.ldw) return launch_rnn_wide(a, 3, s);
    if (a.H % 4 != 0 || a.H > 256) return hipErrorInvalidValue;      // a wave per 32 hidden units, 512 threads
    static const int use16 = 1;
    if (rnn_x3_enabled(a)) return launch_rnn_x3(a, 3, s);
    if (a.xg2) return hipErrorInvalidValue;                           // only rnn_x3 folds the opposite direction's step
    if (use16 && (a.H == 32 || a.H == 64 || a.H == 128) && (reinterpret_cast<uintptr_t>(a.w_hh) & 15) == 0) {
        const size_t lds16 = (size_t)16 * (a.H + 4) * sizeof(float);
        const dim3 grid((a.B + 15) / 16);
        switch (a.H) {
            case 32: hipLaunchKernelGGL(gru16_kernel<32>, grid, dim3(64), lds16, s, a); break;
            case 64: hipLaunchKernelGGL(gru16_kernel<64>, grid, dim3(128), lds16, s, a); break;
            default: hipLaunchKernelGGL(gru16_kernel<128>, grid, dim3(256), lds16, s, a); break;
        }
        return hipGetLastError();
    }
    const int waves = (a.H + 31) / 32;
    const size_t lds = (size_t)32 * (a.H + 4) * sizeof(float);
    {
        hipError_t ea = nww_allow_lds(reinterpret_cast<const void*>(gru_kernel), lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(gru_kernel, dim3((a.B + 31) / 32), dim3(waves * 64), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ classifier tail
// 16 clips per workgroup pass; x, emb and hid tiles in LDS, weights through L1 (lanes of a 16-lane group share a weight
// row -> broadcast loads).  ~10 k MACs per clip: this replaces three tiny GEMM launches and the sigmoid launch, whose
// cost was their launch-to-launch latency, not their arithmetic.  Every output is one fmaf chain in ascending k.
// up to four outputs of one clip at once: four independent fmaf chains (ascending k) sharing the x fragment, operands
// fetched 16 bytes at a time so that loads of the next k block are in flight while this one is multiplied
__device__ __forceinline__ void tail_dot4(const float* __restrict__ xr, const float* __restrict__ w0, const float* __restrict__ w1,
                                          const float* __restrict__ w2, const float* __restrict__ w3, int K, bool vec, float (&acc)[4]) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
    int k = 0;
    if (vec) {
#pragma unroll 2
        for (; k + 4 <= K; k += 4) {
            const float4 x = *reinterpret_cast<const float4*>(xr + k);
            const float4 a = *reinterpret_cast<const float4*>(w0 + k), b = *reinterpret_cast<const float4*>(w1 + k);
            const float4 c = *reinterpret_cast<const float4*>(w2 + k), d = *reinterpret_cast<const float4*>(w3 + k);
            acc[0] = fmaf(x.x, a.x, acc[0]); acc[1] = fmaf(x.x, b.x, acc[1]); acc[2] = fmaf(x.x, c.x, acc[2]); acc[3] = fmaf(x.x, d.x, acc[3]);
            acc[0] = fmaf(x.y, a.y, acc[0]); acc[1] = fmaf(x.y, b.y, acc[1]); acc[2] = fmaf(x.y, c.y, acc[2]); acc[3] = fmaf(x.y, d.y, acc[3]);
            acc[0] = fmaf(x.z, a.z, acc[0]); acc[1] = fmaf(x.z, b.z, acc[1]); acc[2] = fmaf(x.z, c.z, acc[2]); acc[3] = fmaf(x.z, d.z, acc[3]);
            acc[0] = fmaf(x.w, a.w, acc[0]); acc[1] = fmaf(x.w, b.w, acc[1]); acc[2] = fmaf(x.w, c.w, acc[2]); acc[3] = fmaf(x.w, d.w, acc[3]);
        }
    }
    for (; k < K; ++k) {
        const float x = xr[k];
        acc[0] = fmaf(x, w0[k], acc[0]); acc[1] = fmaf(x, w1[k], acc[1]); acc[2] = fmaf(x, w2[k], acc[2]); acc[3] = fmaf(x, w3[k], acc[3]);
    }
}

// CT clips per workgroup pass (16, or 4 for small batches: four times the threads per clip), NG = 256 / CT output groups; a thread
// computes outputs g, g + NG, g + 2 NG, g + 3 NG at a time, each ONE fmaf chain in ascending k whatever CT is - a clip's results
// do not depend on the tile shape (batch invariance).
template <int ACT, int CT>
__global__ void __launch_bounds__(256) classifier_tail_kernel(TailArgs a) {
    constexpr int NG = 256 / CT;
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    const int Kin = a.Kin, E = a.E, Hd = E / 2;
    const int ldx = ((Kin + 3) & ~3) + 4, lde = ((E + 3) & ~3) + 4, ldh = Hd + 1;      // 16-byte aligned rows, +4: conflict-free
    float* xs = tsm;                      // [CT][ldx]
    float* es = xs + CT * ldx;            // [CT][lde]
    float* hs = es + CT * lde;            // [CT][ldh]
    float* ys = hs + CT * ldh;            // [16][ldx]: the DNN body's second tile (a.ln0_w only)
    // LayerNorm + activation of the 16 tile rows, in place or into another tile: a wave per row (four rows each), the row in four
    // registers per lane - layernorm_kernel's arithmetic (Kin <= 256)
    auto tile_ln = [&](const float* src, float* dst, const float* w, const float* b) {
        const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
        for (int r = (CT / 4) * wv; r < (CT / 4) * (wv + 1); ++r) {
            float v[4], sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int i = lane + 64 * j; v[j] = i < Kin ? src[r * ldx + i] : 0.0f; sum += v[j]; }
            const float mu = wave_sum(sum) / (float)Kin;
            float q = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; if (lane + 64 * j < Kin) q = fmaf(d, d, q); }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)Kin + 1e-5f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j;
                if (i < Kin) dst[r * ldx + i] = act_ct<ACT>((v[j] - mu) * rstd * w[i] + b[i]);
            }
        }
    };
    const int tid = threadIdx.x, c = tid % CT, g = tid / CT;                          // clip slot, output group
    const bool vx = (Kin & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.We) & 15) == 0);
    const bool ve = (E & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.W0) & 15) == 0);
    for (int b0 = blockIdx.x * CT; b0 < a.B; b0 += gridDim.x * CT) {
        for (int idx = tid; idx < CT * Kin; idx += 256) {
            const int cc = idx / Kin, k = idx - cc * Kin;
            float v = 0.0f;
            if (b0 + cc < a.B) {
                if (a.parts) {                                 // the producer's split-K partials: its reduce + epilogue, done here
                    const size_t o = (size_t)(b0 + cc) * Kin + k;
                    // all loads of a batch of eight partials in flight before the (ordered) adds
                    float acc = 0.0f;
                    for (int z0 = 0; z0 < a.nparts; z0 += 8) {
                        float pv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) pv[j] = z0 + j < a.nparts ? a.parts[(size_t)(z0 + j) * a.part_stride + o] : 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (z0 + j < a.nparts) acc = (z0 + j == 0) ? pv[j] : acc + pv[j];
                    }
                    v = acc + (a.in_bias ? a.in_bias[k] : 0.0f);
                    if (a.in_alpha) v = v * a.in_alpha[k] + a.in_beta[k];
                    v = act_apply(v, a.in_act);
                } else {
                    v = a.x[(size_t)(b0 + cc) * Kin + k];
                }
            }
            xs[cc * ldx + k] = v;
        }
        __syncthreads();
        if (a.ln0_w) {                                         // the DNN body (wave-uniform)
            tile_ln(xs, xs, a.ln0_w, a.ln0_b);
            __syncthreads();
            for (int m = 0; m < a.n_mid; ++m) {
                const float* Wm = a.mid_W[m];
                const bool vm = (Kin & 3) == 0 && ((reinterpret_cast<uintptr_t>(Wm) & 15) == 0);
                for (int e0 = g; e0 < Kin; e0 += 4 * NG) {
                    const int e1 = min(e0 + NG, Kin - 1), e2 = min(e0 + 2 * NG, Kin - 1), e3 = min(e0 + 3 * NG, Kin - 1);
                    float acc[4];
                    tail_dot4(xs + c * ldx, Wm + (size_t)e0 * Kin, Wm + (size_t)e1 * Kin, Wm + (size_t)e2 * Kin, Wm + (size_t)e3 * Kin, Kin, vm, acc);
                    const int eo[4] = {e0, e0 + NG, e0 + 2 * NG, e0 + 3 * NG};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (eo[r] < Kin) ys[c * ldx + eo[r]] = acc[r] + a.mid_b[m][eo[r]];
                }
                __syncthreads();
                tile_ln(ys, xs, a.mid_lnw[m], a.mid_lnb[m]);
                __syncthreads();
            }
        }
        // embedding: outputs e = g, g+16, ...; four at a time
        for (int e0 = g; e0 < E; e0 += 4 * NG) {
            const int e1 = min(e0 + NG, E - 1), e2 = min(e0 + 2 * NG, E - 1), e3 = min(e0 + 3 * NG, E - 1);
            float acc[4];
            tail_dot4(xs + c * ldx, a.We + (size_t)e0 * Kin, a.We + (size_t)e1 * Kin, a.We + (size_t)e2 * Kin, a.We + (size_t)e3 * Kin, Kin, vx, acc);
            const int es_[4] = {e0, e0 + NG, e0 + 2 * NG, e0 + 3 * NG};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (es_[r] < E) {
                    const float v = acc[r] + (a.be ? a.be[es_[r]] : 0.0f);
                    es[c * lde + es_[r]] = v;
                    if (b0 + c < a.B) a.emb[(size_t)(b0 + c) * E + es_[r]] = v;
                }
        }
        __syncthreads();
        for (int j0 = g; j0 < Hd; j0 += 4 * NG) {
            const int j1 = min(j0 + NG, Hd - 1), j2 = min(j0 + 2 * NG, Hd - 1), j3 = min(j0 + 3 * NG, Hd - 1);
            float acc[4];
            tail_dot4(es + c * lde, a.W0 + (size_t)j0 * E, a.W0 + (size_t)j1 * E, a.W0 + (size_t)j2 * E, a.W0 + (size_t)j3 * E, E, ve, acc);
            const int js[4] = {j0, j0 + NG, j0 + 2 * NG, j0 + 3 * NG};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (js[r] < Hd) hs[c * ldh + js[r]] = act_ct<ACT>(acc[r] + a.b0[js[r]]);
        }
        __syncthreads();
        if (tid < CT && b0 + tid < a.B) {                      // (tid 0 is among them when B >= 1: its stores precede the flag)
            const float* hr = hs + tid * ldh;
            float acc = 0.0f;
            for (int j = 0; j < Hd; ++j) acc = fmaf(hr[j], a.w3[j], acc);
            const float l = acc + a.b3[0];
            a.logits[b0 + tid] = l;
            if (a.probs) a.probs[b0 + tid] = 1.0f / (1.0f + expf(-l));
            if (a.done_flag) __threadfence_system();           // this thread's results reach the host before the barrier below
        }
        __syncthreads();
    }
    if (a.done_flag && gridDim.x == 1 && tid == 0) {           // everything above is visible system-wide before the word is
        __threadfence_system();
        __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

bool tail_supported(int Kin, int E) { return Kin >= 1 && Kin <= 512 && E >= 2 && E <= 256 && (E % 2) == 0; }

hipError_t launch_classifier_tail(const TailArgs& a, hipStream_t s) {
    if (!tail_supported(a.Kin, a.E)) return hipErrorInvalidValue;
    if (a.ln0_w && (a.Kin > 256 || a.n_mid < 0 || a.n_mid > 4)) return hipErrorInvalidValue;
    const size_t lds = (size_t)16 * ((a.ln0_w ? 2 : 1) * (((a.Kin + 3) & ~3) + 4) + (((a.E + 3) & ~3) + 4) + (a.E / 2 + 1)) * sizeof(float);
    // small batches: four clips per workgroup (64 threads per clip instead of 16; measured at B = 32 / 1024 / 4096: DNN body + tail
    // 0.035 -> 0.021 ms, CRNN tail 0.0255 -> 0.022, CNN tail 0.020 -> 0.029).  B = 5 .. 16 keep ONE 16-clip workgroup: the
    // completion word of the interpreter's zero-copy calls is written by a launch of a single workgroup only
    static const int ct4_max = 1024;
    const bool small = a.B <= 4 || (a.B > 16 && a.B <= ct4_max);
    const int ct = small ? 4 : 16;
    int grid = (a.B + ct - 1) / ct;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    if (small) {
#define TAIL_CALL(A) hipLaunchKernelGGL((classifier_tail_kernel<A, 4>), dim3(grid), dim3(256), lds, s, a)
        NWW_DISPATCH_ACT(a.act, TAIL_CALL)
#undef TAIL_CALL
    } else {
#define TAIL_CALL(A) hipLaunchKernelGGL((classifier_tail_kernel<A, 16>), dim3(grid), dim3(256), lds, s, a)
        NWW_DISPATCH_ACT(a.act, TAIL_CALL)
#undef TAIL_CALL
    }
    return hipGetLastError();
}
