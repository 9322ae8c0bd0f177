// bc_chain.hip - BcResNet block i (architectures.py:632-647) chained with block i + 1's depthwise 3x3, one clip per workgroup pass:
//     h[m][n]   = BN_s( xs[m] . Wsc[n] ) + act( BN_1( d[m] . Wpw[n] ) )       (dual_x3.hip's product; m = pixel of the clip)
//     d'[o][n]  = sum_taps dw'[tap][n] h[(oy sh - 1 + dy, ox sw - 1 + dx)][n]  (the next block's depthwise, pad 1)
//     xs'[o][n] = h[(oy sh, ox sw)][n]                                          (the next block's shortcut rows)
// h - the block's output - lives only in LDS: unchained, it is written by dual_x3 (102 KB per clip behind block 1), read back by the
// depthwise kernel and read a third time, strided, by the next dual_x3; here a clip's rows go HBM -> registers -> MFMA -> LDS plane ->
// depthwise -> HBM, and what reaches HBM is the quarter-size (d', xs') pair.  Block 1 -> 2: 204 + 129 KB of traffic per clip become
// 155; block 2 -> 3: 106 + 82 become 110.
//
// Persistent workgroups of eight waves, one per CU (the plane [pixels][N + 4] float32 + ALL packed weights of the block stay in LDS:
// 133 / 140 KB), looping over clips.  A wave's work item = 32 pixels x (all | half) of the 32-output blocks; the rows of the NEXT clip's
// items are loaded into registers as soon as the current ones are converted, so they fly under the MFMAs and the depthwise phase.
// Arithmetic = dual_x3.hip's AT = 3 (float32 tensors, two binary16 terms per operand, per-pixel scale) or AT = 2 (binary16 tensors with
// plan-time scales); the depthwise sums run in dwconv3x3_nhwc_x4_kernel's tap order and fmaf chain, so the float32 results are
// bit-identical to the unchained kernels'.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "layers.h"
#include "bc_chain.h"
#include "dual_x3.h"
#include "split_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CH_WAVES = 8, CH_THREADS = 64 * CH_WAVES;

template <int ACT>
__device__ __forceinline__ float chain_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return nww_silu(v);
    return v;
}

__host__ __device__ constexpr int chain_blk_bytes(int K16) { return (2 * K16 * 2 * 1024 + 512 + 4095) & ~4095; }

// K16 = K / 16; AT = 3 (float32 tensors) / 2 (scaled binary16 tensors) / 1 (bf16 tensors: a bf16 value times its row's power-of-two
// scale IS a binary16 value - one activation term, per-pixel scale as for float32); BSPLIT = items per 32-pixel group (each owns NB / BSPLIT output
// blocks); MAXR = items per wave and clip; SW = the next block's depthwise stride along x (1 or 2)
template <int K16, int ACT, int AT, int BSPLIT, int MAXR, int SW>
__global__ void __launch_bounds__(CH_THREADS) bc_chain_kernel(ChainArgs a) {
    constexpr int K = 16 * K16, N = 2 * K, NB = N / 32, NBW = NB / BSPLIT;
    constexpr int FRAG = 2 * K16 * 2 * 1024, BLK = chain_blk_bytes(K16);
    constexpr int PITCH = N + 4;                               // floats per pixel of the plane (16-byte stores of 32 lanes: conflict-free)
    constexpr int Q = N / 4;                                   // depthwise phase: a thread owns a channel quad of four outputs along x
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wl = lds;                                   // [NB][BLK] packed weights + folded BN
    float* dww = reinterpret_cast<float*>(lds + NB * BLK);     // [9][N]
    float* zpix = dww + 9 * N;                                 // one all-zero pixel: where the out-of-plane taps read
    float* plane = zpix + PITCH;                               // [P][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int P = a.H * a.W, Po = a.Ho * a.Wo;
    const int ngroups = (P + 31) / 32, nitems = ngroups * BSPLIT;

    for (int i = tid; i < NB * BLK / 16; i += CH_THREADS) reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(a.packed)[i];
    for (int i = tid; i < 9 * N / 4; i += CH_THREADS) reinterpret_cast<float4*>(dww)[i] = reinterpret_cast<const float4*>(a.dw_wt)[i];
    if (tid < PITCH) zpix[tid] = 0.0f;

    // ---- register stage: the raw half rows (features 16 kb + 8 h + e of d and xs) of the wave's items of ONE clip
    typedef typename std::conditional<AT == 3, float4, uint2>::type raw_t;       // 4 features
    raw_t raw[MAXR][2][K16][2];
    int item_g[MAXR], item_b[MAXR];
    bool item_ok[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int it = wave + CH_WAVES * r;
        item_ok[r] = it < nitems;
        const int itc = min(it, nitems - 1);                   // surplus waves repeat the last item (no stores): no divergent loads
        item_g[r] = itc / BSPLIT; item_b[r] = itc - item_g[r] * BSPLIT;
    }
    auto load_raw = [&](int r, int clip) {
        const int pix = min(32 * item_g[r] + n, P - 1);
        const size_t row = ((size_t)clip * P + pix) * K + 8 * h;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const raw_t* src = reinterpret_cast<const raw_t*>(AT == 3 ? static_cast<const void*>(static_cast<const float*>(p ? a.xs : a.d) + row)
                                                                      : static_cast<const void*>(static_cast<const uint16_t*>(p ? a.xs : a.d) + row));
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                raw[r][p][kb][0] = src[4 * kb];                // 16 features = four raw_t
                raw[r][p][kb][1] = src[4 * kb + 1];
            }
        }
    };
    int clip = blockIdx.x;
    if (clip < a.B) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) load_raw(r, clip);
    }
    __syncthreads();

    for (; clip < a.B; clip += gridDim.x) {
        const int nclip = min(clip + (int)gridDim.x, a.B - 1);
        // ---- products: item by item, the block's output into the plane
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            bf16x8 xf[2][K16][AT == 3 ? 2 : 1];
            float pin[2] = {1.0f, 1.0f};
            if constexpr (AT == 3) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float m = 0.0f;
#pragma unroll
                    for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float4 q = raw[r][p][kb][j];
                            m = fmaxf(fmaxf(m, fabsf(q.x)), fmaxf(fabsf(q.y), fmaxf(fabsf(q.z), fabsf(q.w))));
                        }
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
                    const float sc = __uint_as_float((268u - eb) << 23);
                    pin[p] = __uint_as_float((eb - 14u) << 23);
#pragma unroll
                    for (int kb = 0; kb < K16; ++kb) {
                        const float4 q0 = raw[r][p][kb][0], q1 = raw[r][p][kb][1];
                        uint32_t hi[4], lo[4];
                        nww_split2h(q0.x * sc, q0.y * sc, hi[0], lo[0]);
                        nww_split2h(q0.z * sc, q0.w * sc, hi[1], lo[1]);
                        nww_split2h(q1.x * sc, q1.y * sc, hi[2], lo[2]);
                        nww_split2h(q1.z * sc, q1.w * sc, hi[3], lo[3]);
                        xf[p][kb][0] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
                        xf[p][kb][1] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
                    }
                }
            } else if constexpr (AT == 1) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    uint32_t mb = 0;                           // largest |bf16| of the row as bits (integer order = magnitude order)
#pragma unroll
                    for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint2 q = raw[r][p][kb][j];
                            mb = max(max(mb, q.x & 0x7fffu), max((q.x >> 16) & 0x7fffu, max(q.y & 0x7fffu, (q.y >> 16) & 0x7fffu)));
                        }
                    mb = max(mb, (uint32_t)__shfl_xor((int)mb, 32, 64));
                    const uint32_t eb = min(max(mb >> 7, 16u), 254u);
                    const float sc = __uint_as_float((268u - eb) << 23);
                    pin[p] = __uint_as_float((eb - 14u) << 23);
#pragma unroll
                    for (int kb = 0; kb < K16; ++kb) {
                        const uint2 q0 = raw[r][p][kb][0], q1 = raw[r][p][kb][1];
                        const uint32_t u[4] = {q0.x, q0.y, q1.x, q1.y};
                        uint32_t o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const nww_f32x2 v = {__uint_as_float(u[e] << 16) * sc, __uint_as_float(u[e] & 0xffff0000u) * sc};
                            o[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nww_f16x2));
                        }
                        xf[p][kb][0] = __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int kb = 0; kb < K16; ++kb) {
                        const uint2 q0 = raw[r][p][kb][0], q1 = raw[r][p][kb][1];
                        xf[p][kb][0] = __builtin_bit_cast(bf16x8, make_uint4(q0.x, q0.y, q1.x, q1.y));
                    }
            }
            load_raw(r, nclip);                                // the next clip's rows of this item: in flight from here on
            const int pix = 32 * item_g[r] + n;
            const bool st_ok = item_ok[r] && pix < P;
            float* prow = plane + (size_t)min(pix, P - 1) * PITCH;
#pragma unroll 1
            for (int bi = 0; bi < NBW; ++bi) {
                const int blk = item_b[r] * NBW + bi;
                const unsigned char* wp = wl + blk * BLK + lane * 16;
                f32x16 acc[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[p][q] = 0.0f;
                // weight fragments one k step ahead of the MFMAs (LDS latency under the previous step's matrix instructions)
                f16x8 nw[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t) nw[p][t] = *reinterpret_cast<const f16x8*>(wp + ((p * K16) * 2 + t) * 1024);
#pragma unroll
                for (int kb = 0; kb < K16; ++kb) {
                    f16x8 cw[2][2];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int t = 0; t < 2; ++t) cw[p][t] = nw[p][t];
                    if (kb + 1 < K16) {
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int t = 0; t < 2; ++t) nw[p][t] = *reinterpret_cast<const f16x8*>(wp + ((p * K16 + kb + 1) * 2 + t) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const f16x8 x0 = __builtin_bit_cast(f16x8, xf[p][kb][0]);
                        // small terms first: lo*hi, hi*lo, hi*hi (dual_x3.hip: mfma3h / mfma2h)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cw[p][1], x0, acc[p], 0, 0, 0);
                        if constexpr (AT == 3)
                            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cw[p][0], __builtin_bit_cast(f16x8, xf[p][kb][1]), acc[p], 0, 0, 0);
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cw[p][0], x0, acc[p], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (AT == 3 || AT == 1) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { acc[0][q] *= pin[0]; acc[1][q] *= pin[1]; }
                }
                // lane (pixel n, half h), register 4g + q = output channel 32 blk + 8g + 4h + q
                const float* aff = reinterpret_cast<const float*>(wl + blk * BLK + FRAG) + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a1 = *reinterpret_cast<const float4*>(aff + 8 * g), b1 = *reinterpret_cast<const float4*>(aff + 32 + 8 * g);
                    const float4 as = *reinterpret_cast<const float4*>(aff + 64 + 8 * g), bs = *reinterpret_cast<const float4*>(aff + 96 + 8 * g);
                    float4 o;
                    o.x = (acc[1][4 * g + 0] * as.x + bs.x) + chain_act<ACT>(acc[0][4 * g + 0] * a1.x + b1.x);
                    o.y = (acc[1][4 * g + 1] * as.y + bs.y) + chain_act<ACT>(acc[0][4 * g + 1] * a1.y + b1.y);
                    o.z = (acc[1][4 * g + 2] * as.z + bs.z) + chain_act<ACT>(acc[0][4 * g + 2] * a1.z + b1.z);
                    o.w = (acc[1][4 * g + 3] * as.w + bs.w) + chain_act<ACT>(acc[0][4 * g + 3] * a1.w + b1.w);
                    if (st_ok) *reinterpret_cast<float4*>(prow + 32 * blk + 8 * g + 4 * h) = o;
                }
            }
        }
        __syncthreads();
        // ---- the next block's depthwise 3x3 and strided centres out of the plane: four outputs along x per thread, the 3 x ((4 - 1) SW + 3)
        // input columns read once, row by row (dwconv3x3_nhwc_x4_kernel's organisation, tap order and fmaf chain); out-of-plane taps
        // read the zero pixel (one select on the address, no divergent branches)
        {
            constexpr int NX = 4, COLS = (NX - 1) * SW + 3;
            const int Wg = (a.Wo + NX - 1) / NX, nwork = Q * Wg * a.Ho;
            for (int it = tid; it < nwork; it += CH_THREADS) {
                const int q = it % Q;
                const int t = it / Q, xg = t % Wg, oy = t / Wg;
                float4 w[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(dww + k * N + 4 * q);
                float4 acc[NX], centre[NX];
#pragma unroll
                for (int j = 0; j < NX; ++j) acc[j] = centre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int x_first = xg * NX * SW - 1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = oy * a.sh - 1 + dy;
                    const bool oky = yy >= 0 && yy < a.H;
                    const float* prow = plane + yy * a.W * PITCH + 4 * q;
                    float4 v[COLS];
#pragma unroll
                    for (int cx = 0; cx < COLS; ++cx) {
                        const int xx = x_first + cx;
                        const bool ok = oky && xx >= 0 && xx < a.W;
                        v[cx] = *reinterpret_cast<const float4*>(ok ? prow + xx * PITCH : zpix + 4 * q);
                    }
                    if (dy == 1) {
#pragma unroll
                        for (int j = 0; j < NX; ++j) centre[j] = v[j * SW + 1];
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int j = 0; j < NX; ++j) {
                            const float4 pv = v[j * SW + dx], qw = w[dy * 3 + dx];
                            acc[j].x = fmaf(pv.x, qw.x, acc[j].x); acc[j].y = fmaf(pv.y, qw.y, acc[j].y);
                            acc[j].z = fmaf(pv.z, qw.z, acc[j].z); acc[j].w = fmaf(pv.w, qw.w, acc[j].w);
                        }
                }
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const int ox = xg * NX + j;
                    if (ox < a.Wo) {
                        const size_t oi = ((size_t)clip * Po + (size_t)oy * a.Wo + ox) * N + 4 * q;
                        if constexpr (AT == 3) {
                            *reinterpret_cast<float4*>(static_cast<float*>(a.d_out) + oi) = acc[j];
                            *reinterpret_cast<float4*>(static_cast<float*>(a.xs_out) + oi) = centre[j];
                        } else {
                            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(a.d_out) + oi) =
                                make_uint2(nww_pk_act16(AT, acc[j].x, acc[j].y, a.d_mul), nww_pk_act16(AT, acc[j].z, acc[j].w, a.d_mul));
                            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(a.xs_out) + oi) =
                                make_uint2(nww_pk_act16(AT, centre[j].x, centre[j].y, a.xs_mul), nww_pk_act16(AT, centre[j].z, centre[j].w, a.xs_mul));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

size_t chain_lds_bytes(int K, int H, int W) {
    const int N = 2 * K;
    return (size_t)(N / 32) * chain_blk_bytes(K / 16) + (size_t)9 * N * 4 + (size_t)(H * W + 1) * (N + 4) * 4;
}
int chain_items(int K, int H, int W) { return ((H * W + 31) / 32) * (K == 64 ? 2 : 1); }

}  // namespace

bool bc_chain_supported(int K, int H, int W) {
    if ((K != 32 && K != 64) || H < 1 || W < 1) return false;
    // K = 32: up to two items per wave; K = 64: one (the raw rows of an item are 64 registers there)
    return chain_lds_bytes(K, H, W) <= (size_t)160 * 1024 && chain_items(K, H, W) <= (K == 32 ? 2 : 1) * CH_WAVES;
}

hipError_t launch_bc_chain(const ChainArgs& a, int K, int act, int max_grid, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (!bc_chain_supported(K, a.H, a.W) || (a.act16 < 0 || a.act16 > 2) || a.Ho < 1 || a.Wo < 1 || (a.sw != 1 && a.sw != 2)) return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(a.d) | reinterpret_cast<uintptr_t>(a.xs) | reinterpret_cast<uintptr_t>(a.d_out) | reinterpret_cast<uintptr_t>(a.xs_out) |
          reinterpret_cast<uintptr_t>(a.packed) | reinterpret_cast<uintptr_t>(a.dw_wt)) & 15) != 0)
        return hipErrorInvalidValue;
    const size_t lds = chain_lds_bytes(K, a.H, a.W);
    const int grid = a.B < max_grid ? a.B : max_grid;
    hipError_t e = hipSuccess;
#define CHAIN_GO(K16V, ACTV, ATV, BSV, MAXRV, SWV)                                                                     \
    {                                                                                                                  \
        e = nww_allow_lds(reinterpret_cast<const void*>(bc_chain_kernel<K16V, ACTV, ATV, BSV, MAXRV, SWV>), lds);      \
        if (e == hipSuccess) hipLaunchKernelGGL((bc_chain_kernel<K16V, ACTV, ATV, BSV, MAXRV, SWV>), dim3(grid), dim3(CH_THREADS), lds, s, a); \
    }
#define CHAIN_SW(K16V, ACTV, ATV, BSV, MAXRV)                                                                          \
    if (a.sw == 2) CHAIN_GO(K16V, ACTV, ATV, BSV, MAXRV, 2) else CHAIN_GO(K16V, ACTV, ATV, BSV, MAXRV, 1)
#define CHAIN_AT(K16V, ACTV, BSV, MAXRV)                                                                               \
    if (a.act16 == 2) { CHAIN_SW(K16V, ACTV, 2, BSV, MAXRV) } else if (a.act16 == 1) { CHAIN_SW(K16V, ACTV, 1, BSV, MAXRV) } else { CHAIN_SW(K16V, ACTV, 3, BSV, MAXRV) }
#define CHAIN_ACT(K16V, BSV, MAXRV)                                                                                    \
    switch (act) {                                                                                                     \
        case ACT_RELU: CHAIN_AT(K16V, ACT_RELU, BSV, MAXRV) break;                                                     \
        case ACT_GELU: CHAIN_AT(K16V, ACT_GELU, BSV, MAXRV) break;                                                     \
        case ACT_SILU: CHAIN_AT(K16V, ACT_SILU, BSV, MAXRV) break;                                                     \
        default: return hipErrorInvalidValue;                                                                          \
    }
    if (K == 32) {
        if (chain_items(K, a.H, a.W) <= CH_WAVES) CHAIN_ACT(2, 1, 1) else CHAIN_ACT(2, 1, 2)
    } else {
        CHAIN_ACT(4, 2, 1)
    }
#undef CHAIN_ACT
#undef CHAIN_AT
#undef CHAIN_SW
#undef CHAIN_GO
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
