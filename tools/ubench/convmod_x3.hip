// convmod_x3.hip - the Conformer's convolution module in ONE launch on the binary16 matrix cores (two-term operand splitting, "f16x3"):
//     h <- h + pw2( swish( BN( dwconv_k31( GLU( pw1( LayerNorm(h) ) ) ) ) ) )          (architectures.py:441-470 ConvolutionModule, :514-515)
// As three launches (lin_x3 LayerNorm + conv1 + GLU -> dwconv1d + BN + swish -> lin_x3 conv2 + residual: 0.101 + 0.077 + 0.073 ms at the
// BASELINE batch) the gated activations and the depthwise output make two round trips through HBM (4 x 119 MB).  The depthwise
// convolution runs along time inside a clip, so a workgroup owns ONE clip and the two tensors live in its LDS:
//   1. the four waves (32 rows each): rows -> LayerNorm -> the row's power of two -> two binary16 terms: B fragments (lin_x3.hip's prologue);
//      five 32-feature blocks of [a | b] weights (lin_x3's GLU packing, 40 KB each) stream through two LDS buffers by LDS-DMA;
//      Yt = W . Xt transposed, a * sigmoid(b) -> G [row][feature] float32 in LDS (pitch 148: conflict-free 16-byte stores);
//   2. thread = (channel pair, third of the clip) loads its window of G (rows t0 - 15 .. t0 + PR + 14, zeros outside the clip; lanes =
//      consecutive channel pairs: 8-byte LDS reads) into registers, everyone synchronises, then 31 taps + bias, folded BatchNorm, swish
//      are written back IN PLACE (same fmaf order per output as dwconv1d_blocked_kernel: bit-identical);
//   3. the waves read their rows back as B-fragment halves (two 16-byte reads per k-block), scale by the row's power of two, split, and run
//      conv2 (five 20 KB blocks) with the residual added in the transposed epilogue of lin_x3.hip (coalesced 128-byte row pieces).
// The arithmetic per element is that of the three launches (same scales, same order): results are bit-identical to them.
// 158 KB of LDS, one workgroup of four waves per CU.  (Eight waves - four helpers that only halve the depthwise phase - leave 256 registers
// per lane: the matrix waves then spill, and every scratch reload waits on vmcnt, which the in-order counter turns into a wait for the
// weight chunks just requested: 0.40 ms per launch against 0.25 for the three launches it replaces.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "layers.h"
#include "convmod_x3.h"
#include "lin_x3.h"
#include "split_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef NWW_TRACE      // tools/ubench/convmod_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of its second clip
__device__ unsigned long long g_cm_trace[4 * 32];
#define CM_STAMP(k) if (blockIdx.x == 0 && cm_it == 1 && lane == 0) g_cm_trace[wave * 32 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define CM_STAMP(k)
#endif

namespace {

__device__ __forceinline__ float cm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ f16x8 cm_ld(const unsigned char* p) { return *reinterpret_cast<const f16x8*>(p); }
// three products of two-term operands, small terms first (lin_x3.hip: mfma3hl): w = A (weights), x = B (rows)
__device__ __forceinline__ void cm_mfma3(const f16x8& wh, const f16x8& wl, const f16x8& xh, const f16x8& xl, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
}

// [D][KT] depthwise weights -> tap-major [KT][D] (lanes = consecutive channels read consecutive floats)
__global__ void __launch_bounds__(256) cm_taps_kernel(const float* __restrict__ w, float* __restrict__ wt, int D, int KT) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= D * KT) return;
    const int k = i / D, c = i - k * D;
    wt[i] = w[(size_t)c * KT + k];
}

// PR: rows per third of the clip (T <= 3 PR)
template <int K16, int PR>
__global__ void __launch_bounds__(256) convmod_x3_kernel(ConvModArgs a) {
    constexpr int K = 16 * K16, D = K, NB = (D + 31) / 32, KT = 31, HALO = KT / 2;
    constexpr int GP = D + 4, GR = 128;                        // G pitch (floats: an odd number of 16-byte slots), rows
    constexpr int FR = K16 * 2 * 1024, CH = (FR + 128 + 4095) & ~4095, STEPS = CH / 4096;     // a weight chunk: one 32-output tile's fragments, 32 biases
    constexpr int NCH = 3 * NB;                                // chunks per clip: conv1 a / b tile per block, then conv2's tiles
    static_assert(D % 16 == 0 && D <= 144 && 3 * PR <= GR + 2 && NB == 5, "convmod_x3 shape (the chunk schedule below is written for five blocks)");
    // Weight chunks stream through FOUR slots (separate LDS objects: the compiler then knows which in-flight LDS-DMA a read may alias),
    // chunk i of a clip in slot i % 4 (a sixteenth, empty chunk keeps the pattern periodic in the clip), requested THREE chunks ahead: a
    // chunk phase is 27 MFMAs (~1.5 k clocks), a trip to L2 two of them.  (First build: [a | b] blocks of 40 KB in two buffers, requested
    // one block ahead - every block then waited out its own fetch, 9 k clocks per block, 0.40 ms per launch against 0.25 for the three
    // separate launches.)
    __shared__ __attribute__((aligned(16))) float G[GR * GP];
    __shared__ __attribute__((aligned(16))) unsigned char ws0[CH];
    __shared__ __attribute__((aligned(16))) unsigned char ws1[CH];
    __shared__ __attribute__((aligned(16))) unsigned char ws2[CH];
    __shared__ __attribute__((aligned(16))) unsigned char ws3[CH];
    const int T = a.T;
    // thread coordinates are re-derived from an OPAQUE copy of threadIdx at the top of every clip: derived from the plain value, ~120 lane
    // addresses (LDS fragment offsets, G rows, residual pointers) are loop-invariant, get hoisted out of the clip loop, spill, and every
    // reload from scratch waits on vmcnt - i.e. on the weight chunks just requested
    int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n = lane & 31, h = lane >> 5;
    int tok = 32 * (wave & 3) + n, tokc = min(tok, T - 1);
    auto rederive = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        tid = t; lane = t & 63; wave = __builtin_amdgcn_readfirstlane(t >> 6);
        n = lane & 31; h = lane >> 5;
        tok = 32 * (wave & 3) + n; tokc = min(tok, T - 1);
    };

    float v[K16][8];                                           // the lane's half row (features 16 kb + 8 h + e): requested one clip ahead
    auto load_rows = [&](int clip) {
        const float* xrow = a.h + ((size_t)clip * T + tokc) * D;
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h);
            const float4 q0 = *reinterpret_cast<const float4*>(xrow + 16 * kb + 8 * h + 4);
            v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
            v[kb][4] = q0.x; v[kb][5] = q0.y; v[kb][6] = q0.z; v[kb][7] = q0.w;
        }
    };
    // the row's largest magnitude -> its power-of-two scale, two binary16 terms (lin_x3.hip)
    auto split_rows = [&](f16x8 (&xf)[K16][2], float w_un) {
        float m = 0.0f;
#pragma unroll
        for (int kb = 0; kb < K16; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[kb][e]));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const uint32_t eb = min(max(__float_as_uint(m) >> 23, 16u), 254u);
        const float sc = __uint_as_float((268u - eb) << 23);
#pragma unroll
        for (int kb = 0; kb < K16; ++kb) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nww_split2h(v[kb][2 * e] * sc, v[kb][2 * e + 1] * sc, hi[e], lo[e]);
            xf[kb][0] = __builtin_bit_cast(f16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
            xf[kb][1] = __builtin_bit_cast(f16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
        }
        return __uint_as_float((eb - 14u) << 23) * w_un;       // 1 / (row scale x weight scale)
    };

    constexpr bool MW = true;                                  // (every wave is a matrix wave; see the header)
    // chunk c of a clip (c < NCH; c >= NCH: chunk c - NCH - 1 of the next clip, NCH itself is the empty one) -> its slot, by LDS-DMA (matrix waves)
    auto fetch = [&](auto cc) {
        if constexpr (MW) {
            constexpr int C = decltype(cc)::value, c = C > NCH ? C - NCH - 1 : C;
            if constexpr (C != NCH) {
                const unsigned char* src = c < 2 * NB ? ((c & 1) ? a.packed1b : a.packed1a) + (size_t)(c >> 1) * CH : a.packed2 + (size_t)(c - 2 * NB) * CH;
                unsigned char* slot = (C & 3) == 0 ? ws0 : (C & 3) == 1 ? ws1 : (C & 3) == 2 ? ws2 : ws3;
                const unsigned char* sp = src + tid * 16;
                unsigned char* dst = slot + wave * 1024;       // wave-uniform
#pragma unroll
                for (int j = 0; j < STEPS; ++j)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sp + j * 4096),
                                                     (void __attribute__((address_space(3)))*)(dst + j * 4096), 16, 0, 0);
            }
        }
    };
    auto slot_of = [&](auto cc) -> const unsigned char* {
        constexpr int C = decltype(cc)::value;
        return (C & 3) == 0 ? ws0 : (C & 3) == 1 ? ws1 : (C & 3) == 2 ? ws2 : ws3;
    };
#define CMC(x) std::integral_constant<int, (x)>{}
    // end of a chunk phase: everything but the fetch this phase issued has landed (the fetch is the LAST vector-memory operation a phase issues
    // before this wait, so older loads, stores and the two chunks ahead are covered), then the barrier
    auto phase_end = [&](auto fetched) {
        if constexpr (MW) {
            if constexpr (decltype(fetched)::value) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STEPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    };
    if ((int)blockIdx.x < a.B) {
        fetch(CMC(0)); fetch(CMC(1));
        if constexpr (MW) load_rows((int)blockIdx.x);
    }
    [[maybe_unused]] int cm_it = 0;
    for (int clip = (int)blockIdx.x; clip < a.B; clip += (int)gridDim.x, ++cm_it) {
        rederive();
        CM_STAMP(0)
        // per-channel parameters are re-read every clip (L1 / K$ hits): their addresses are made opaque per iteration, or the compiler hoists
        // the 144 LayerNorm + 34 depthwise values per lane out of the clip loop and spills them (429 registers of scratch)
        const float *ln_w = a.ln_w, *ln_b = a.ln_b, *dw_t = a.dw_t, *dw_b = a.dw_b, *bn_a = a.bn_a, *bn_b = a.bn_b;
        asm volatile("" : "+s"(ln_w), "+s"(ln_b), "+s"(dw_t), "+s"(dw_b), "+s"(bn_a), "+s"(bn_b));
        f16x8 xf[K16][2];
        float pin = 1.0f;
        if constexpr (MW) {
            // ---- LayerNorm of the lane's half row -> B fragments
            float s = 0.0f;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[kb][e];
            s += __shfl_xor(s, 32, 64);
            const float mu = s / (float)K;
            float q = 0.0f;
#pragma unroll
            for (int kb = 0; kb < K16; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[kb][e] - mu; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 32, 64);
            const float rstd = 1.0f / sqrtf(q / (float)K + 1e-5f);
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                const float4 w0 = *reinterpret_cast<const float4*>(ln_w + 16 * kb + 8 * h), w1 = *reinterpret_cast<const float4*>(ln_w + 16 * kb + 8 * h + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(ln_b + 16 * kb + 8 * h), c1 = *reinterpret_cast<const float4*>(ln_b + 16 * kb + 8 * h + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[kb][e] = (v[kb][e] - mu) * rstd * w[e] + c[e];
            }
            pin = split_rows(xf, a.w1_un);
        }
        fetch(CMC(2));
        phase_end(std::true_type{});
        CM_STAMP(1)

        // one 32-output tile: Yt = W chunk . Xt
        auto tile = [&](const unsigned char* slot) {
            const unsigned char* wp = slot + lane * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            f16x8 nh = cm_ld(wp), nl = cm_ld(wp + 1024);
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                const f16x8 ch = nh, cl = nl;
                if (kb + 1 < K16) { nh = cm_ld(wp + (kb + 1) * 2048); nl = cm_ld(wp + (kb + 1) * 2048 + 1024); }
                cm_mfma3(ch, cl, xf[kb][0], xf[kb][1], acc);
            }
            return acc;
        };
        // ---- conv1 (pointwise) + GLU: chunks 2 blk (a tile) and 2 blk + 1 (b tile) -> G[row][32 blk ..]
        auto pw1_block = [&](auto bb) {
            constexpr int blk = decltype(bb)::value;
            f32x16 acc_a;
            fetch(CMC(2 * blk + 3));
            if constexpr (MW) {
                acc_a = tile(slot_of(CMC(2 * blk)));
                // scale and bias here: this chunk's slot is refilled (chunk 2 blk + 4) while the gate tile runs
                const float* bpa = reinterpret_cast<const float*>(slot_of(CMC(2 * blk)) + FR) + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bpa + 8 * g);
                    acc_a[4 * g] = acc_a[4 * g] * pin + b0.x; acc_a[4 * g + 1] = acc_a[4 * g + 1] * pin + b0.y;
                    acc_a[4 * g + 2] = acc_a[4 * g + 2] * pin + b0.z; acc_a[4 * g + 3] = acc_a[4 * g + 3] * pin + b0.w;
                }
            }
            phase_end(std::true_type{});
            fetch(CMC(2 * blk + 4));
            if constexpr (MW) {
                f32x16 acc_b = tile(slot_of(CMC(2 * blk + 1)));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_b[r] *= pin;
                const float* bpb = reinterpret_cast<const float*>(slot_of(CMC(2 * blk + 1)) + FR) + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * blk + 8 * g + 4 * h;
                    if (col < D) {                             // D % 4 == 0: the four features are in or out together
                        const float4 b1 = *reinterpret_cast<const float4*>(bpb + 8 * g);
                        float4 o = make_float4(acc_a[4 * g], acc_a[4 * g + 1], acc_a[4 * g + 2], acc_a[4 * g + 3]);
                        o.x *= cm_sigmoid(acc_b[4 * g] + b1.x);
                        o.y *= cm_sigmoid(acc_b[4 * g + 1] + b1.y);
                        o.z *= cm_sigmoid(acc_b[4 * g + 2] + b1.z);
                        o.w *= cm_sigmoid(acc_b[4 * g + 3] + b1.w);
                        *reinterpret_cast<float4*>(G + tok * GP + col) = o;
                    }
                }
            }
            phase_end(std::true_type{});
            CM_STAMP(2 + blk)
        };
        pw1_block(CMC(0)); pw1_block(CMC(1)); pw1_block(CMC(2)); pw1_block(CMC(3)); pw1_block(CMC(4));

        // ---- depthwise conv along time + folded BatchNorm + swish, in place in G: thread = (channel, third of the clip)
        {
            const int tid_o = tid;
            const int cp = tid_o % (D / 2), part = tid_o / (D / 2), c = 2 * cp;
            const bool on = tid_o < 3 * (D / 2);
            const int t0 = part * PR;
            float2 win[PR + KT - 1], wv[KT];
            float2 bs = make_float2(0.f, 0.f), al = bs, be = bs;
            if (on) {
#pragma unroll
                for (int k = 0; k < KT; ++k) wv[k] = *reinterpret_cast<const float2*>(dw_t + k * D + c);
                bs = *reinterpret_cast<const float2*>(dw_b + c); al = *reinterpret_cast<const float2*>(bn_a + c); be = *reinterpret_cast<const float2*>(bn_b + c);
#pragma unroll
                for (int j = 0; j < PR + KT - 1; ++j) {
                    const int tt = t0 - HALO + j;
                    win[j] = (tt >= 0 && tt < T) ? *reinterpret_cast<const float2*>(G + tt * GP + c) : make_float2(0.f, 0.f);
                }
            }
            __syncthreads();                                   // every window is in registers
            CM_STAMP(7)
            if (on) {
                // four outputs of both channels at a time: eight independent fmaf chains (a lone wave per SIMD has nothing else to issue while a
                // dependent chain waits for its own result); per output still ascending k
                constexpr int JB = 4;
#pragma unroll
                for (int j0 = 0; j0 < PR; j0 += JB) {
                    float ax[JB], ay[JB];
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) { ax[jj] = 0.0f; ay[jj] = 0.0f; }
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
#pragma unroll
                        for (int jj = 0; jj < JB; ++jj)
                            if (j0 + jj < PR) {
                                ax[jj] = fmaf(win[j0 + jj + k].x, wv[k].x, ax[jj]); ay[jj] = fmaf(win[j0 + jj + k].y, wv[k].y, ay[jj]);
                                asm volatile("" : "+v"(ax[jj]), "+v"(ay[jj]));     // pins the tap here (pure arithmetic floats freely otherwise: each output's taps end up back to back again)
                            }
                    }
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj)
                        if (j0 + jj < PR) {
                            const int j = j0 + jj;
                            const float zx = (ax[jj] + bs.x) * al.x + be.x, zy = (ay[jj] + bs.y) * al.y + be.y;
                            if (t0 + j < T) *reinterpret_cast<float2*>(G + (t0 + j) * GP + c) = make_float2(zx * cm_sigmoid(zx), zy * cm_sigmoid(zy));
                        }
                }
            }
        }
        __syncthreads();
        CM_STAMP(8)

        // ---- conv2 (pointwise) + residual
        float pin2 = 1.0f;
        constexpr int TP = 36;
        float* tb = G + (wave & 3) * 32 * TP;
        const int tq = lane & 7;
        const float* t_rrow[4];
        float* t_orow[4];
        bool t_ok[4];
        float4 rres[3][4];                                     // residual pieces of blocks blk, blk + 1, blk + 2 (requested two blocks ahead)
        auto load_res = [&](auto bb) {
            constexpr int blk = decltype(bb)::value;
            const int col = 32 * blk + 4 * tq;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rres[blk % 3][j] = (t_ok[j] && col < D) ? *reinterpret_cast<const float4*>(t_rrow[j] + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if constexpr (MW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rj = 32 * (wave & 3) + 8 * j + (lane >> 3);
                t_ok[j] = rj < T;
                const size_t rr = (size_t)clip * T + min(rj, T - 1);
                t_rrow[j] = a.h + rr * D;
                t_orow[j] = a.out + rr * D;
            }
            load_res(CMC(0)); load_res(CMC(1));
#pragma unroll
            for (int kb = 0; kb < K16; ++kb) {
                const float4 p0 = *reinterpret_cast<const float4*>(G + tokc * GP + 16 * kb + 8 * h);
                const float4 q0 = *reinterpret_cast<const float4*>(G + tokc * GP + 16 * kb + 8 * h + 4);
                v[kb][0] = p0.x; v[kb][1] = p0.y; v[kb][2] = p0.z; v[kb][3] = p0.w;
                v[kb][4] = q0.x; v[kb][5] = q0.y; v[kb][6] = q0.z; v[kb][7] = q0.w;
            }
            pin2 = split_rows(xf, a.w2_un);
        }
        __syncthreads();                                       // G is dead: its first 18 KB become the waves' transpose tiles
        CM_STAMP(9)
        auto pw2_block = [&](auto bb) {
            constexpr int blk = decltype(bb)::value, C = 2 * NB + blk;
            constexpr bool F = C + 3 != NCH;                   // (chunk NCH is the empty one)
            if constexpr (MW) {
                if constexpr (blk + 2 < NB) load_res(CMC(blk + 2));
                if constexpr (blk + 1 == NB) load_rows(min(clip + (int)gridDim.x, a.B - 1));     // the next clip's rows (unconditional: see above)
            }
            fetch(CMC(C + 3));
            if constexpr (MW) {
                f32x16 acc = tile(slot_of(CMC(C)));
                // (waited for before the stores: vmcnt counts stores too - lin_x3.hip; the chunk just requested may stay in flight)
                if constexpr (F) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STEPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= pin2;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(tb + n * TP + 8 * g + 4 * h) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                __builtin_amdgcn_wave_barrier();
                const int col = 32 * blk + 4 * tq;
                const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(slot_of(CMC(C)) + FR) + 4 * tq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 y = *reinterpret_cast<const float4*>(tb + (8 * j + (lane >> 3)) * TP + 4 * tq);
                    float4 o = make_float4(y.x + b0.x, y.y + b0.y, y.z + b0.z, y.w + b0.w);
                    const float4 r4 = rres[blk % 3][j];
                    o.x = r4.x + o.x; o.y = r4.y + o.y; o.z = r4.z + o.z; o.w = r4.w + o.w;
                    if (t_ok[j] && col < D) *reinterpret_cast<float4*>(t_orow[j] + col) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            CM_STAMP(10 + blk)
        };
        pw2_block(CMC(0)); pw2_block(CMC(1)); pw2_block(CMC(2)); pw2_block(CMC(3)); pw2_block(CMC(4));
    }
#undef CMC
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

bool convmod_x3_supported(int T, int D, int KT) { return D == 144 && KT == 31 && T >= 1 && T <= 128; }

hipError_t launch_convmod_taps(const float* w, float* wt, int D, int KT, hipStream_t s) {
    hipLaunchKernelGGL(cm_taps_kernel, dim3((unsigned)((D * KT + 255) / 256)), dim3(256), 0, s, w, wt, D, KT);
    return hipGetLastError();
}

hipError_t launch_convmod_x3(const ConvModArgs& a, int D, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (!convmod_x3_supported(a.T, D, 31)) return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(a.h) | reinterpret_cast<uintptr_t>(a.out)) & 15) != 0) return hipErrorInvalidValue;
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const dim3 grid(a.B < cus ? a.B : cus);
    if (a.T <= 102) hipLaunchKernelGGL((convmod_x3_kernel<9, 34>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((convmod_x3_kernel<9, 43>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
