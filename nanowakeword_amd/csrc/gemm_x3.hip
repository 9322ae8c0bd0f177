// gemm_x3.hip - float32 GEMM on the bf16 matrix cores by exact operand splitting (gfx950).
//
// C = post(A W^T) with A [M][K] and W [N][K] float32.  Every float32 x is written as hi + mid + lo, three bf16 values
// holding its 24 significant bits exactly (hi = x with the low 16 bits cleared, mid likewise of x - hi, lo the rest).
// A product x*w is then the sum of nine bf16 x bf16 products, each exact in float32; the six largest
//     hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid
// are issued as v_mfma_f32_32x32x16_bf16 with float32 accumulation; the three dropped ones are below 2^-23 of the
// product, i.e. the size of ONE float32 rounding, of which an fmaf chain over K commits K.  The result is float32
// arithmetic to within its own rounding noise (tests: <= 2e-6 relative to the v_mfma_f32_32x32x2_f32 path on every
// head) at 16/6 = 2.7x the float32 MFMA rate.  Determinism and batch invariance are unchanged: fixed K order per
// output, split-K chunks depend on (N, K) only.
//
// Tiling: workgroup = 4 waves stacked along M, BM = 128 rows x BN = 32*CB columns, BK = 32 (two MFMA K blocks).  LDS
// rows are [3 terms][32 k] bf16 + 16 B pad = 208 B, which makes the per-lane 16-byte fragment reads conflict-free; two
// workgroups share a CU; the next k-tile's global loads travel in registers during the multiplication.  A is split
// while it is staged into LDS (5.5 VALU ops per element, amortised over 32*CB columns); W is split once at
// nww_finalize into the same [N][K/16][3][16] layout so its tiles are plain 16-byte copies.
// H2 instances ("f16x3", GemmArgs::h2): the operands, scaled by powers of two fixed at plan time (a_scale on load, the weights
// at pack time; c_scale undoes both on the way out), are split into TWO binary16 terms hi = RN16(v), lo = RN16(v - hi) holding
// 22-23 of the 24 significant bits; hi*hi, hi*lo, lo*hi go to v_mfma_f32_32x32x16_f16 - half the matrix instructions, the
// float32 MFMA's accuracy against float64 (trunk_b.hip has the argument).  LDS rows are [2 terms][32 k] + 16 B pad = 144 B.
// Round 3, measured and not kept: an 8-wave version with two LDS stages, ONE barrier per k-tile, the split + store of tile
// k + 1 placed between the MFMAs of tile k and global loads three tiles ahead (0 spills, parity-green, bit-identical
// results) ran fc1 in 0.098 ms against 0.090 for this kernel.  The step draws 1350 W of the package's 1400 W cap
// (tools/power_probe.sh): at the cap a kernel's time follows the energy of its instructions, and that version read every
// A fragment from LDS twice (two waves per row block).
// Also measured and not kept, for the small-M instance (fc1 at B = 1, 24 us event to event): one wave per (32 rows, 32 columns,
// chunk) with no LDS and no barrier, loads four k-tiles ahead (27 us), and four waves sharing the staging of eight k-tiles per
// barrier with weight fragments loaded straight from the split layout (31 us) - both bit-identical.  The call is bound by
// streaming the 9.8 MB of split weights through the 64 sequential (column tile, split-K chunk) accumulation chains that
// batch invariance fixes, not by the per-tile staging round.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "split_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

#ifdef X3_TRACE      // tools/ubench/gemm_x3_trace.hip: s_memtime stamps of workgroup 0's first k-tiles (phase boundaries per wave)
__device__ unsigned long long x3_trace_buf[4 * 16 * 4];
#define X3_STAMP(kt_rel, ph)                                                                                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (kt_rel) < 16 && (threadIdx.x & 63) == 0)        \
        x3_trace_buf[((threadIdx.x >> 6) * 16 + (kt_rel)) * 4 + (ph)] = __builtin_readcyclecounter();
#else
#define X3_STAMP(kt_rel, ph)
#endif
namespace {
constexpr int X3_BM = 128;
// bytes per LDS row: NT terms x 32 k x 2 bytes + 16 pad (an odd number of 16-byte slots: conflict-free b128 reads)
template <bool H2> struct X3A { static constexpr int NT = H2 ? 2 : 3, ROW = 64 * NT + 16; };

// 1 / (1 + e^-v) on the hardware exp2 and reciprocal (1 ulp each; relative error <= 3e-7): 4 instructions where
// expf + IEEE division take ~35 - the epilogue of a short-K GEMM is as long as its main loop otherwise.
__device__ __forceinline__ float x3_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
template <int ACT>
__device__ __forceinline__ float x3_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return nww_gelu(v);
    if (ACT == ACT_SILU) return v * x3_sigmoid(v);
    if (ACT == ACT_SIGMOID) return x3_sigmoid(v);
    return v;
}

// x -> three float32 bit patterns whose upper halves are the bf16 terms (lo is truncated when packed; it has at
// most 8 significant bits left, so nothing is lost)
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
// (a >> 16) | (b & 0xffff0000): bf16 of a in the low half (element k), of b in the high half (element k+1)
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// the partial products of one 16-k block, small terms first: six of bf16 terms (a, w = hi / mid / lo) or three of binary16 terms
template <bool H2>
__device__ __forceinline__ void x3_products(const uint4* a, const uint4* w, f32x16& acc) {
    if (H2) {
        const f16x8 ah = __builtin_bit_cast(f16x8, a[0]), al = __builtin_bit_cast(f16x8, a[1]);
        const f16x8 wh = __builtin_bit_cast(f16x8, w[0]), wl = __builtin_bit_cast(f16x8, w[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
    } else {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]), al = __builtin_bit_cast(bf16x8, a[H2 ? 1 : 2]);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, w[0]), wm = __builtin_bit_cast(bf16x8, w[1]), wl = __builtin_bit_cast(bf16x8, w[H2 ? 1 : 2]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
    }
}
}  // namespace

// W [N][K] float32 -> [N][KB][3][16] bf16, KB = 2 ceil(K/32) (whole 32-k tiles: the kernels load without a bound check), zero beyond K
__global__ void __launch_bounds__(256) split_weights_x3_kernel(const float* __restrict__ W, uint16_t* __restrict__ out,
                                                               int N, int K, int KB) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // one (n, k) element
    const size_t total = (size_t)N * KB * 16;
    if (idx >= total) return;
    const int kk = (int)(idx & 15);
    const size_t nb = idx >> 4;
    const int kb = (int)(nb % KB), n = (int)(nb / KB);
    const int k = kb * 16 + kk;
    const float x = k < K ? W[(size_t)n * K + k] : 0.0f;
    uint32_t hi, mid, lo;
    split3(x, hi, mid, lo);
    uint16_t* o = out + nb * 48 + kk;
    o[0] = (uint16_t)(hi >> 16); o[16] = (uint16_t)(mid >> 16); o[32] = (uint16_t)(lo >> 16);
}

// W [N][K] float32 -> [N][KB][2][16] binary16 terms of W * scale, zero beyond K
__global__ void __launch_bounds__(256) split_weights_h2_kernel(const float* __restrict__ W, uint16_t* __restrict__ out,
                                                               int N, int K, int KB, float scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // one (n, k) element
    const size_t total = (size_t)N * KB * 16;
    if (idx >= total) return;
    const int kk = (int)(idx & 15);
    const size_t nb = idx >> 4;
    const int kb = (int)(nb % KB), n = (int)(nb / KB);
    const int k = kb * 16 + kk;
    const float x = k < K ? W[(size_t)n * K + k] * scale : 0.0f;
    uint32_t hi, lo;
    nww_split2h(x, 0.0f, hi, lo);
    uint16_t* o = out + nb * 32 + kk;
    o[0] = (uint16_t)hi; o[16] = (uint16_t)lo;
}

// NST = register stages of global loads in flight (k-tiles of lookahead).  The bulk shapes use 1 (two workgroups per
// CU cover each other); the small-M instance <1, 8> (M <= 64: the interpreter's B = 1 .. 16 calls) is pure load
// latency - 25 dependent k-tiles of ~2.4 us each - and runs 32-column tiles (4x the workgroups) with eight k-tiles in
// flight (only rows 0..63 of A are staged, so a stage is 16 registers).  Per-output summation order is the same in every instance: same k-tile sequence, same products, same
// split-K chunks - results stay bit-identical across batch sizes.
template <int CB, int NST, int ACT, bool H2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(184))) gemm_x3_kernel(GemmArgs g) {
    constexpr int NT = X3A<H2>::NT, X3_ROW = X3A<H2>::ROW, PB = 2 * NT;   // terms, LDS row bytes, 16-byte pieces of a row's 16-k block
    // accumulators in AGPRs: two workgroups share a CU, and one's bf16 MFMAs run beside the other's split/stage VALU
    // work only in the AGPR form (tools/ubench/mfma_valu_overlap.hip).  The empty asm flips hipcc's
    // choice; the plain launch bound keeps the register file unsplit, amdgpu_num_vgpr caps the VGPR side so that
    // VGPRs + accumulator AGPRs <= 256 (two waves per SIMD).
    { float agpr_hint = 0.0f; asm volatile("; mfma accumulators in AGPRs" : "+a"(agpr_hint)); }
    constexpr int BN = 32 * CB;
    constexpr int WPIECES = BN * 2 * PB;                       // 16-byte pieces of a W tile (32 k x NT terms per row)
    constexpr int WLD = (WPIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto As = [&](int) { return smem; };
    auto Ws = [&](int) { return smem + X3_BM * X3_ROW; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    // (an XCD-aware tile map - every XCD keeping a few W chunks in its L2 - measured no change: rocprofv3 FETCH_SIZE of fc1 is A's
    // 211 MB already, the split W is served from L2 as it is)
    const int bm = bxi * X3_BM, bn = byi * BN;
    const int KT = (g.K + 31) >> 5, KB = 2 * KT;               // 32-k tiles, 16-k blocks of the split weights (zero padded to whole tiles)
    int kt_begin = 0, kt_end = KT;
    if (g.splitk > 1) {
        const int kc = (KT + g.splitk - 1) / g.splitk;
        kt_begin = bzi * kc;
        kt_end = min(KT, kt_begin + kc);
    }
    const uint4* Wx = reinterpret_cast<const uint4*>(g.Wx3);

    // A loader: thread -> rows (tid>>3) + 32q, floats 4*(tid&7) .. +3 of the 32-k tile
    const int lr = tid >> 3, lq = tid & 7;
    const float* arow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        arow[q] = g.a_blocked ? g.A + ((size_t)(bm >> 7) * g.a_blocked * 128 + lr + 32 * q) * 32 + 4 * lq
                              : g.A + (size_t)min(bm + lr + 32 * q, g.M - 1) * g.lda + 4 * lq;
    const int a_kstep = g.a_blocked ? 128 * 32 : 32;           // floats between consecutive k-tiles of a row
    constexpr int AQ = (NST > 1 && CB == 1) ? 2 : 4;           // the small-M instance (M <= 64) stages rows 0..63 only
    struct Stage { f32x4v a[AQ]; u32x4v w[WLD]; };
    // Unconditional loads: the split weights are zero padded to whole k-tiles, and a float4 of A beyond K (K % 4 == 0:
    // gemm_x3_usable) is fetched from the row's first tile instead - finite values that meet those zeros.  Loads under exec-mask
    // branches (bound checks, or a select that hipcc turns back into a branch) made its wait-count pass put s_waitcnt vmcnt(0)
    // behind every group of them: no load of a later stage stayed in flight across a k-tile, NST > 1 bought nothing.
    auto gload = [&](int kt, Stage& st) {
        const int k = kt * 32 + 4 * lq;
        const size_t a_off = k + 4 <= g.K ? (size_t)kt * a_kstep : 0;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            st.a[q] = *reinterpret_cast<const f32x4v*>(arow[q] + a_off);
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int p = tid + 256 * j;
            if (WPIECES % 256 == 0 || p < WPIECES) {
                const int row = p / (2 * PB), c = p - row * (2 * PB);
                const int n = min(bn + row, g.N - 1);
                const int kb = 2 * kt + c / PB;                // 16-k block of this piece
                st.w[j] = *reinterpret_cast<const u32x4v*>(Wx + ((size_t)n * KB + kb) * PB + (c % PB));
            }
        }
    };
    const float a_scale = H2 ? g.a_scale : 1.0f;
    const float a_lim = H2 ? (g.a_clamp > 0.0f ? g.a_clamp * a_scale : 65504.0f) : 0.0f;      // scaled clamp (65504: no clamp asked - saturate instead of inf)
    auto lstore = [&](int buf, const Stage& st) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            unsigned char* d = As(buf) + (lr + 32 * q) * X3_ROW + 8 * lq;
            if (H2) {
                uint32_t h0, l0, h1, l1;
                nww_split2h(__builtin_amdgcn_fmed3f(st.a[q][0] * a_scale, -a_lim, a_lim), __builtin_amdgcn_fmed3f(st.a[q][1] * a_scale, -a_lim, a_lim), h0, l0);
                nww_split2h(__builtin_amdgcn_fmed3f(st.a[q][2] * a_scale, -a_lim, a_lim), __builtin_amdgcn_fmed3f(st.a[q][3] * a_scale, -a_lim, a_lim), h1, l1);
                *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(d + 64) = make_uint2(l0, l1);
            } else {
                uint32_t hi[4], mid[4], lo[4];
                split3(st.a[q][0], hi[0], mid[0], lo[0]); split3(st.a[q][1], hi[1], mid[1], lo[1]);
                split3(st.a[q][2], hi[2], mid[2], lo[2]); split3(st.a[q][3], hi[3], mid[3], lo[3]);
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]));
                *reinterpret_cast<uint2*>(d + 64) = make_uint2(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]));
                *reinterpret_cast<uint2*>(d + 128) = make_uint2(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]));
            }
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int p = tid + 256 * j;
            if (WPIECES % 256 == 0 || p < WPIECES) {
                const int row = p / (2 * PB), c = p - row * (2 * PB);
                const int cp = c % PB;                         // (term, half) inside the 16-k block
                *reinterpret_cast<u32x4v*>(Ws(buf) + row * X3_ROW + (cp >> 1) * 64 + (c / PB) * 32 + (cp & 1) * 16) = st.w[j];
            }
        }
    };

    f32x16 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    // One LDS stage per workgroup (53 KB for BN = 128, so two workgroups share a CU and cover each other's staging
    // phases); global loads run NST k-tiles ahead in NST register stages.
    Stage st[NST];
    // every load is issued unconditionally (a k-tile index beyond the chunk repeats its last tile and is never multiplied): loads
    // under "is there another tile" / bound-check branches get an s_waitcnt vmcnt(0) behind every group from hipcc's wait-count
    // pass, and the next tile's loads then do not stay in flight under this tile's products (fc1: 0.096 -> 0.064 ms).
    // (Measured and not kept: more than one register stage in flight for the bulk instances.  hipcc drains ALL stages at the loop
    // head - 0.082 ms with the old branchy loads, no better than one stage with these; inline-asm loads with hand-counted vmcnt
    // are not safe here: the allocator re-uses a destination register of a load it believes complete.)
#pragma unroll
    for (int q = 0; q < NST; ++q) gload(min(kt_begin + q, kt_end - 1), st[q]);
    const bool has_rows = bm + wave * 32 < g.M;                  // waves whose 32 rows lie beyond M only help with the staging
    const int a_off = (wave * 32 + i) * X3_ROW + 16 * h, w_off = i * X3_ROW + 16 * h;
    auto multiply = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned char* ap = As(0) + a_off + 32 * kk;
            uint4 af[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) af[t] = *reinterpret_cast<const uint4*>(ap + 64 * t);
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const unsigned char* wp = Ws(0) + w_off + c * 32 * X3_ROW + 32 * kk;
                uint4 wf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(wp + 64 * t);
                x3_products<H2>(af, wf, acc[c]);               // small terms first, the dominant hi*hi last
            }
        }
    };
    for (int kt = kt_begin; kt < kt_end; kt += NST) {
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            if (kt + q < kt_end) {
                X3_STAMP(kt + q - kt_begin, 0)
                __syncthreads();                               // everyone is done reading the previous tile
                X3_STAMP(kt + q - kt_begin, 1)
                lstore(0, st[q]);
                __syncthreads();
                X3_STAMP(kt + q - kt_begin, 2)
            }
            gload(min(kt + q + NST, kt_end - 1), st[q]);
            if (kt + q < kt_end && (NST == 1 || has_rows)) multiply();
            X3_STAMP(kt + q - kt_begin, 3)
        }
    }

    const int m0 = bm + wave * 32;
    if (m0 >= g.M) return;
    if (H2) {                     // back to the true scale (a power of two: every rounding behind this is the unscaled one's)
        const float c_scale = g.c_scale;
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] *= c_scale;
    }
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int n = bn + c * 32 + i;
        if (n >= g.N) continue;
        if (g.splitk > 1) {
            float* part = g.splitk_ws + (size_t)bzi * g.M * g.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) part[(size_t)m * g.N + n] = acc[c][r];
            }
            continue;
        }
        const float bias = g.bias ? g.bias[n] : 0.0f;
        const float al2 = g.alpha ? g.alpha[n] : 1.0f, be2 = g.alpha ? g.beta[n] : 0.0f;
        // lane (column n, row half h): rows m0 + (r & 3) + 8 (r >> 2) + 4 h.  Uniform choices (residual, full tile) are made
        // once, outside the sixteen-row loops; the activation is a template parameter (the run-time switch, per-element
        // bounds checks and the precise expf / division of the first version made the epilogue of a K = 144 GEMM as
        // long as its main loop: 103 M -> 54 M VALU instructions on the Conformer's linear1).
        float* cp = g.C + (size_t)(m0 + 4 * h) * g.ldc + n;
        const float* rp = g.res ? g.res + (size_t)(m0 + 4 * h) * g.ldres + n : nullptr;
        const bool full = m0 + 32 <= g.M;
        auto finish = [&](float a) {
            float v = a + bias;
            v = v * al2 + be2;                 // alpha = 1, beta = 0 without a folded BatchNorm: exact
            return x3_act<ACT>(v);
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

// Small M with split-K (fc1 / layer1 of an interpreter call: M = 1 .. 64, K >= 4096): the call is one round trip to memory
// plus ONE dependent MFMA chain per (32 columns, split-K chunk) - batch invariance fixes the chunk and the order of the
// 6 x 2 products of each of its k-tiles, so a chunk of 25 k-tiles costs 300 x 32 matrix-pipe clocks whatever else happens.
// gemm_x3_kernel<1, 8> spends ~1500 clocks per k-tile on that (stage, barrier, multiply, barrier; eight k-tiles of loads
// in flight).  Here the eight waves of a workgroup put the WHOLE chunk in flight at once - every wave loads the weight
// fragments (16-byte pieces of the split layout, straight into MFMA operand registers) and the A rows of its own
// CH_TPW k-tiles - and then the accumulator walks through the waves in k order, handed on
// through 4 KB of LDS: no staging, one barrier per CH_TPW k-tiles.  Same products, same order, same chunks as every
// other instance: results are bit-identical.
constexpr int CH_NW = 8, CH_TPW = 4;
template <bool H2>
__global__ void __launch_bounds__(64 * CH_NW) gemm_x3_chain_kernel(GemmArgs g) {
    constexpr int NT = X3A<H2>::NT, PB = 2 * NT;
    const float a_scale = H2 ? g.a_scale : 1.0f, c_scale = H2 ? g.c_scale : 1.0f;
    const float a_lim = H2 ? (g.a_clamp > 0.0f ? g.a_clamp * a_scale : 65504.0f) : 0.0f;
    __shared__ __attribute__((aligned(16))) float4 hand[4 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, h = lane >> 5;
    const int bm = blockIdx.x * 32, bn = blockIdx.y * 32;
    const int KT = (g.K + 31) >> 5, KB = 2 * KT;
    const int kc = (KT + g.splitk - 1) / g.splitk;
    const int kt_begin = blockIdx.z * kc, kt_end = min(KT, kt_begin + kc);
    const int mrow = min(bm + i, g.M - 1);
    const uint4* wrow = reinterpret_cast<const uint4*>(g.Wx3) + (size_t)min(bn + i, g.N - 1) * KB * PB + h;
    const float* abase = g.a_blocked ? g.A + ((size_t)(mrow >> 7) * g.a_blocked * 128 + (mrow & 127)) * 32 + 8 * h
                                     : g.A + (size_t)mrow * g.lda + 8 * h;
    const size_t a_kstep = g.a_blocked ? 128 * 32 : 32;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (kt_begin >= kt_end) {             // an empty split-K chunk (no split rule produces one today): its partial sums are zeros, not stale workspace
        if (wave == 0 && bn + i < g.N) {
            float* part = g.splitk_ws + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) part[(size_t)m * g.N + bn + i] = 0.0f;
            }
        }
        return;
    }
    for (int round = kt_begin; round < kt_end; round += CH_NW * CH_TPW) {
        const int t0 = round + wave * CH_TPW;
        const int nt = max(0, min(CH_TPW, kt_end - t0));                      // this wave's k-tiles (wave-uniform)
        uint4 wf[CH_TPW][2][NT];
        float4 ar[CH_TPW][2][2];
        // straight-line loads: k-tiles beyond the wave's share repeat its last valid one (never multiplied)
#pragma unroll
        for (int t = 0; t < CH_TPW; ++t) {
            const int kt = min(t0 + t, kt_end - 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bool in_k = kt * 32 + 16 * kk + 8 * h < g.K;            // K % 8 == 0: a lane's eight k are all inside or all outside
                const float* ap = in_k ? abase + (size_t)kt * a_kstep + 16 * kk : g.A;
                ar[t][kk][0] = *reinterpret_cast<const float4*>(ap);
                ar[t][kk][1] = *reinterpret_cast<const float4*>(ap + 4);
                if (!in_k) { ar[t][kk][0] = make_float4(0.f, 0.f, 0.f, 0.f); ar[t][kk][1] = ar[t][kk][0]; }
                const int kb = 2 * kt + kk;
#pragma unroll
                for (int term = 0; term < NT; ++term) wf[t][kk][term] = wrow[(size_t)kb * PB + 2 * term];
            }
        }
        const int active = min(CH_NW, (kt_end - round + CH_TPW - 1) / CH_TPW);    // waves that hold k-tiles of this round
        const bool last_round = round + CH_NW * CH_TPW >= kt_end;
        for (int s = 0; s < active; ++s) {
            if (wave == s) {
                if (s > 0 || round > kt_begin) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = hand[q * 64 + lane];
                        acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
                    }
                }
#pragma unroll
                for (int t = 0; t < CH_TPW; ++t) {
                    if (t >= nt) continue;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        // A is split here, one k-block at a time, in the shadow of the previous block's MFMAs (all of A's
                        // fragments beside all of W's would not fit the register file)
                        const float x[8] = {ar[t][kk][0].x, ar[t][kk][0].y, ar[t][kk][0].z, ar[t][kk][0].w,
                                            ar[t][kk][1].x, ar[t][kk][1].y, ar[t][kk][1].z, ar[t][kk][1].w};
                        uint4 af[NT];
                        if (H2) {
                            uint32_t hh[4], ll[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                nww_split2h(__builtin_amdgcn_fmed3f(x[2 * j] * a_scale, -a_lim, a_lim), __builtin_amdgcn_fmed3f(x[2 * j + 1] * a_scale, -a_lim, a_lim), hh[j], ll[j]);
                            af[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                            af[1] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                        } else {
                            uint32_t hi[8], mid[8], lo[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) split3(x[j], hi[j], mid[j], lo[j]);
                            af[0] = make_uint4(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]), pack_hi16(hi[4], hi[5]), pack_hi16(hi[6], hi[7]));
                            af[1] = make_uint4(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]), pack_hi16(mid[4], mid[5]), pack_hi16(mid[6], mid[7]));
                            af[NT - 1] = make_uint4(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]), pack_hi16(lo[4], lo[5]), pack_hi16(lo[6], lo[7]));
                        }
                        x3_products<H2>(af, wf[t][kk], acc);                          // gemm_x3_kernel's order
                    }
                }
                if (last_round && s == active - 1) {                          // the chunk's partial sums
                    const int n = bn + i;
                    float* part = g.splitk_ws + (size_t)blockIdx.z * g.M * g.N;
                    if (n < g.N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = bm + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (m < g.M) part[(size_t)m * g.N + n] = H2 ? acc[r] * c_scale : acc[r];
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hand[q * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
            __syncthreads();
        }
    }
}

size_t gemm_x3_weight_bytes(int N, int K) { return (size_t)N * (2 * ((K + 31) / 32)) * 96; }      // (the two-term image needs 64 of the 96)

hipError_t launch_split_weights_h2(const float* W, void* out, int N, int K, float scale, hipStream_t s) {
    const int KB = 2 * ((K + 31) / 32);
    const size_t total = (size_t)N * KB * 16;
    hipLaunchKernelGGL(split_weights_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W,
                       reinterpret_cast<uint16_t*>(out), N, K, KB, scale);
    return hipGetLastError();
}

hipError_t launch_split_weights_x3(const float* W, void* out, int N, int K, hipStream_t s) {
    const int KB = 2 * ((K + 31) / 32);
    const size_t total = (size_t)N * KB * 16;
    hipLaunchKernelGGL(split_weights_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W,
                       reinterpret_cast<uint16_t*>(out), N, K, KB);
    return hipGetLastError();
}

// column blocks per workgroup: least padded width first, then the widest tile (A is re-read once per column tile)
static int x3_pick_cb(int N) {
    int best = 2;
    long best_cost = -1;
    for (int cb = 2; cb <= 6; ++cb) {
        const long tiles = (N + 32 * cb - 1) / (32 * cb);
        const long cost = tiles * 32 * cb * 8 - cb;            // padded width dominates, wider tile breaks ties
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cb; }
    }
    return best;
}

bool gemm_x3_usable(const GemmArgs& g) {
    return g.Wx3 != nullptr && g.N >= 32 && g.K >= 32 && (g.K % 4 == 0) && (g.lda % 4 == 0) && (!g.a_blocked || g.K % 32 == 0) &&
           ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
}

hipError_t launch_gemm_x3(const GemmArgs& g, hipStream_t s) {
    const int sk = (g.splitk > 1 && g.splitk_ws) ? g.splitk : 1;
    GemmArgs a = g;
    a.splitk = sk;
    const bool h2 = a.h2 != 0;
    const size_t row_b = h2 ? X3A<true>::ROW : X3A<false>::ROW;
#define X3_GO1(CBV, NSTV, ACTV, H2V, GRID, LDSB)                                                                   \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(gemm_x3_kernel<CBV, NSTV, ACTV, H2V>), LDSB);   \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((gemm_x3_kernel<CBV, NSTV, ACTV, H2V>), GRID, dim3(256), LDSB, s, a);                   \
    }
#define X3_GO(CBV, NSTV, ACTV, GRID, LDSB)                                                                         \
    if (h2) X3_GO1(CBV, NSTV, ACTV, true, GRID, LDSB) else X3_GO1(CBV, NSTV, ACTV, false, GRID, LDSB)
#define X3_ACT(CBV, NSTV, GRID, LDSB)                                                                              \
    switch (sk > 1 ? (int)ACT_NONE : a.act) {      /* split-K partials take no epilogue: one instance serves them all */ \
        case ACT_RELU: X3_GO(CBV, NSTV, ACT_RELU, GRID, LDSB) break;                                               \
        case ACT_GELU: X3_GO(CBV, NSTV, ACT_GELU, GRID, LDSB) break;                                               \
        case ACT_SILU: X3_GO(CBV, NSTV, ACT_SILU, GRID, LDSB) break;                                               \
        case ACT_SIGMOID: X3_GO(CBV, NSTV, ACT_SIGMOID, GRID, LDSB) break;                                         \
        default: X3_GO(CBV, NSTV, ACT_NONE, GRID, LDSB) break;                                                     \
    }
    if (g.M <= 64 && sk > 1 && g.K % 8 == 0) {                                 // one memory round trip + one MFMA chain per chunk
        if (h2) hipLaunchKernelGGL(gemm_x3_chain_kernel<true>, dim3((g.M + 31) / 32, (g.N + 31) / 32, sk), dim3(64 * CH_NW), 0, s, a);
        else hipLaunchKernelGGL(gemm_x3_chain_kernel<false>, dim3((g.M + 31) / 32, (g.N + 31) / 32, sk), dim3(64 * CH_NW), 0, s, a);
        return hipGetLastError();
    }
    if (g.M <= 64) {                                           // small batches: latency, not throughput (see the kernel comment)
        dim3 grid1(1, (g.N + 31) / 32, sk);
        const size_t lds1 = (size_t)(X3_BM + 32) * row_b;
        X3_ACT(1, 8, grid1, lds1)
        return hipGetLastError();
    }
    int cb = x3_pick_cb(g.N);
    {   // a mid-sized M leaves CUs idle with the widest tile (CRNN input projection at 1024 streams: 64 row tiles x 2 column
        // tiles): take the widest tile of the same padded width that still gives every CU a workgroup, else the narrowest
        const long mt = (g.M + X3_BM - 1) / X3_BM, pad = (long)((g.N + 32 * cb - 1) / (32 * cb)) * 32 * cb;
        if (mt * ((g.N + 32 * cb - 1) / (32 * cb)) * sk < 256) {
            int pick = cb;
            for (int c = cb - 1; c >= 2; --c) {
                const long tiles = (g.N + 32 * c - 1) / (32 * c);
                if (tiles * 32 * c != pad) continue;
                pick = c;
                if (mt * tiles * sk >= 256) break;
            }
            cb = pick;
        }
    }
    const int bn = 32 * cb;
    dim3 grid((g.M + X3_BM - 1) / X3_BM, (g.N + bn - 1) / bn, sk);
    const size_t lds = (size_t)(X3_BM + bn) * row_b;
    switch (cb) {
        case 2: X3_ACT(2, 1, grid, lds) break;
        case 3: X3_ACT(3, 1, grid, lds) break;
        case 4: X3_ACT(4, 1, grid, lds) break;
        case 5: X3_ACT(5, 1, grid, lds) break;
        case 6: X3_ACT(6, 1, grid, lds) break;
        default: return hipErrorInvalidValue;
    }
#undef X3_ACT
#undef X3_GO
#undef X3_GO1
    return hipGetLastError();
}
