// conv3_x3.hip - Conv2d(32, Cout, 3, padding 1) (+ folded BN) + act (+ MaxPool2 | export-form AvgPool) on the bf16 matrix
// cores by exact operand splitting: the third conv stage of the E2E mel-CNN (32 -> 64, un-pooled, avg-pool fused;
// nanowakeword/modules/architectures.py:851-853) and of CRNN (32 -> 32, pooled; :217-225).
//
// Same arithmetic as conv2 of the fused trunk (trunk_b.hip): every float32 value is hi + mid + lo, three bf16 numbers
// holding its 24 significant bits exactly; the six largest of the nine bf16 x bf16 partial products go to
// v_mfma_f32_32x32x16_bf16 with float32 accumulation (products < 2^-23 of the result are dropped).  The float32-MFMA
// kernel this replaces (conv3x3_mfma_kernel, trunk.hip) spends 144 x 64 = 9216 matrix-pipe clocks per 32 pixel x 32
// channel tile, this one 108 x 32 = 3456 - and on gfx950 the float32 MFMA runs at the VALU rate and blocks the VALU
// while it runs, the bf16 one does neither.
//
// One workgroup (8 waves) = one (clip, 32-channel output group); a workgroup keeps its group for its whole life, so
// the group's weights are split and laid out as MFMA B fragments in LDS once:
//   Wt [tap 9][k-block 2][term 3][k-half 2][cout 32][8 cin] bf16 = 54 KB   (one 16-byte fragment per lane, consecutive)
//   A3 [H + 3][W + 2] pixels x 208 B ([term 3][32 cin] bf16 + 16 B pad: conflict-free 16-byte fragment reads), zero halo
// Per clip the 32 input planes are read once from HBM (coalesced), split, and stored channels-last into A3; a tile is
// 2 rows x 16 columns (the accumulator's 4-register groups are 2x2 windows, as in the trunk); per tap and 16-channel
// block a wave reads three fragments per tile and three weight fragments, issued one step ahead of the MFMAs.
// The next clip's planes are fetched into registers under the MFMA loop (one pixel x 32 channels per thread).
//
// Measured (E2E mel-CNN, 16 x 25 x 32 -> 64 + avg-pool, B = 4096): 0.83 ms (float32 MFMA) -> 0.415 ms; B = 1: 63 -> 21 us.
// Ablation of the 0.415: no MFMA loop 0.07, no epilogue 0.35, no LDS re-reads 0.40 - the loop runs at ~68 % of the
// nominal bf16 matrix rate (0.184 ms for the 6 x 2 x 9 x 32 x 64 x 400 multiply-adds per clip) and LDS bandwidth is
// not the limit; LDS capacity is (A3 107 KB + Wt 54 KB = one workgroup per CU, so staging / epilogue do not overlap).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
// H2 instances ("f16x3", ConvMfmaArgs::h2_in > 0): the planes times h2_in and the weights times h2_w (powers of two fixed at plan
// time from bounds on the tensors) are split into TWO binary16 terms (split_h2.h), three partial products per operand pair go
// to v_mfma_f32_32x32x16_f16 and the sums are scaled back before the epilogue: half the MFMAs, 144-byte pixels, 36 KB of weights.
#include "layers.h"
#include "trunk.h"
#include "conv_tile_epilogue.h"
#include "split_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef NWW_TRACE      // tools/ubench/conv3_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of its first 16 items
__device__ unsigned long long g_c3_trace[8 * 16 * 8];
#define C3_STAMP(k) if (blockIdx.x == 0 && c3_item < 16 && lane == 0) g_c3_trace[(wave * 16 + c3_item) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define C3_STAMP(k)
#endif

namespace {
// C1 input channels (32; 64 in the two-term form only): TS bytes per term of a pixel, KB sixteen-channel k-blocks
template <bool H2, int C1 = 32> struct C3A { static constexpr int NT = H2 ? 2 : 3, TS = 2 * C1, PS3 = TS * NT + 16, KB = C1 / 16, WT_BYTES = 9 * KB * NT * 1024; };
constexpr int PS3_MAX = 208, WT_BYTES_MAX = 9 * 2 * 3 * 1024;
// A3 row pitch.  A 16-lane group of a fragment read covers 2 rows x 8 pixels; with 208-byte pixels the 8 pixels of a row take the
// 16-byte slots {0, 13, 10, 7, 4, 1, 14, 11} of the 256-byte bank row, so the second row must sit 8 slots (128 bytes mod 256) away to
// take the other eight.  (W + 2) * 208 is 160 mod 256 for the 16-wide planes of the E2E / CRNN heads: both rows on the same slots,
// 36-44 % of the kernel's LDS cycles were conflict cycles (profiles/r03_pmc_all_configs.csv).  The pitch is padded - or, by up to
// 32 bytes, SHORTENED: the tail of the right halo pixel then overlaps the head of the next row's left halo pixel, both zero for ever.
// Planes that would no longer fit the CU's LDS with the padding keep the dense pitch.
// (144-byte pixels of the two-term form take the slots {0, 9, 6, 15, 10, 3, 12, 5}: the same rule)
// (272-byte pixels of the 64-channel two-term form take the slots {0 .. 7}: the same rule)
static int conv3_row_pitch(int H, int W, int avg_ow, bool h2, int C1 = 32) {
    const int NT = h2 ? 2 : 3, PS3 = 2 * C1 * NT + 16, WT_BYTES = 9 * (C1 / 16) * NT * 1024;
    const int dense = (W + 2) * PS3;
    static const int padded = 1;      // 0: dense rows (round 3), for A/B runs
    if (!padded) return dense;
    int pad = (128 - dense % 256 + 256) % 256;               // multiple of 16
    if (pad > 128 && pad - 256 >= -32) pad -= 256;
    const int npix = (H + 3) * (W + 1);
    const long extra = (avg_ow <= 0 || npix >= 512) ? 0 : (long)(512 - npix) * 16;
    if ((long)(H + 3) * (dense + pad) + 48 + WT_BYTES + extra > 160 * 1024) return dense;
    return dense + pad;
}

__device__ __forceinline__ void split3c(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}

// STRIP: planes of more than 512 pixels (a second stage at 50 x 32, the third stage of clips longer than ~1.3 s) go through LDS in strips of
// a.strip_h (even) rows plus their halo rows; a work item is (clip, strip), the halo rows are staged like the others (zeros outside the plane)
template <int ACT, bool POOL, bool AVG, bool H2, int C1 = 32, bool STRIP = false>
__global__ void __launch_bounds__(512) conv3_x3_kernel(ConvMfmaArgs a) {
    static_assert(!STRIP || (!AVG && C1 == 32), "strips: pooled or raw planes of the 32-channel instances");
    constexpr int NW = 8, NTHR = 512;
    constexpr int NT = C3A<H2, C1>::NT, PS3 = C3A<H2, C1>::PS3, WT_BYTES = C3A<H2, C1>::WT_BYTES, TS = C3A<H2, C1>::TS, KB = C3A<H2, C1>::KB;
    const float s_in = H2 ? a.h2_in : 1.0f, s_w = H2 ? a.h2_w : 1.0f, c_out = H2 ? 1.0f / (a.h2_in * a.h2_w) : 1.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int H = a.H, W = a.W, Wp = W + 2;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const int rowB = a.row_pitch;
    const int SH = STRIP ? a.strip_h : H, nS = STRIP ? (H + SH - 1) / SH : 1;      // rows per strip, strips per plane
    const int a3_bytes = (SH + 3) * rowB + 32;                   // (+ 32: the last row's right halo pixel when the pitch is shortened)
    unsigned char* A3 = lds3;
    unsigned char* Wt = lds3 + ((a3_bytes + 15) & ~15);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hi = lane >> 5;
    const int ngroups = a.Cout / 32;
    const int grp = (int)blockIdx.x % ngroups;                 // fixed for the workgroup's life
    // ---- zero A3 (halo stays zero), split this group's weights into B fragments
    for (int k = tid; k < a3_bytes / 4; k += NTHR) reinterpret_cast<uint32_t*>(A3)[k] = 0u;
    for (int idx = tid; idx < 32 * C1 * 9; idx += NTHR) {      // (cout, cin, tap)
        const int co = idx / (C1 * 9), r = idx - co * (C1 * 9), ci = r / 9, tap = r - ci * 9;
        const float wv = a.w[((size_t)(32 * grp + co) * a.w_cin + a.w_cin_off + ci) * 9 + tap];
        const int kb = ci >> 4, kh = (ci >> 3) & 1, e = ci & 7;
        unsigned char* d = Wt + ((tap * KB + kb) * NT) * 1024 + (kh * 32 + co) * 16 + e * 2;
        if (H2) {
            uint32_t th, tl;
            nww_split2h(wv * s_w, 0.0f, th, tl);
            *reinterpret_cast<uint16_t*>(d) = (uint16_t)th;
            *reinterpret_cast<uint16_t*>(d + 1024) = (uint16_t)tl;
        } else {
            uint32_t th, tm, tl;
            split3c(wv, th, tm, tl);
            *reinterpret_cast<uint16_t*>(d) = (uint16_t)(th >> 16);
            *reinterpret_cast<uint16_t*>(d + 1024) = (uint16_t)(tm >> 16);
            *reinterpret_cast<uint16_t*>(d + 2 * 1024) = (uint16_t)(tl >> 16);
        }
    }
    const int seq_ch = (POOL && a.seq_out) ? a.Cout * Ho : 0;
    const int cout = 32 * grp + i;
    const float bias = a.bias ? a.bias[cout] : 0.0f;
    const float al = a.alpha ? a.alpha[cout] : 1.0f, be = a.alpha ? a.beta[cout] : 0.0f;
    const bool bn = a.alpha != nullptr;
    const int nRp = POOL ? H / 2 : (H + 1) / 2, nX = (W + 15) / 16;
    // streaming hop: pooled rows [keep_lo, keep_hi] come from the previous hop's sequence buffer, the tile rows outside are computed
    const bool carry = POOL && a.seq_out && a.seq_prev != nullptr && a.keep_hi >= a.keep_lo;
    const int n_keep = carry ? a.keep_hi - a.keep_lo + 1 : 0;
    const int nT = (nRp - n_keep) * nX;
    auto tile_of = [&](int u) {                                 // u-th computed tile -> tile index R * nX + X
        const int Ru = u / nX, Xu = u - Ru * nX;
        return (carry && Ru >= a.keep_lo ? Ru + n_keep : Ru) * nX + Xu;
    };
    const int t_base = nT / NW, t_rem = nT - t_base * NW;
    const int t_begin = wave * t_base + min(wave, t_rem), t_end = t_begin + t_base + (wave < t_rem ? 1 : 0);
    const int dyi = (i >> 1) & 1, xi = 2 * (i >> 2) + (i & 1);
    const int lane_off = dyi * rowB + xi * PS3 + 16 * hi;      // the lane's pixel inside a tile, its 8 channels of a k-block
    const unsigned char* wlane = Wt + lane * 16;

    // Fused export-form AvgPool (full-height windows along x, _export/onnx.py:146-152).  A lane's registers 4k+q hold column 16X + 4k + 2hi + (q & 1) of rows 2R + (q >> 1),
    // so the two rows of a column are added first (row 2R, then 2R + 1) and the window test runs once per column.
    auto avg_epilogue = [&](const f32x16& acc, int R, int X, float* wsum, AvgWin aw) {
        const bool row1 = 2 * R + 1 < H;
        if (aw.along_y) {      // (wave-uniform) windows along y: the lane's eight columns of a row are added first, then the row's windows
            float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v0 = acc[4 * k + dx] + bias, v1 = acc[4 * k + 2 + dx] + bias;
                    if (bn) { v0 = v0 * al + be; v1 = v1 * al + be; }
                    v0 = trunk_act<ACT>(v0);
                    v1 = trunk_act<ACT>(v1);
                    const bool okx = 16 * X + 4 * k + 2 * hi + dx < W;
                    r0 += okx ? v0 : 0.0f;
                    r1 += okx ? v1 : 0.0f;
                }
            if (!row1) r1 = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < aw.ow) {
                    wsum[j] += (unsigned)(2 * R - j * aw.sw) < (unsigned)aw.kw ? r0 : 0.0f;
                    wsum[j] += (unsigned)(2 * R + 1 - j * aw.sw) < (unsigned)aw.kw ? r1 : 0.0f;
                }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v0 = acc[4 * k + dx] + bias, v1 = acc[4 * k + 2 + dx] + bias;
                if (bn) { v0 = v0 * al + be; v1 = v1 * al + be; }
                v0 = trunk_act<ACT>(v0);
                v1 = trunk_act<ACT>(v1);
                const int x = 16 * X + 4 * k + 2 * hi + dx;
                const float col = x < W ? v0 + (row1 ? v1 : 0.0f) : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < aw.ow) wsum[j] += (unsigned)(x - j * aw.sw) < (unsigned)aw.kw ? col : 0.0f;
            }
    };

    float pre[32];                                            // the thread's 32 channel values of the next item's pixel
    // conv for tile t (and t + 1 when TWO): 9 taps x KB sixteen-channel blocks x 6 (3) products
    auto tiles = [&](int u, auto two_c, float* outb, float* wsum, AvgWin aw, const float* accb, int y0) {
        constexpr bool TWO = decltype(two_c)::value;
        const int t = tile_of(u);
        const int R0 = t / nX, X0 = t - R0 * nX;
        const int t1 = TWO ? tile_of(u + 1) : t;
        const int R1 = t1 / nX, X1 = t1 - R1 * nX;
        const unsigned char* pa = A3 + lane_off + (2 * R0) * rowB + 16 * X0 * PS3;
        const unsigned char* pb = A3 + lane_off + (2 * R1) * rowB + 16 * X1 * PS3;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        if (accb) {                                            // (wave-uniform) k-split pass: start from the sums over the earlier input channels
            auto load_acc = [&](f32x16& acc, int R, int X) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int y = y0 + 2 * R + ((r >> 1) & 1), x = 16 * X + 4 * (r >> 2) + 2 * hi + (r & 1);
                    acc[r] = (y < H && x < W) ? accb[(size_t)i * H * W + y * W + x] : 0.0f;
                }
            };
            load_acc(acc0, R0, X0);
            if (TWO) load_acc(acc1, R1, X1);
        }
        bf16x8 na[NT], nb[NT], nw[NT];
        auto fetch = [&](int step) {                           // step = tap * KB + k-block
            const int tap = step / KB, kb = step % KB;
            const int off = (tap / 3) * rowB + (tap % 3) * PS3 + 32 * kb;
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) {
                na[tm] = *reinterpret_cast<const bf16x8*>(pa + off + TS * tm);
                if (TWO) nb[tm] = *reinterpret_cast<const bf16x8*>(pb + off + TS * tm);
                nw[tm] = *reinterpret_cast<const bf16x8*>(wlane + (step * NT + tm) * 1024);
            }
        };
        // terms 0 = hi, 1 = mid / lo, 2 = lo; smallest products first (the order of trunk_x3's tap_mfma<6>)
        auto products = [&](const bf16x8* x, const bf16x8* w, f32x16& acc) {
            if (H2) {
                const f16x8 xh = __builtin_bit_cast(f16x8, x[0]), xl = __builtin_bit_cast(f16x8, x[1]);
                const f16x8 wh = __builtin_bit_cast(f16x8, w[0]), wl2 = __builtin_bit_cast(f16x8, w[1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], w[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[NT - 1], w[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], w[NT - 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], w[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], w[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], w[0], acc, 0, 0, 0);
            }
        };
        fetch(0);
#pragma unroll
        for (int step = 0; step < 9 * KB; ++step) {
            bf16x8 ca[NT], cb[NT], cw[NT];
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) { ca[tm] = na[tm]; if (TWO) cb[tm] = nb[tm]; cw[tm] = nw[tm]; }
            if (step + 1 < 9 * KB) fetch(step + 1);
            __builtin_amdgcn_sched_barrier(0);                 // next step's LDS reads stay above this step's MFMAs
            products(ca, cw, acc0);
            if (TWO) products(cb, cw, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (H2) {                                              // back to the true scale (a power of two)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= c_out; if (TWO) acc1[r] *= c_out; }
        }
        if (AVG) {
            avg_epilogue(acc0, R0, X0, wsum, aw);
            if (TWO) avg_epilogue(acc1, R1, X1, wsum, aw);
        } else {
            conv_tile_epilogue<ACT, POOL, AVG>(acc0, R0, X0, bias, al, be, bn, outb, i, hi, Ho, Wo, y0 / 2, wsum, aw, seq_ch);
            if (TWO) conv_tile_epilogue<ACT, POOL, AVG>(acc1, R1, X1, bias, al, be, bn, outb, i, hi, Ho, Wo, y0 / 2, wsum, aw, seq_ch);
        }
    };

    // Staging: thread p < H*W owns pixel p (C1 > 32: thread p + q H*W owns the q-th 32-channel chunk of pixel p).  Its 32 channel
    // values (32 coalesced plane reads) are fetched into registers one clip AHEAD - the loads are in flight under the MFMA loop -
    // then split and written channels-last as twelve 16-byte LDS stores.  (conv3_x3_fits guarantees H*W * C1/32 <= 512.)
    // (STRIP: thread p < (SH + 3) * W owns pixel p of the strip's rows y0 - 1 .. y0 + SH + 1 - A3 row = strip row, zeros outside the plane)
    const int b_first = (int)blockIdx.x / ngroups, b_step = (int)gridDim.x / ngroups;
    const int HW = H * W;
    const bool stager = STRIP ? tid < (SH + 3) * W : tid < HW * (C1 / 32);
    const int chunk = (C1 > 32 && stager) ? tid / HW : 0, pix = tid - chunk * HW;
    const int py = stager ? pix / W : 0, px = stager ? pix - py * W : 0;
    unsigned char* my_px = A3 + (STRIP ? py : py + 1) * rowB + (px + 1) * PS3 + 64 * chunk;
    // dense planes [B][32][H][W], or (streaming hop) per-clip rings of rows written by the fused trunk: row py of the window at
    // ring row (in_row0 + py) % in_ring_rows
    const size_t in_clip = a.in_ring_rows ? a.in_clip_stride : (size_t)a.w_cin * HW;
    const size_t in_ch = a.in_ring_rows ? a.in_ch_stride : (size_t)HW;
    const int in_px = a.in_ring_rows ? ((a.in_row0 + py) % a.in_ring_rows) * W + px : pix;
    const int n_items = a.B * nS;                               // work items: clips (STRIP: clip-major (clip, strip) pairs)
    auto prefetch = [&](int item) {
        if (stager && item < n_items) {
            if (STRIP) {
                const int b = item / nS, y = (item - b * nS) * SH - 1 + py;
                const bool in_plane = (unsigned)y < (unsigned)H;
                const float* xin = a.in + (size_t)b * in_clip + (size_t)a.w_cin_off * in_ch + (in_plane ? y * W + px : 0);
#pragma unroll
                for (int c = 0; c < 32; ++c) pre[c] = in_plane ? xin[(size_t)c * in_ch] : 0.0f;
                return;
            }
            const float* xin = a.in + (size_t)item * in_clip + (size_t)(a.w_cin_off + 32 * chunk) * in_ch + in_px;
#pragma unroll
            for (int c = 0; c < 32; ++c) pre[c] = xin[(size_t)c * in_ch];
        }
    };
    // per-lane avg-pool partials live in the 16 pad bytes of pixels 0..511 (never read by the MFMAs, never written by
    // the staging) or, for small planes, in their own region behind the weights
    // thread t's partial sums: the pad of pixel t while there are pixels, behind the weights after that
    // (the right halo column is left out: with a shortened pitch its pad overlaps the next row's left halo pixel)
    const int npix = (H + 3) * (Wp - 1);
    auto part_ptr = [&](int t) {
        return reinterpret_cast<float*>(t < npix ? A3 + (size_t)(t / (Wp - 1)) * rowB + (size_t)(t % (Wp - 1)) * PS3 + TS * NT : Wt + WT_BYTES + (size_t)(t - npix) * 16);
    };
    float* my_part = part_ptr(tid);
    // the reducing lanes' 2 * NW partial addresses do not change from item to item (forming one costs an integer division: 16 of them per
    // item were 3400 of the E2E stage's 16 700 clocks, on two waves with six waiting - tools/ubench/conv3_trace)
    int red_off[AVG ? 2 * NW : 1];
    if (AVG && tid < 32 * a.avg_ow) {
        const int ci = tid / a.avg_ow, j = tid - ci * a.avg_ow;
#pragma unroll
        for (int k = 0; k < 2 * NW; ++k)
            red_off[AVG ? k : 0] = (int)(reinterpret_cast<unsigned char*>(part_ptr((k >> 1) * 64 + (k & 1) * 32 + ci) + j) - lds3);
    }
    prefetch(b_first);
    __syncthreads();
    [[maybe_unused]] int c3_item = -1;
    for (int item = b_first; item < n_items; item += b_step) {
        ++c3_item;
        C3_STAMP(0)
        const int b = STRIP ? item / nS : item;
        const int y0 = STRIP ? (item - b * nS) * SH : 0;       // first plane row of the strip
        int tb = t_begin, te = t_end;
        if (STRIP) {                                           // the last strip may be shorter: the wave's share of this strip's tiles
            const int hs = min(SH, H - y0), nTi = (POOL ? hs / 2 : (hs + 1) / 2) * nX;
            const int base = nTi / NW, rem = nTi - base * NW;
            tb = wave * base + min(wave, rem); te = tb + base + (wave < rem ? 1 : 0);
        }
        if (stager) {
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                if (H2) {
                    uint32_t th[4], tl[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) nww_split2h(pre[8 * c8 + 2 * e] * s_in, pre[8 * c8 + 2 * e + 1] * s_in, th[e], tl[e]);
                    *reinterpret_cast<uint4*>(my_px + 16 * c8) = make_uint4(th[0], th[1], th[2], th[3]);
                    *reinterpret_cast<uint4*>(my_px + TS + 16 * c8) = make_uint4(tl[0], tl[1], tl[2], tl[3]);
                } else {
                    uint32_t th[4], tm[4], tl[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t h0, m0, l0, h1, m1, l1;
                        split3c(pre[8 * c8 + 2 * e], h0, m0, l0);
                        split3c(pre[8 * c8 + 2 * e + 1], h1, m1, l1);
                        th[e] = (h0 >> 16) | (h1 & 0xffff0000u);
                        tm[e] = (m0 >> 16) | (m1 & 0xffff0000u);
                        tl[e] = (l0 >> 16) | (l1 & 0xffff0000u);
                    }
                    *reinterpret_cast<uint4*>(my_px + 16 * c8) = make_uint4(th[0], th[1], th[2], th[3]);
                    *reinterpret_cast<uint4*>(my_px + TS + 16 * c8) = make_uint4(tm[0], tm[1], tm[2], tm[3]);
                    *reinterpret_cast<uint4*>(my_px + 2 * TS + 16 * c8) = make_uint4(tl[0], tl[1], tl[2], tl[3]);
                }
            }
        }
        C3_STAMP(1)
        __syncthreads();
        C3_STAMP(2)
        prefetch(item + b_step);
        C3_STAMP(3)
        // planes [Cout][Ho][Wo], or (seq_out, pooled mode) the clip's sequence rows [Wo][Cout * Ho] at channel 32 grp
        float* outb = seq_ch ? a.out + (size_t)b * Wo * seq_ch + (size_t)32 * grp * Ho
                             : a.out + ((size_t)b * a.Cout + 32 * grp) * Ho * Wo;
        float wsum[4] = {0.f, 0.f, 0.f, 0.f};
        const AvgWin aw{a.avg_kw, a.avg_sw, a.avg_ow, a.avg_y};
        const float* accb = a.acc_in ? a.acc_in + ((size_t)b * a.Cout + 32 * grp) * HW : nullptr;
        int t = tb;
        for (; t + 1 < te; t += 2) tiles(t, std::true_type{}, outb, wsum, aw, accb, y0);
        if (t < te) tiles(t, std::false_type{}, outb, wsum, aw, accb, y0);
        if (carry) {                                           // kept rows: the previous hop's values, keep_shift rows further down
            const float* prev = a.seq_prev + (size_t)b * Wo * seq_ch + (size_t)32 * grp * Ho;
            for (int idx = tid; idx < Wo * 32 * n_keep; idx += NTHR) {
                const int x = idx / (32 * n_keep), r2 = idx - x * (32 * n_keep), c = r2 / n_keep, j = a.keep_lo + (r2 - c * n_keep);
                outb[(size_t)x * seq_ch + c * Ho + j] = prev[(size_t)x * seq_ch + c * Ho + j + a.keep_shift];
            }
        }
        if (AVG) *reinterpret_cast<float4*>(my_part) = make_float4(wsum[0], wsum[1], wsum[2], wsum[3]);
        C3_STAMP(4)
        __syncthreads();                                       // every wave is done reading A3; partials visible
        C3_STAMP(5)
        if (AVG) {
            // fixed-order reduction: one lane per (channel, window) adds its channel's 2 * NW partials in (wave, half)
            // order.  The next clip's staging may start meanwhile: it touches neither pads nor the partial region, and
            // the barrier behind it separates these reads from the next partial writes.
            const float inv = 1.0f / (float)((a.avg_y ? W : H) * a.avg_kw);
            if (tid < 32 * a.avg_ow) {                        // (32 * avg_ow <= 128 lanes)
                const int ci = tid / a.avg_ow, j = tid - ci * a.avg_ow;
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < 2 * NW; ++k) sum += *reinterpret_cast<const float*>(lds3 + red_off[AVG ? k : 0]);      // (wave, half) order
                a.out[((size_t)b * a.Cout + 32 * grp + ci) * a.avg_ow + j] = sum * inv;
            }
        }
    }
}
}  // namespace

static size_t conv3_x3_a3_bytes(int H, int W, int avg_ow, bool h2, int C1 = 32) { return ((size_t)(H + 3) * conv3_row_pitch(H, W, avg_ow, h2, C1) + 32 + 15) & ~(size_t)15; }

static size_t conv3_x3_lds(int H, int W, int avg_ow, bool h2, int C1 = 32) {
    const int npix = (H + 3) * (W + 1);                        // the avg-pool partials live in the pixels' pads; the rest behind the weights
    return conv3_x3_a3_bytes(H, W, avg_ow, h2, C1) + (size_t)9 * (C1 / 16) * (h2 ? 2 : 3) * 1024 + ((avg_ow <= 0 || npix >= 512) ? 0 : (size_t)(512 - npix) * 16);
}
size_t conv3_x3_lds_bytes(int H, int W, int avg_ow) { return conv3_x3_lds(H, W, avg_ow, false); }

// (decided on the three-term form's footprint in every arithmetic: which kernel a shape takes does not depend on the arithmetic switch)
bool conv3_x3_fits(int H, int W, int Cout, int avg_ow, int pool) {
    if (Cout % 32 != 0 || H < 2 || W < 2 || H * W > 512) return false;
    if (avg_ow > 0 && (pool || avg_ow > 4)) return false;
    return conv3_x3_lds_bytes(H, W, avg_ow) <= 160 * 1024;
}

// planes of more than 512 pixels: rows per strip - the largest even count whose (rows + 3) x W pixels have a stager thread each and whose LDS
// plane fits beside the weights (three-term footprint, like conv3_x3_fits); 0: no such count
int conv3_x3_strip_rows(int H, int W, int Cout) {
    if (Cout % 32 != 0 || H < 2 || W < 2 || W > 102) return 0;
    for (int sh = (512 / W - 3) & ~1; sh >= 2; sh -= 2)
        if (conv3_x3_lds(sh, W, 0, false) <= 160 * 1024) return sh;
    return 0;
}
size_t conv3_x3_strip_lds_bytes(int sh, int W) { return conv3_x3_lds(sh, W, 0, false); }

// deeper stages of a CRNN conv stack (crnn_cnn_channels beyond the default three: architectures.py:217-225): 64 input channels, pooled
// planes, two-term form only - the three-term weights of one 32-channel output group alone would take 110 KB of LDS (and 128 input
// channels 147 KB in the two-term form: those stay on the general kernel)
bool conv3_x3_wide_fits(int Cin, int H, int W, int Cout) {
    if (Cin != 64 || Cout % 32 != 0 || H < 2 || W < 2 || H * W * (Cin / 32) > 512) return false;
    return conv3_x3_lds(H, W, 0, true, Cin) <= 160 * 1024;
}
size_t conv3_x3_wide_lds_bytes(int Cin, int H, int W) { return conv3_x3_lds(H, W, 0, true, Cin); }

hipError_t launch_conv3_x3(const ConvMfmaArgs& a, int max_grid, hipStream_t s) {
    const bool h2 = a.h2_in > 0.0f;
    const bool wide = a.Cin != 32;
    if (a.w_cin < 32 || a.w_cin % 32 != 0 || a.w_cin_off % 32 != 0 || a.w_cin_off + a.Cin > a.w_cin) return hipErrorInvalidValue;
    if ((a.w_cin != a.Cin || a.acc_in) && (h2 || a.in_ring_rows)) return hipErrorInvalidValue;     // k-split passes: three-term form, dense planes
    const int sh = a.strip_h;
    if (sh) {
        if (wide || a.avg_ow > 0 || a.in_ring_rows || a.seq_prev || sh < 2 || (sh & 1) || (sh + 3) * a.W > 512 || a.Cout % 32 != 0 ||
            conv3_x3_lds(sh, a.W, 0, h2) > 160 * 1024) return hipErrorInvalidValue;
    } else if (wide ? !(h2 && a.pool && a.avg_ow == 0 && a.in_ring_rows == 0 && conv3_x3_wide_fits(a.Cin, a.H, a.W, a.Cout))
                    : !conv3_x3_fits(a.H, a.W, a.Cout, a.avg_ow, a.pool)) return hipErrorInvalidValue;
    const size_t lds = conv3_x3_lds(sh ? sh : a.H, a.W, a.avg_ow, h2, a.Cin);
    ConvMfmaArgs aa = a;
    aa.row_pitch = conv3_row_pitch(sh ? sh : a.H, a.W, a.avg_ow, h2, a.Cin);
    const int ngroups = a.Cout / 32;
    long want = (long)a.B * ngroups * (sh ? (a.H + sh - 1) / sh : 1);
    int grid = (int)(want < max_grid ? want : max_grid);
    grid -= grid % ngroups;
    if (grid < ngroups) grid = ngroups;
#define C3_LAUNCH1(ACTV, POOLV, AVGV, H2V)                                                                         \
    {                                                                                                              \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv3_x3_kernel<ACTV, POOLV, AVGV, H2V>), lds); \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((conv3_x3_kernel<ACTV, POOLV, AVGV, H2V>), dim3(grid), dim3(512), lds, s, aa);           \
    }
#define C3_LAUNCH(ACTV, POOLV, AVGV)                                                                               \
    if (h2) C3_LAUNCH1(ACTV, POOLV, AVGV, true) else C3_LAUNCH1(ACTV, POOLV, AVGV, false)
#define C3_WIDE(ACTV, C1V)                                                                                                   \
    {                                                                                                                        \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv3_x3_kernel<ACTV, true, false, true, C1V>), lds);      \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL((conv3_x3_kernel<ACTV, true, false, true, C1V>), dim3(grid), dim3(512), lds, s, aa);               \
    }
#define C3_STRIP(ACTV, POOLV, H2V)                                                                                           \
    {                                                                                                                        \
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(conv3_x3_kernel<ACTV, POOLV, false, H2V, 32, true>), lds); \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL((conv3_x3_kernel<ACTV, POOLV, false, H2V, 32, true>), dim3(grid), dim3(512), lds, s, aa);          \
    }
#define C3_ACT(ACTV)                                                                                               \
    if (a.Cin == 64) C3_WIDE(ACTV, 64)                                                                             \
    else if (sh) { if (!a.pool) return hipErrorInvalidValue; if (h2) C3_STRIP(ACTV, true, true) else C3_STRIP(ACTV, true, false) } \
    else if (a.avg_ow > 0) C3_LAUNCH(ACTV, false, true) else if (a.pool) C3_LAUNCH(ACTV, true, false) else C3_LAUNCH(ACTV, false, false)
    switch (a.act) {
        case ACT_RELU: C3_ACT(ACT_RELU) break;
        case ACT_GELU: C3_ACT(ACT_GELU) break;
        case ACT_SILU: C3_ACT(ACT_SILU) break;
        case ACT_NONE:                                         // k-split pass: raw sums, un-pooled planes (no bias, no BN)
            if (h2 || wide || a.pool || a.avg_ow > 0 || a.bias || a.alpha || a.seq_out) return hipErrorInvalidValue;
            if (sh) C3_STRIP(ACT_NONE, false, false) else C3_LAUNCH1(ACT_NONE, false, false, false)
            break;
        default: return hipErrorInvalidValue;
    }
#undef C3_ACT
#undef C3_WIDE
#undef C3_STRIP
#undef C3_LAUNCH
#undef C3_LAUNCH1
    return hipGetLastError();
}
