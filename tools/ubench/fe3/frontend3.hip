// frontend3.hip - the frontend on the matrix pipe (gfx950): framing / window / 400-point real DFT as a PRIME-FACTOR 25 x 16 pair of
// small dense products on v_mfma_f32_16x16x32_f16 in the two-term binary16 arithmetic, then |X|^2, sparse mel, dB.  fe3.h has the
// algebra and the reference lines it replaces (nanowakeword/_export/onnx.py:42-83; AmplitudeToDB architectures.py:837,875).
//
// Work item = 16 consecutive frames of one clip (the N of the matrix instruction); a workgroup of FE3_NW = 4 waves walks items:
//   P0  staging   int16 PCM -> two binary16 planes per sample class (hi = RN16(x), lo = x - hi: exact), 16 classes x 184 entries;
//                 a thread reads four 4-byte pairs 16 samples apart (two classes x four consecutive plane entries), ONE ITEM AHEAD,
//                 and stores 8 bytes per class and term; the reflect-padded edges of a centred clip take single 2-byte loads
//   P1  stage 1   wave w owns classes 4 w .. 4 w + 3 (their window-folded 25-point matrices stay in 64 registers for the whole
//                 launch): B fragment = 8 consecutive plane entries of the lane's frame (4-byte aligned, conflict-free), 6 MFMAs
//                 per class; the float32 accumulators leave as hi / lo' binary16 (5 VALU per pair) in 16-byte stores: the lane
//                 holds re / im of one k2 for its four classes = one chunk of a stage-2 B fragment
//   P2  stage 2   13 tiles (one per k2 = 0..12) over the waves: two 16-byte LDS reads, 6 MFMAs against the ONE 16-point matrix
//                 (16 registers), re^2 + im^2 of the lane's four bins -> power rows (bin maps in registers)
//   P3  mel + dB  wave w owns frames 4 w .. 4 w + 3: lane = filter, taps in registers (frontend2's S4), v_log_f32, 16-byte copy-out
// Three workgroup barriers per item, 52 KB of LDS per workgroup.  HBM traffic per clip: 2 N bytes read + 4 n_mels T written.
// MEASURED: 0.21 ms per 4096 clips against frontend2's 0.155 - the kernel is OPT-IN (NWW_FE3 = 1).  Its 37 M VALU
// instructions per launch are half of frontend2's 78 M, but the dataflow moves ~200 KB per 16-frame item through LDS (planes 12 + 32,
// Z 27 + 27, powers 13 + 82 re-read lane = filter, dB stage 8), and at the 64 (stores) to 128 (reads) bytes per clock the LDS pipe
// sustains that is 2 500-3 500 clocks per item whatever the occupancy: 2 waves per SIMD (this default) and 4 (-DFE3_NW=16) run the same.
// Results do not depend on how frames are grouped into items (a matrix column never sees its neighbours): any batch size, frame
// subset (streaming hop) or ring placement gives the same bits per frame.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "fe3.h"
#include "fe_steps.h"
#include "frontend.h"
#include "fe3_tables.h"
#include "layers.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float fe3_db(float mel, float amin, float mult, float floor_db) {
    const float db = (mult * 0.30102999566398120f) * __log2f(fmaxf(mel, amin));
    return mel > amin ? db : floor_db;      // the clamp floor exactly as the reference computes it (-100 dB)
}

// two int16 samples (as float32: exact) -> hi = (RN16(a), RN16(b)), lo = (a - hi.a, b - hi.b) (exact: |lo| <= 8), packed binary16
__device__ __forceinline__ void fe3_split_pcm(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2v v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
// two stage-1 accumulators -> hi = RN16(acc 2^-12), lo' = RN16(acc - hi 2^12), packed (a low).  hi goes through the compiler
// (two multiplies and v_cvt_pk_f16_f32): an inline-asm statement is opaque to hipcc's hazard recognizer, and a VALU instruction
// that reads an MFMA result needs the wait states only the compiler inserts (read straight from asm the accumulators were stale:
// NaN everywhere).  The lo' instructions consume hi, so they issue behind those waits.
__device__ __forceinline__ void fe3_split_acc(float a, float b, float dn, float up, uint32_t& hi, uint32_t& lo) {
    const f32x2v v = {a * dn, b * dn};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(up), "v"(a));
    asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(up), "v"(b));
}

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

#ifdef NWW_TRACE      // tools/ubench/fe3_trace.hip: s_memtime per wave at the phase boundaries of the first items of the first workgroups
#define FE3_TRACE_PARAM , unsigned long long* __restrict__ trace
#define FE3_TRACE_ARG , g_fe3_trace
#define FE3_STAMP(k)                                                                                                   \
    if (trace && blockIdx.x < 8 && it_no < 6 && lane == 0) trace[((blockIdx.x * 6 + it_no) * 16 + wv) * 8 + (k)] = __builtin_amdgcn_s_memtime();
unsigned long long* g_fe3_trace = nullptr;
#else
#define FE3_TRACE_PARAM
#define FE3_TRACE_ARG
#define FE3_STAMP(k)
#endif
#define FE3_PARAMS                                                                                                        \
    const int16_t *__restrict__ pcm, size_t row_stride, int B, int N, int T, int ngroups, int pad, int n_mels, float amin,  \
        float db_mult, float floor_db, const FeTables *__restrict__ gtb, const Fe3Plan *__restrict__ plan,                  \
        float *__restrict__ out_db, float *__restrict__ out_mel, int frames_major, Fe2Sub sub FE3_TRACE_PARAM

// FAST_OUT: frames-major log-mel only (the PCM -> logit path and the streaming rings); MAXT: register taps of the mel stage.
// FE3_NW waves per workgroup work on FE3_IPW items (of 16 frames) per pass:
//   4 waves x 1 item (default): four classes per wave, 224 registers, two workgroups per CU - 0.213-0.222 ms per 4096 clips
//   16 waves x 2 items (-DFE3_NW=16): ONE class per wave - its stage-1 matrices are 16 registers instead of 64, the resident set ~60 -
//      so a wave fits 128 registers and the CU holds 16 waves = FOUR per SIMD (one 1024-thread workgroup, 104 KB of LDS): 0.224 ms,
//      THE SAME: occupancy is not what bounds this kernel, the LDS pipe is (~200 KB per item through 64-128 B per clock: DESIGN 4.1)
//   8 x 1: spills 27-83 registers at 128, 0.25 ms
#ifndef FE3_NW
#define FE3_NW 4
#endif
#define FE3_IPW (FE3_NW == 16 ? 2 : 1)               // items per workgroup pass
#define FE3_CPW (16 / FE3_NW)                        // sample classes per wave (stage 1)
#define FE3_FPW (FE3_IPW * FE3_F / FE3_NW)           // frames per wave (mel stage): 2 or 4, all in one item
#define FE3_TPW ((FE3_IPW * FE3_NK2 + FE3_NW - 1) / FE3_NW)    // stage-2 tiles per wave: tile q = wv + FE3_NW i -> (item q / 13, k2 = q % 13)
#define FE3_SLOT_BYTES (FE3_XP_BYTES + 2 * FE3_ZT_BYTES + FE3_P_BYTES)
#define FE3_NTASK (FE3_PL / 4 * 8)                   // staging tasks per item (368)
template <int FAST_OUT, int MAXT, bool RING>
__global__ void __launch_bounds__(64 * FE3_NW, FE3_NW == 16 ? 4 : FE3_NW == 8 ? 4 : 2) fe3_kernel(FE3_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per item slot: [planes: term, class, FE3_PL binary16][Z: term, k2, frame rows of 64 bytes][power rows + zero tail]; the dB stage
    // of a slot aliases its Z (free after stage 2)
    auto xp_of = [&](int sl) { return reinterpret_cast<unsigned char*>(smem) + sl * FE3_SLOT_BYTES; };
    auto zb_of = [&](int sl) { return reinterpret_cast<unsigned char*>(smem) + sl * FE3_SLOT_BYTES + FE3_XP_BYTES; };
    auto pw_of = [&](int sl) { return reinterpret_cast<float*>(smem + sl * FE3_SLOT_BYTES + FE3_XP_BYTES + 2 * FE3_ZT_BYTES); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, g = lane >> 4;

    // ---- launch-resident registers: the wave's stage-1 matrices (its classes x two row tiles x hi / lo), the 16-point matrix, the
    // bin maps of its stage-2 tiles, the lane's mel filter
    f16x8 a1[FE3_CPW][2][2];
#pragma unroll
    for (int ci = 0; ci < FE3_CPW; ++ci)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                a1[ci][mt][tm] = *reinterpret_cast<const f16x8*>(plan->a1[FE3_CPW * wv + ci][mt][tm][lane]);
    f16x8 a2[2][2];                      // hi, lo; the hi 2^-12 variant the lo' product needs is four v_pk_mul_f16 away (exact: powers of two)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int v = 0; v < 2; ++v) a2[mt][v] = *reinterpret_cast<const f16x8*>(plan->a2[mt][v][lane]);
    u32x2 bmr[FE3_TPW];
#pragma unroll
    for (int i = 0; i < FE3_TPW; ++i) bmr[i] = *reinterpret_cast<const u32x2*>(plan->bin[min(wv + FE3_NW * i, FE3_IPW * FE3_NK2 - 1) % FE3_NK2][lane]);
    const float p_scale = plan->p_scale;
    float wreg[MAXT];
    int mel_lo_lane;
    {
        // the lane's taps start at a bin that is a multiple of four (zero weights in front: fmaf(p, 0, 0) = +0, the sum is unchanged
        // bit for bit) so that four powers arrive per 16-byte LDS read
        const int j = min(lane, n_mels - 1);
        const int lo = gtb->mel_lo[j], sh4 = lo & 3;
        mel_lo_lane = lo - sh4;
        const int cnt = lane < n_mels ? gtb->mel_cnt[j] : 0, off = gtb->mel_off[j];
#pragma unroll
        for (int i = 0; i < MAXT; ++i) wreg[i] = (i >= sh4 && i - sh4 < cnt) ? gtb->melw[off + i - sh4] : 0.0f;
    }
    // LDS: zero once - the power rows' bins 201.. and tails are read by zero-weight taps and must be finite; the planes of a slot
    // that never gets an item (odd item count) feed MFMAs whose results are never stored but must not trap on garbage
    for (int i = tid; i < FE3_IPW * FE3_SLOT_BYTES / 4; i += 64 * FE3_NW) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    __syncthreads();                                         // (the first pass's staging stores must not race the zeroing)
    const bool aligned = ((reinterpret_cast<uintptr_t>(pcm) | (row_stride * sizeof(int16_t))) & 3) == 0;
    const float z_dn = FE3_Z_DOWN, z_up = FE3_Z_UP;

    const int total = B * ngroups;
    auto geom = [&](int item, int& b, int& t0, int& nf) {
        b = item / ngroups;
        const int gi = item - b * ngroups;
        if (!RING || sub.nr == 0) {
            t0 = gi * FE3_F;
            nf = min(FE3_F, T - t0);
        } else {                                             // frame subset (streaming hop): groups of range r follow those of r - 1
            int r = 0;
            while (r + 1 < sub.nr && gi >= sub.gend[r]) ++r;
            t0 = sub.t0[r] + (gi - (r ? sub.gend[r - 1] : 0)) * FE3_F;
            nf = min(FE3_F, sub.t1[r] - t0);
        }
    };
    // Staging task (slot, u, v): classes 2 u, 2 u + 1, plane entries 4 v .. 4 v + 3 of the slot's item (368 tasks per item, 46 blocks of
    // 8); the source sample of (entry li, class c) is t0 hop + 16 li + c - pad.  Interior tasks of an aligned clip are four 4-byte loads
    // 16 samples apart, issued ONE PASS AHEAD (pf: in flight during the previous pass's stages); tasks that touch the reflect padding or
    // the clip's end (and odd-aligned clips) take 2-byte loads at staging time.
    constexpr int TPI = FE3_IPW == 2 ? 512 : 64 * FE3_NW;    // threads that stage one item
    constexpr int NRD = FE3_IPW == 2 ? 1 : (FE3_NTASK + TPI - 1) / TPI;
    const int st_slot = FE3_IPW == 2 ? tid >> 9 : 0, st_t = FE3_IPW == 2 ? (tid & 511) : tid;
    const int su = st_t & 7, sv = st_t >> 3;
    uint32_t pf[NRD][4];
    auto task_s0 = [&](int t0, int rd) { return t0 * FE3_HOP - pad + 2 * su + 64 * (sv + (TPI / 8) * rd); };
    auto task_fast = [&](int s0) { return aligned && s0 >= 0 && s0 + 50 <= N; };
    auto prefetch = [&](int pass) {
        const int item = pass * FE3_IPW + st_slot;
        if (item >= total) return;
        int b, t0, nf;
        geom(item, b, t0, nf);
        const int16_t* x = pcm + (size_t)b * row_stride;
#pragma unroll
        for (int rd = 0; rd < NRD; ++rd) {
            const int s0 = task_s0(t0, rd);
            if (sv + (TPI / 8) * rd < FE3_PL / 4 && task_fast(s0)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pf[rd][r] = *reinterpret_cast<const uint32_t*>(x + s0 + 16 * r);
            }
        }
    };
    const int npass = (total + FE3_IPW - 1) / FE3_IPW;
    if ((int)blockIdx.x < npass) prefetch(blockIdx.x);
    int it_no = -1;
    for (int pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        ++it_no;
        FE3_STAMP(0)
        // ---- geometry of the pass's items (wave-uniform); a missing second item keeps nf = 0: computed on stale planes, never stored
        int gb[FE3_IPW], gt0[FE3_IPW], gnf[FE3_IPW];
#pragma unroll
        for (int sl = 0; sl < FE3_IPW; ++sl) {
            gb[sl] = 0; gt0[sl] = 0; gnf[sl] = 0;
            if (pass * FE3_IPW + sl < total) geom(pass * FE3_IPW + sl, gb[sl], gt0[sl], gnf[sl]);
        }
        // ---- P0: staging.  Frames past the clip's last (nf < 16) and the K padding read clamped positions: finite, never used.
        if (pass * FE3_IPW + st_slot < total) {
            const int t0 = FE3_IPW == 2 ? (st_slot ? gt0[FE3_IPW - 1] : gt0[0]) : gt0[0];
            const int16_t* x = pcm + (size_t)(FE3_IPW == 2 ? (st_slot ? gb[FE3_IPW - 1] : gb[0]) : gb[0]) * row_stride;
            unsigned char* xp = xp_of(st_slot);
#pragma unroll
            for (int rd = 0; rd < NRD; ++rd) {
                const int v = sv + (TPI / 8) * rd;
                if (v < FE3_PL / 4) {
                    const int s0 = task_s0(t0, rd);
                    uint32_t d[4];
                    if (task_fast(s0)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[r] = pf[rd][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int sa = fe_reflect(s0 + 16 * r, N), sb = fe_reflect(s0 + 16 * r + 1, N);
                            sa = min(max(sa, 0), N - 1);
                            sb = min(max(sb, 0), N - 1);
                            d[r] = (uint32_t)(uint16_t)x[sa] | ((uint32_t)(uint16_t)x[sb] << 16);
                        }
                    }
                    float fa[4], fb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { fa[r] = (float)(int16_t)(d[r] & 0xffffu); fb[r] = (float)(int16_t)(d[r] >> 16); }
                    u32x2 ha, la, hb, lb;
                    uint32_t h, l;
                    fe3_split_pcm(fa[0], fa[1], h, l); ha[0] = h; la[0] = l;
                    fe3_split_pcm(fa[2], fa[3], h, l); ha[1] = h; la[1] = l;
                    fe3_split_pcm(fb[0], fb[1], h, l); hb[0] = h; lb[0] = l;
                    fe3_split_pcm(fb[2], fb[3], h, l); hb[1] = h; lb[1] = l;
                    unsigned char* q = xp + ((2 * su) * FE3_PL + 4 * v) * 2;
                    *reinterpret_cast<u32x2*>(q) = ha;
                    *reinterpret_cast<u32x2*>(q + FE3_PL * 2) = hb;
                    *reinterpret_cast<u32x2*>(q + 16 * FE3_PL * 2) = la;
                    *reinterpret_cast<u32x2*>(q + 17 * FE3_PL * 2) = lb;
                }
            }
        }
        FE3_STAMP(1)
        if (pass + (int)gridDim.x < npass) prefetch(pass + gridDim.x);
        __syncthreads();
        FE3_STAMP(2)
        // ---- P1: stage 1, the wave's classes for every item of the pass -> their part of Z chunk (class / 4) of every (k2, frame) row.
        // All fragments first, then the MFMAs product by product: independent accumulator chains keep the matrix pipe issuing.
        {
            f16x8 xh[FE3_IPW][FE3_CPW], xl[FE3_IPW][FE3_CPW];
#pragma unroll
            for (int sl = 0; sl < FE3_IPW; ++sl)
#pragma unroll
                for (int ci = 0; ci < FE3_CPW; ++ci) {
                    const unsigned char* q = xp_of(sl) + ((FE3_CPW * wv + ci) * FE3_PL + FE3_LPF * fl + 8 * g) * 2;
                    const uint32_t* qh = reinterpret_cast<const uint32_t*>(q);
                    const uint32_t* ql = reinterpret_cast<const uint32_t*>(q + 16 * FE3_PL * 2);
                    u32x4 bh, bl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bh[e] = qh[e]; bl[e] = ql[e]; }
                    xh[sl][ci] = __builtin_bit_cast(f16x8, bh);
                    xl[sl][ci] = __builtin_bit_cast(f16x8, bl);
                }
            f32x4 acc[FE3_IPW][FE3_CPW][2];
#pragma unroll
            for (int sl = 0; sl < FE3_IPW; ++sl)
#pragma unroll
                for (int ci = 0; ci < FE3_CPW; ++ci)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[sl][ci][mt] = mfma16(a1[ci][mt][0], xh[sl][ci], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int sl = 0; sl < FE3_IPW; ++sl)
#pragma unroll
                for (int ci = 0; ci < FE3_CPW; ++ci)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[sl][ci][mt] = mfma16(a1[ci][mt][0], xl[sl][ci], acc[sl][ci][mt]);
#pragma unroll
            for (int sl = 0; sl < FE3_IPW; ++sl)
#pragma unroll
                for (int ci = 0; ci < FE3_CPW; ++ci)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[sl][ci][mt] = mfma16(a1[ci][mt][1], xh[sl][ci], acc[sl][ci][mt]);
            const int zsub = (FE3_CPW * wv & 3) * 4;            // byte offset of the wave's classes inside their 16-byte chunk
#pragma unroll
            for (int sl = 0; sl < FE3_IPW; ++sl)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint32_t vh[FE3_CPW], vl[FE3_CPW];
#pragma unroll
                        for (int ci = 0; ci < FE3_CPW; ++ci) fe3_split_acc(acc[sl][ci][mt][2 * hf], acc[sl][ci][mt][2 * hf + 1], z_dn, z_up, vh[ci], vl[ci]);
                        const int k2 = 8 * mt + 2 * g + hf;
                        if (k2 < FE3_NK2) {
                            unsigned char* q = zb_of(sl) + fe3_z_off(k2, fl, (FE3_CPW * wv) >> 2) + zsub;
                            if constexpr (FE3_CPW == 4) {
                                *reinterpret_cast<u32x4*>(q) = u32x4{vh[0], vh[1], vh[2], vh[3]};
                                *reinterpret_cast<u32x4*>(q + FE3_ZT_BYTES) = u32x4{vl[0], vl[1], vl[2], vl[3]};
                            } else if constexpr (FE3_CPW == 2) {
                                *reinterpret_cast<u32x2*>(q) = u32x2{vh[0], vh[1]};
                                *reinterpret_cast<u32x2*>(q + FE3_ZT_BYTES) = u32x2{vl[0], vl[1]};
                            } else {
                                *reinterpret_cast<uint32_t*>(q) = vh[0];
                                *reinterpret_cast<uint32_t*>(q + FE3_ZT_BYTES) = vl[0];
                            }
                        }
                    }
        }
        FE3_STAMP(3)
        __syncthreads();
        FE3_STAMP(4)
        // ---- P2: stage 2, tiles q = wv, wv + FE3_NW, ... (item q / 13, k2 = q % 13) -> power rows
        {
            f16x8 zh[FE3_TPW], zl[FE3_TPW];
#pragma unroll
            for (int ti = 0; ti < FE3_TPW; ++ti) {
                const int q = min(wv + FE3_NW * ti, FE3_IPW * FE3_NK2 - 1);        // waves without a last tile recompute another one (never stored)
                const unsigned char* zq = zb_of(q / FE3_NK2) + fe3_z_off(q % FE3_NK2, fl, g);
                zh[ti] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zq));
                zl[ti] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zq + FE3_ZT_BYTES));
            }
            f32x4 acc[FE3_TPW][2];
#pragma unroll
            for (int ti = 0; ti < FE3_TPW; ++ti)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[ti][mt] = mfma16(a2[mt][0], zh[ti], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int ti = 0; ti < FE3_TPW; ++ti)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[ti][mt] = mfma16(a2[mt][1], zh[ti], acc[ti][mt]);
#pragma unroll
            for (int ti = 0; ti < FE3_TPW; ++ti)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[ti][mt] = mfma16(a2[mt][0] * (_Float16)FE3_Z_DOWN, zl[ti], acc[ti][mt]);
#pragma unroll
            for (int ti = 0; ti < FE3_TPW; ++ti) {
                const int q = wv + FE3_NW * ti;
                if (q < FE3_IPW * FE3_NK2) {                     // wave-uniform
                    float* prow = pw_of(q / FE3_NK2) + fl * FE3_PP;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const uint32_t bw = bmr[ti][mt];
                        const int ia = (int)(int16_t)(bw & 0xffffu), ib = (int)(int16_t)(bw >> 16);
                        const f32x4 c = acc[ti][mt];
                        const float pa = fmaf(c[1], c[1], c[0] * c[0]) * p_scale;
                        const float pb = fmaf(c[3], c[3], c[2] * c[2]) * p_scale;
                        if (ia >= 0) prow[ia] = pa;
                        if (ib >= 0) prow[ib] = pb;
                    }
                }
            }
        }
        FE3_STAMP(5)
        __syncthreads();
        FE3_STAMP(6)
        // ---- P3: mel + dB of the wave's FE3_FPW frames (all in one item; lane = filter), then its rows leave as 16-byte pieces
        const int msl = (FE3_FPW * wv) >> 4, f0 = (FE3_FPW * wv) & 15;
        const int m_nf = FE3_IPW == 2 ? (msl ? gnf[FE3_IPW - 1] : gnf[0]) : gnf[0];
        const int m_b = FE3_IPW == 2 ? (msl ? gb[FE3_IPW - 1] : gb[0]) : gb[0], m_t0 = FE3_IPW == 2 ? (msl ? gt0[FE3_IPW - 1] : gt0[0]) : gt0[0];
        const int nfw = min(max(m_nf - f0, 0), FE3_FPW);
        if (nfw > 0) {
            const float* pws = pw_of(msl);
            float* stage = reinterpret_cast<float*>(zb_of(msl));
#pragma unroll 1
            for (int f = 0; f < nfw; f += 2) {
                const float* pa = pws + (f0 + f) * FE3_PP + mel_lo_lane;
                const float* pb = pa + FE3_PP;                  // f0 + f + 1 <= 15: the row exists
                float ma = 0.0f, mb = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXT; i += 4) {
                    const float4 a4 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(pa + i, 16));
                    const float4 b4 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(pb + i, 16));
                    ma = fmaf(a4.x, wreg[i], ma); mb = fmaf(b4.x, wreg[i], mb);
                    ma = fmaf(a4.y, wreg[i + 1], ma); mb = fmaf(b4.y, wreg[i + 1], mb);
                    ma = fmaf(a4.z, wreg[i + 2], ma); mb = fmaf(b4.z, wreg[i + 2], mb);
                    ma = fmaf(a4.w, wreg[i + 3], ma); mb = fmaf(b4.w, wreg[i + 3], mb);
                }
                if (lane < n_mels) {
                    const float da = fe3_db(ma, amin, db_mult, floor_db), db2 = fe3_db(mb, amin, db_mult, floor_db);
                    if (FAST_OUT) {
                        float* st = stage + (f0 + f) * n_mels + lane;
                        st[0] = da;
                        st[n_mels] = db2;                        // a row past nfw is staged too and never copied out
                    } else {
#pragma unroll
                        for (int qd = 0; qd < 2; ++qd) {
                            const float m = qd ? mb : ma, db = qd ? db2 : da;
                            if (f + qd < nfw) {
                                const int t = m_t0 + f0 + f + qd;
                                if (frames_major) {
                                    if (out_db) out_db[((size_t)m_b * T + t) * n_mels + lane] = db;
                                    if (out_mel) out_mel[((size_t)m_b * T + t) * n_mels + lane] = m;
                                } else {
                                    const size_t o = ((size_t)m_b * n_mels + lane) * T + t;
                                    if (out_db) out_db[o] = db;
                                    if (out_mel) out_mel[o] = m;
                                }
                            }
                        }
                    }
                }
            }
            if (FAST_OUT) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int cnt = nfw * n_mels;                   // n_mels % 4 == 0 (launcher)
                const float* src = stage + f0 * n_mels;
                for (int i = 4 * lane; i < cnt; i += 256) {
                    const float4 v = *reinterpret_cast<const float4*>(src + i);
                    if (RING && sub.ring_rows) {
                        float* base = out_db + (size_t)m_b * sub.out_clip_stride;
                        const int f = i / n_mels, j = i - f * n_mels, xr = sub.row0 + m_t0 + f0 + f;
                        *reinterpret_cast<float4*>(base + (size_t)xr * n_mels + j) = v;
                        *reinterpret_cast<float4*>(base + (size_t)(xr < sub.ring_rows ? xr + sub.ring_rows : xr - sub.ring_rows) * n_mels + j) = v;
                    } else {
                        *reinterpret_cast<float4*>(out_db + ((size_t)m_b * T + m_t0 + f0) * n_mels + i) = v;
                    }
                }
            }
        }
        FE3_STAMP(7)
        // the next pass's stage-1 stores into Z (= this pass's dB stage) come after its staging barrier, which every wave reaches
        // only when its copy-out above is done; its power rows are written after two more barriers
    }
}

}  // namespace

bool fe3_supported(const FeParams& p, int max_taps) {
    return p.n_fft == FE_NFFT && p.hop == FE3_HOP && p.n_mels >= 1 && p.n_mels <= 64 && max_taps <= 25;
}

hipError_t fe3_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p, const FeTables* d_tables,
                      const Fe3Plan* d_plan, float* d_db, float* d_mel, int frames_major, int max_taps, int max_grid,
                      hipStream_t stream, const Fe2Sub* subset) {
    if (!fe3_supported(p, max_taps) || !d_plan) return hipErrorInvalidValue;
    Fe2Sub sub = subset ? *subset : Fe2Sub{};
    if (sub.nr < 0 || sub.nr > 4) return hipErrorInvalidValue;
    if ((sub.nr || sub.ring_rows) && !(frames_major && d_db && !d_mel && (p.n_mels & 3) == 0)) return hipErrorInvalidValue;
    if (sub.ring_rows && (sub.ring_rows < T || sub.row0 < 0 || sub.row0 >= sub.ring_rows)) return hipErrorInvalidValue;
    for (int r = 0; r < sub.nr; ++r)
        if (sub.t0[r] < 0 || sub.t1[r] <= sub.t0[r] || sub.t1[r] > T) return hipErrorInvalidValue;
    int ngroups = 0;
    if (sub.nr == 0) {
        ngroups = (T + FE3_F - 1) / FE3_F;
    } else {
        for (int r = 0; r < sub.nr; ++r) { ngroups += (sub.t1[r] - sub.t0[r] + FE3_F - 1) / FE3_F; sub.gend[r] = ngroups; }
    }
    const bool fast = frames_major && d_db && !d_mel && (p.n_mels & 3) == 0;
    const bool ring = sub.nr > 0 || sub.ring_rows > 0;
    if (ring && !fast) return hipErrorInvalidValue;
    auto kern = max_taps <= 17 ? (ring ? fe3_kernel<1, 20, true> : fast ? fe3_kernel<1, 20, false> : fe3_kernel<0, 20, false>)
                               : (ring ? fe3_kernel<1, 28, true> : fast ? fe3_kernel<1, 28, false> : fe3_kernel<0, 28, false>);
    const int lds = FE3_IPW * FE3_SLOT_BYTES;
    {
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(kern), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    const long long total = ((long long)B * ngroups + FE3_IPW - 1) / FE3_IPW;      // passes
    if (FE3_NW == 16) max_grid = max_grid / 3 > 0 ? max_grid / 3 : 1;             // one 16-wave workgroup per CU (the caller passes 3 x CUs)
    int grid = (int)(total < max_grid ? total : max_grid);
    if (grid < 1) grid = 1;
    const int pad = p.center ? FE_NFFT / 2 : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FE3_NW), lds, stream, d_pcm, row_stride, B, N, T, ngroups, pad, p.n_mels, p.amin, p.db_mult,
                       p.db_mult * log10f(p.amin), d_tables, d_plan, d_db, d_mel, frames_major, sub FE3_TRACE_ARG);
    return hipGetLastError();
}
