// frontend2.hip - wave-private fused framing / Hann / 400-point real FFT / power / mel / dB kernel for gfx950.
//
// The arithmetic of fe_steps.h (400 real -> 200 complex = 8 x 25) in a wave-private execution structure:
// every WAVE owns FE2_G = 8 consecutive frames of one clip and runs all stages on them by itself, so the main loop
// has no workgroup barrier at all - the 12 waves of a CU drift apart and cover each other's LDS / global latency.
//   S1  lane = (frame slot 0..1, column n2 0..24): 8 sample pairs straight from global memory (4-byte loads, L1/L2
//       hits: frames overlap 60 %), window + radix-8 + twiddle with the lane's 8 window pairs and 7 twiddles held in
//       registers for the whole launch; next item's samples are prefetched into registers during S3/S4
//   S2  lane = (frame 0..7, row k1 0..7): 25-point DFT in registers, written back in natural bin order
//   S3  lane = bin (k and 64 + k): split + |X|^2, twiddles in registers, powers written in place
//   S4  mel contraction on v_mfma_f32_16x16x4_f32 (M = frames, N = 16 filters, K = the tile's own bin range; the
//       matrix pipe is otherwise idle in this kernel and runs beside the other waves' VALU work) or, for A/B
//       measurements, the sparse VALU loop of the first kernel; 10 log10 on the accumulator layout
//   out the wave's 8 x n_mels block leaves through a small LDS stage as 16-byte coalesced stores
// LDS: 12.8 KB per wave (400 dwords per frame, reused in place by every stage) -> 12 waves per CU.
// HBM traffic per clip: 2*N bytes read + 4*n_mels*frames written; nothing else leaves the CU.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "fe_steps.h"
#include "frontend.h"
#include "layers.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Fe2MelLds {                 // sparse-mel tables of the VALU variant (packed: 3 KB)
    uint32_t desc[FE_MAX_MELS];    // lo | cnt << 8 | off << 16
    float w[FE_MAX_MELW];
};

// LDS traffic inside one wave is ordered by the hardware; these fences only stop the compiler from moving LDS
// accesses across a stage boundary (no instruction is emitted for wavefront scope).
__device__ __forceinline__ void fe2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8 (even, odd) sample pairs of column n2 of an interior frame starting at sample `base` (even): aligned 4-byte loads
__device__ __forceinline__ void fe2_load_column(const int16_t* __restrict__ x, int base, int n2, uint32_t v[8]) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(x + base) + n2;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = p[25 * n1];
}

// 10 log10 on the hardware log2 (v_log_f32, 1 ulp): |error| <= 2e-5 dB down to the -100 dB floor, against ~25 VALU
// instructions for the correctly rounded log10f - and S4 evaluates it 16 times per item on the accumulator layout.
__device__ __forceinline__ float fe2_db(float mel, float amin, float mult, float floor_db) {
    const float db = (mult * 0.30102999566398120f) * __log2f(fmaxf(mel, amin));
    return mel > amin ? db : floor_db;      // the clamp floor exactly as the reference computes it (-100 dB)
}

// MEL: how S4 contracts the powers with the filterbank
//   2  lane = filter (n_mels <= 64, every filter <= MAXT taps): the lane's weights stay in MAXT registers for the whole
//      launch, one fmaf per tap and one 16-byte LDS read per four taps (the taps start at a multiple of four bins, zero weights
//      in front), zero-padded to MAXT - same summation order as the sparse loop.
//      The default: on gfx950 the float32 MFMA runs at the VALU rate AND blocks the SIMD's VALU while it runs
//      (tools/ubench/mfma_valu_overlap.hip: a v_mfma_f32_16x16x4_f32 wave and a VALU wave on one SIMD take a + b,
//      AGPR accumulators or not), so the dense 16 x 16 x K tiles with half their rows empty cost 2048 matrix-pipe
//      clocks per item against ~900 for these 370 VALU instructions.
//   1  v_mfma_f32_16x16x4_f32 tiles (any n_mels <= 128, any filterbank)
//   0  sparse loop over LDS tables (any filterbank; A/B reference)
// FAST_OUT: frames-major log-mel only (the PCM -> logit path): branch-free S4 epilogue
#define FE2_PARAMS                                                                                                      \
    const int16_t *__restrict__ pcm, size_t row_stride, int B, int N, int T, int ngroups, int hop, int pad, int n_mels,    \
        float amin, float db_mult, float floor_db, const FeTables *__restrict__ gtb, const Fe2MelPlan *__restrict__ plan,  \
        float *__restrict__ out_db, float *__restrict__ out_mel, int frames_major, int dbg, int gsz, Fe2Sub sub
#define FE2_ARGS pcm, row_stride, B, N, T, ngroups, hop, pad, n_mels, amin, db_mult, floor_db, gtb, plan, out_db, out_mel, frames_major, dbg, gsz, sub
// RING: the streaming instances (frame subsets, ring-addressed output; Fe2Sub) - compiled apart so that the batch kernels keep their
// register budget (three more VGPRs put the 28-tap instance into scratch)
template <int MEL, int FAST_OUT, int MAXT, bool RING>
__device__ __forceinline__ void fe2_wave_body(FE2_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MFMA_MEL = MEL == 1;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwv = blockDim.x >> 6;
    const int tb_bytes = MEL == 0 ? (int)sizeof(Fe2MelLds) : 0;
    Fe2MelLds* mt = reinterpret_cast<Fe2MelLds*>(smem);
    float* slab = reinterpret_cast<float*>(smem + tb_bytes) + wv * (FE2_G * FE2_FRAME_DW);
    if (MEL == 0) {       // the only workgroup barrier of the kernel: sparse-mel tables -> LDS, once
        for (int i = threadIdx.x; i < FE_MAX_MELS; i += blockDim.x)
            mt->desc[i] = (uint32_t)gtb->mel_lo[i] | ((uint32_t)gtb->mel_cnt[i] << 8) | ((uint32_t)gtb->mel_off[i] << 16);
        for (int i = threadIdx.x; i < FE_MAX_MELW; i += blockDim.x) mt->w[i] = gtb->melw[i];
        __syncthreads();
    }
    // ---- per-lane constants, resident in registers for the whole launch
    const int n2 = lane % 25, slot = lane / 25;             // S1: lanes 50..63 idle
    nww_c32 win[8], tw[7];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) win[n1] = gtb->win2[25 * n1 + n2];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw[k1 - 1] = gtb->tw200[k1 * 25 + n2];
    const int k3a = lane, k3b = 64 + lane;                   // S3: bins k3a (all lanes) and k3b (lanes 0..36)
    const bool has_b = k3b <= 100;
    const nww_c32 twa = gtb->tw400[k3a], twb = gtb->tw400[has_b ? k3b : 100];
    const int ia2 = k3a ? FE_M - k3a : 0;                    // partner bin of k3a (Z[200] = Z[0])
    const int f2 = lane >> 3, k1_2 = lane & 7;               // S2: frame, row
    // copy-out: stage offsets of the lane's four 16-byte pieces of the (frames x n_mels) block (index division done once)
    int co_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * lane + 256 * r, f = i / n_mels, j = i - f * n_mels;
        co_off[r] = f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + j;
    }
    // S4 (MFMA): per-chunk metadata, lane c of one register = chunk c (read back with v_readlane)
    // S4 (register filters): lane = filter
    // A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), 16 lanes over the 64 banks
    // (MI355X_MICROARCH.md, LDS): lanes of a group conflict when their 16-byte pieces differ by a multiple of 256 bytes.  A group takes SIXTEEN
    // CONSECUTIVE filters (their first bins lie within ~64 bins of each other: distinct or identical pieces); with lane = filter the
    // groups mixed filters 100+ bins apart and every S4 read cost almost two LDS cycles (SQ_LDS_BANK_CONFLICT 7.3e6 of 4.2e7 per launch).
    const int mel_l5 = lane & 31;
    const int mel_j = (lane & 32) + ((mel_l5 < 4) ? mel_l5 : (mel_l5 < 12) ? 12 + mel_l5 : (mel_l5 < 16) ? mel_l5 - 8 : (mel_l5 < 20) ? 8 + mel_l5
                                     : (mel_l5 < 28) ? mel_l5 - 12 : mel_l5);
    float wreg[MAXT];
    int mel_lo_lane = 0;
    if (MEL == 2) {
        // the lane's taps start at a bin that is a MULTIPLE OF FOUR (zero weights in front of a filter that starts later:
        // fmaf(p, 0, 0) = +0, the sum is unchanged bit for bit) so that S4 reads four powers per 16-byte LDS instruction -
        // 5 instead of 20 LDS reads per frame and lane (one cycle per tap instead of two, before conflicts)
        const int j = min(mel_j, n_mels - 1);
        const int lo = gtb->mel_lo[j], sh4 = lo & 3;
        mel_lo_lane = lo - sh4;
        const int cnt = mel_j < n_mels ? gtb->mel_cnt[j] : 0, off = gtb->mel_off[j];
#pragma unroll
        for (int i = 0; i < MAXT; ++i) wreg[i] = (i >= sh4 && i - sh4 < cnt) ? gtb->melw[off + i - sh4] : 0.0f;
    }
    const int mel_nchunks = MFMA_MEL ? plan->nchunks : 0;
    const uint32_t mel_meta = MFMA_MEL ? plan->chunk_meta[lane] : 0u;
    const bool aligned = ((reinterpret_cast<uintptr_t>(pcm) | (row_stride * sizeof(int16_t))) & 3) == 0;

    const int total = B * ngroups;
    const int stride = gridDim.x * nwv;
    uint32_t cur[8];                                         // samples of the S1 iteration about to run (lane's column)
    auto geom = [&](int item, int& b, int& t0, int& nf) {
        b = item / ngroups;
        const int g = item - b * ngroups;
        if (!RING || sub.nr == 0) {
            t0 = g * gsz;                                    // gsz = frames per group: FE2_G, or fewer for a handful of clips
            nf = min(gsz, T - t0);
        } else {                                             // frame subset (streaming hop): groups of range r follow those of r - 1
            int r = 0;
            while (r + 1 < sub.nr && g >= sub.gend[r]) ++r;
            t0 = sub.t0[r] + (g - (r ? sub.gend[r - 1] : 0)) * gsz;
            nf = min(gsz, sub.t1[r] - t0);
        }
    };
    // An item is "interior" when all its frames lie inside the clip and the clip is 4-byte aligned: S1 reads its
    // samples straight from global memory, one iteration ahead (and the first iteration of the NEXT item during
    // S3/S4).  Edge items (first / last group of a centred clip; odd alignment) stage their reflect-padded span in
    // LDS first - in the regions of frames 6 and 7, which S1 overwrites last.
    auto interior = [&](int t0, int nf) {
        const int s0 = t0 * hop - pad;
        return aligned && s0 >= 0 && s0 + (nf - 1) * hop + FE_NFFT <= N;
    };
    auto prefetch_first = [&](int item) {
        int b, t0, nf;
        geom(item, b, t0, nf);
        if (interior(t0, nf) && slot < nf && slot < 2)
            fe2_load_column(pcm + (size_t)b * row_stride, (t0 + slot) * hop - pad, n2, cur);
    };
    int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * nwv + wv));
    if (item < total) prefetch_first(item);
    for (; item < total; item += stride) {
        int b, t0, nf;
        geom(item, b, t0, nf);
        const int16_t* x = pcm + (size_t)b * row_stride;
        const bool inner = interior(t0, nf);
        const uint32_t* span = reinterpret_cast<const uint32_t*>(slab + 6 * FE2_FRAME_DW);
        if (!inner) {      // edge item: reflect-padded span -> LDS
            const int s0 = t0 * hop - pad, npairs = ((nf - 1) * hop + FE_NFFT) >> 1;
            uint32_t* sp = reinterpret_cast<uint32_t*>(slab + 6 * FE2_FRAME_DW);
            for (int i = lane; i < npairs; i += 64) {
                const int q = s0 + 2 * i;
                sp[i] = (uint32_t)(uint16_t)x[fe_reflect(q, N)] | ((uint32_t)(uint16_t)x[fe_reflect(q + 1, N)] << 16);
            }
            fe2_wave_sync();
            if (slot < nf && slot < 2) {
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) cur[n1] = span[((slot * hop) >> 1) + n2 + 25 * n1];
            }
        }
        // ---- S1: window + radix-8 + twiddle -> Y[f][k1][n2]; two frames per iteration, next iteration's samples in flight
        const int nit = (nf + 1) >> 1;                          // frame pairs that exist (a short group stops early)
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            const int f = 2 * it + slot, fn = f + 2;
            uint32_t nxt[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) nxt[n1] = 0u;
            if (slot < 2 && fn < nf && it + 1 < FE2_G / 2) {
                if (inner) {
                    fe2_load_column(x, (t0 + fn) * hop - pad, n2, nxt);
                } else {
#pragma unroll
                    for (int n1 = 0; n1 < 8; ++n1) nxt[n1] = span[((fn * hop) >> 1) + n2 + 25 * n1];
                }
            }
            if (slot < 2 && f < nf && !(dbg & 1)) {
                nww_c32 z[8];
                fe2_s1(cur, win, tw, z);
                nww_c32* y = reinterpret_cast<nww_c32*>(slab + f * FE2_FRAME_DW) + n2;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) y[k1 * 25] = z[k1];
            }
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) cur[n1] = nxt[n1];
        }
        fe2_wave_sync();
        // ---- S2: 25-point DFT of row k1, in place, output in natural bin order Z[k1 + 8 k2]
        if (f2 < nf && !(dbg & 2)) {
            nww_c32* zf = reinterpret_cast<nww_c32*>(slab + f2 * FE2_FRAME_DW);
            const nww_c32* row = zf + k1_2 * 25;
            nww_c32* dst = zf + k1_2;
            dft25<true>([&](int i) { return row[i]; }, [&](int i, nww_c32 v) { dst[8 * i] = v; });
        }
        fe2_wave_sync();
        // next item's samples: in flight during S3/S4 (S2, the register-hungry stage, is behind us)
        if (item + stride < total) prefetch_first(item + stride);
        // ---- S3: split + power, in place.  All of a frame's reads precede its writes (one wave, LDS in order).
        if (!(dbg & 4)) {
            const nww_c32* z0 = reinterpret_cast<const nww_c32*>(slab);
            nww_c32 a1 = z0[k3a], b1 = z0[ia2], a2 = z0[has_b ? k3b : 0], b2 = z0[has_b ? FE_M - k3b : 0];
            // unrolled over the group's frames; the next frame's bins are read unconditionally (behind the group's last frame the region
            // holds stale data that is never used), so the four operands rotate by renaming instead of through register copies
#pragma unroll
            for (int f = 0; f < FE2_G; ++f) {
                if (f >= nf) break;                              // wave-uniform
                nww_c32 na1 = a1, nb1 = b1, na2 = a2, nb2 = b2;
                if (f + 1 < FE2_G) {
                    const nww_c32* zn = reinterpret_cast<const nww_c32*>(slab + (f + 1) * FE2_FRAME_DW);
                    na1 = zn[k3a]; nb1 = zn[ia2]; na2 = zn[has_b ? k3b : 0]; nb2 = zn[has_b ? FE_M - k3b : 0];
                }
                float pa1, pb1, pa2, pb2;
                fe_s3_core(a1, b1, twa, &pa1, &pb1);
                fe_s3_core(a2, b2, twb, &pa2, &pb2);
                float* p = slab + f * FE2_FRAME_DW + FE2_PSHIFT(f);
                p[k3a] = pa1;
                p[FE_M - k3a] = pb1;
                if (has_b) { p[k3b] = pa2; p[FE_M - k3b] = pb2; }
                a1 = na1; b1 = nb1; a2 = na2; b2 = nb2;
            }
        }
        fe2_wave_sync();
        // ---- S4: mel + dB
        if (!(dbg & 8)) {
            if (MFMA_MEL) {
                // A: lane -> (frame = lane % 16 (rows 8..15 mirror 0..7 and are discarded), k = lane / 16);
                // D: lane -> rows 4 (lane / 16) + i (valid for lanes 0..31), filter 16 t + lane % 16.
                // Chunks of FE2_CHUNK steps; the next chunk's operands (A from LDS, B from the L1-resident plan) are
                // fetched before the current chunk's MFMAs so the matrix pipe never waits on a load.
                const int fa = lane & 7;
                const float* arow = slab + fa * FE2_FRAME_DW + FE2_PSHIFT(fa) + (lane >> 4);
                const float* bp = plan->b + lane;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                auto load_chunk = [&](int c, float (&av)[FE2_CHUNK], float (&bv)[FE2_CHUNK]) {
                    const float* ap = arow + (__builtin_amdgcn_readlane(mel_meta, c) & 0xfff);
                    const float* bq = bp + (size_t)((dbg & 16) ? 0 : c) * (FE2_CHUNK * 64);
#pragma unroll
                    for (int s = 0; s < FE2_CHUNK; ++s) { av[s] = ap[4 * s]; bv[s] = bq[64 * s]; }
                };
                auto run_chunk = [&](int c, const float (&av)[FE2_CHUNK], const float (&bv)[FE2_CHUNK]) {
#pragma unroll
                    for (int s = 0; s < FE2_CHUNK; s += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s + 1], bv[s + 1], acc1, 0, 0, 0);
                    }
                    const uint32_t meta = __builtin_amdgcn_readlane(mel_meta, c);
                    if (meta & 0x10000u) {          // last chunk of its tile: dB + stage, then start the next tile
                        const int j = 16 * ((meta >> 12) & 0xf) + (lane & 15);
                        if (lane < 32 && j < n_mels) {
                            if (FAST_OUT) {
                                // rows of frames >= nf hold garbage: they are staged too and never copied out
                                float* st = slab + (4 * (lane >> 4)) * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(4 * (lane >> 4)) + j;
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    st[i * FE2_FRAME_DW + FE2_PSHIFT(i)] = fe2_db(acc0[i] + acc1[i], amin, db_mult, floor_db);
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int f = 4 * (lane >> 4) + i;
                                    const float m = acc0[i] + acc1[i];
                                    const float db = fe2_db(m, amin, db_mult, floor_db);
                                    if (f < nf) {
                                        if (frames_major) {
                                            slab[f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + j] = db;
                                            if (out_mel) out_mel[((size_t)b * T + t0 + f) * n_mels + j] = m;
                                        } else {
                                            const size_t o = ((size_t)b * n_mels + j) * T + t0 + f;
                                            if (out_db) out_db[o] = db;
                                            if (out_mel) out_mel[o] = m;
                                        }
                                    }
                                }
                            }
                        }
                        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                };
                // two operand buffers, two chunks per trip: a chunk's MFMAs run while the next chunk's loads are in flight
                float a0[FE2_CHUNK], b0[FE2_CHUNK], a1[FE2_CHUNK], b1[FE2_CHUNK];
                const int last = mel_nchunks - 1;
                load_chunk(0, a0, b0);
#pragma unroll 1
                for (int c = 0; c < mel_nchunks; c += 2) {
                    load_chunk(min(c + 1, last), a1, b1);
                    run_chunk(c, a0, b0);
                    load_chunk(min(c + 2, last), a0, b0);
                    if (c + 1 < mel_nchunks) run_chunk(c + 1, a1, b1);
                }
            } else if (MEL == 2) {
                // two frames per trip: two independent fmaf chains; rows of frames >= nf hold stale (finite) data and
                // are staged too - the copy-out takes nf rows only
                const float* p0 = slab + mel_lo_lane;
#pragma unroll 1
                for (int f = 0; f < nf; f += 2) {
                    const float* pa = p0 + f * FE2_FRAME_DW + FE2_PSHIFT(f);        // PSHIFT(f) == PSHIFT(f + 1) for even f
                    const float* pb = pa + FE2_FRAME_DW;
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
                    if (mel_j < n_mels) {
                        const float da = fe2_db(ma, amin, db_mult, floor_db), db2 = fe2_db(mb, amin, db_mult, floor_db);
                        if (FAST_OUT) {
                            float* st = slab + f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + mel_j;
                            st[0] = da;
                            st[FE2_FRAME_DW] = db2;
                        } else {
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int fq = f + q;
                                const float m = q ? mb : ma, db = q ? db2 : da;
                                if (fq < nf) {
                                    if (frames_major) {
                                        slab[fq * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(fq) + mel_j] = db;
                                        if (out_mel) out_mel[((size_t)b * T + t0 + fq) * n_mels + mel_j] = m;
                                    } else {
                                        const size_t o = ((size_t)b * n_mels + mel_j) * T + t0 + fq;
                                        if (out_db) out_db[o] = db;
                                        if (out_mel) out_mel[o] = m;
                                    }
                                }
                            }
                        }
                    }
                }
            } else {
                const int f = lane & 7;
                if (f < nf) {
                    const float* prow = slab + f * FE2_FRAME_DW + FE2_PSHIFT(f);
                    for (int j = lane >> 3; j < n_mels; j += 8) {
                        const uint32_t d = mt->desc[j];
                        const float* p = prow + (d & 0xffu);
                        const float* w = mt->w + (d >> 16);
                        const int n = (d >> 8) & 0xffu;
                        float m = 0.0f;
                        for (int i = 0; i < n; ++i) m = fmaf(p[i], w[i], m);
                        const float db = fe2_db(m, amin, db_mult, floor_db);
                        if (frames_major) {
                            slab[f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + j] = db;
                            if (out_mel) out_mel[((size_t)b * T + t0 + f) * n_mels + j] = m;
                        } else {
                            const size_t o = ((size_t)b * n_mels + j) * T + t0 + f;
                            if (out_db) out_db[o] = db;
                            if (out_mel) out_mel[o] = m;
                        }
                    }
                }
            }
        }
        // ---- out (frames-major): the wave's nf x n_mels block is contiguous in HBM
        if (frames_major && out_db) {
            fe2_wave_sync();
            const int cnt = nf * n_mels;
            if (RING && sub.ring_rows) {
                // streaming ring: frame t of the window sits at row x = row0 + t and again at x +- ring_rows, so that every
                // window [row0, row0 + T) is contiguous whatever row0 is (rows of [0, 2 ring_rows) per stream)
                float* base = out_db + (size_t)b * sub.out_clip_stride;
                for (int i = 4 * lane; i < cnt; i += 256) {                 // n_mels % 4 == 0 (checked by the launcher)
                    const int f = i / n_mels, j = i - f * n_mels, x = sub.row0 + t0 + f;
                    const float4 v = *reinterpret_cast<const float4*>(slab + f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + j);
                    *reinterpret_cast<float4*>(base + (size_t)x * n_mels + j) = v;
                    *reinterpret_cast<float4*>(base + (size_t)(x < sub.ring_rows ? x + sub.ring_rows : x - sub.ring_rows) * n_mels + j) = v;
                }
            } else {
            float* dst = out_db + ((size_t)b * T + t0) * n_mels;
            if ((n_mels & 3) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * lane + 256 * r;
                    if (i < cnt) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(slab + co_off[r]);
                }
            } else {
                for (int i = lane; i < cnt; i += 64) {
                    const int f = i / n_mels, j = i - f * n_mels;
                    dst[i] = slab[f * FE2_FRAME_DW + FE2_STAGE_OFF + FE2_PSHIFT(f) + j];
                }
            }
            }
        }
        fe2_wave_sync();      // the next item's S1 overwrites the stage
    }
}

template <int MEL, int FAST_OUT, int MAXT, bool RING = false>
__global__ void __launch_bounds__(256, 3) fe2_wave_kernel(FE2_PARAMS) { fe2_wave_body<MEL, FAST_OUT, MAXT, RING>(FE2_ARGS); }

// the streaming instances exist for the frames-major fast output with the mel stage on register filters or MFMA tiles
bool fe2_subset_supported(const FeParams& p, int mel_mode) { return mel_mode != 0 && (p.n_mels & 3) == 0; }

int fe2_lds_bytes(int waves, int mel_mode) {
    return waves * FE2_G * FE2_FRAME_DW * 4 + (mel_mode == 0 ? (int)sizeof(Fe2MelLds) : 0);
}

hipError_t fe2_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p,
                      const FeTables* d_tables, const Fe2MelPlan* d_plan, float* d_db, float* d_mel, int frames_major,
                      int mel_mode, int max_taps, int block, int max_grid, hipStream_t stream, const Fe2Sub* subset) {
    if (block < 64 || block > 1024 || (block & 63)) return hipErrorInvalidValue;
    const int nwv = block / 64;
    Fe2Sub sub = subset ? *subset : Fe2Sub{};
    if (sub.nr < 0 || sub.nr > 4) return hipErrorInvalidValue;
    if ((sub.nr || sub.ring_rows) && !(frames_major && d_db && !d_mel && (p.n_mels & 3) == 0)) return hipErrorInvalidValue;
    if (sub.ring_rows && (sub.ring_rows < T || sub.row0 < 0 || sub.row0 >= sub.ring_rows)) return hipErrorInvalidValue;
    auto count_groups = [&](int g) {
        if (sub.nr == 0) return (T + g - 1) / g;
        int n = 0;
        for (int r = 0; r < sub.nr; ++r) { n += (sub.t1[r] - sub.t0[r] + g - 1) / g; sub.gend[r] = n; }
        return n;
    };
    for (int r = 0; r < sub.nr; ++r)
        if (sub.t0[r] < 0 || sub.t1[r] <= sub.t0[r] || sub.t1[r] > T) return hipErrorInvalidValue;
    int ngroups = count_groups(FE2_G);
    // mel_mode: 2 = register-resident filters (falls back to the MFMA tiles when the filterbank does not fit), 1, 0
    int mode = mel_mode;
    if (mode == 2 && (p.n_mels > 64 || max_taps > 25)) mode = 1;      // (up to three more taps than the longest filter: the 16-byte alignment)
    const int lds = fe2_lds_bytes(nwv, mode);
    const int fast = (frames_major && d_db && !d_mel) ? 1 : 0;
    const bool ring = sub.nr > 0 || sub.ring_rows > 0;
    if (ring && (mode == 0 || !fast)) return hipErrorInvalidValue;
    auto kern = mode == 0 ? fe2_wave_kernel<0, 0, 1>
              : mode == 1 ? (ring ? fe2_wave_kernel<1, 1, 1, true> : fast ? fe2_wave_kernel<1, 1, 1> : fe2_wave_kernel<1, 0, 1>)
              : max_taps <= 17 ? (ring ? fe2_wave_kernel<2, 1, 20, true> : fast ? fe2_wave_kernel<2, 1, 20> : fe2_wave_kernel<2, 0, 20>)
                               : (ring ? fe2_wave_kernel<2, 1, 28, true> : fast ? fe2_wave_kernel<2, 1, 28> : fe2_wave_kernel<2, 0, 28>);
    const void* fn = reinterpret_cast<const void*>(kern);
    {
        hipError_t e = nww_allow_lds(fn, (size_t)lds);
        if (e != hipSuccess) return e;
    }
#ifdef NWW_ABLATION      // stage-skipping builds for phase timing (results are garbage): never in the shipped library
    static const int dbg = [] { const char* e = getenv("NWW_FE_DBG"); return e ? atoi(e) : 0; }();
#else
    const int dbg = 0;
#endif
    // A handful of clips (the interpreter's calls) would put 8 frames on each of a few waves and leave the GPU empty: two
    // frames per wave instead (S1 / S3 / S4 scale with the frames, S2 does not) - B = 1: 13 -> 7 us.  Register-filter mel only.
    const int small_g = 2;
    int gsz = FE2_G;
    if (mode == 2 && small_g >= 2 && small_g < FE2_G && (small_g % 2) == 0 && (long long)B * ngroups * 4 <= (long long)max_grid * nwv) {
        gsz = small_g;
        ngroups = count_groups(gsz);
    }
    const long long total = (long long)B * ngroups;
    long long need = (total + nwv - 1) / nwv;
    int grid = (int)(need < max_grid ? need : max_grid);
    if (grid < 1) grid = 1;
    const int pad = p.center ? FE_NFFT / 2 : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, d_pcm, row_stride, B, N, T, ngroups, p.hop, pad, p.n_mels,
                       p.amin, p.db_mult, p.db_mult * log10f(p.amin), d_tables, d_plan, d_db, d_mel, frames_major, dbg, gsz, sub);
    return hipGetLastError();
}
