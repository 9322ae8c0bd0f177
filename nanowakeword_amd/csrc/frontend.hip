// frontend.hip - fused framing / Hann / 400-point real FFT / power / sparse mel / dB kernel for gfx950.
//
// One workgroup processes chunks of FC consecutive frames of one clip:
//   S0  stage the chunk's int16 span (hop*(nf-1)+400 samples, reflect-padded at clip edges) into LDS
//       with 16-byte coalesced global loads (2-byte loads only at reflected edges / misaligned clips)
//   S1  25 radix-8 tasks / frame   (window multiply fused into the load)       -> Y  in LDS
//   S2   8 25-point DFT tasks / frame (5x5 in registers)                       -> Z  in LDS (in place)
//   S3 101 split+power tasks / frame                                           -> P  in LDS
//   S4 n_mels sparse mel + log10 tasks / frame                                 -> HBM (coalesced)
// Task bodies live in fe_steps.h (shared with the CPU emulator used by the non-GPU tests).
//
// HBM traffic per clip: 2*N bytes read (int16 PCM; chunk overlaps of 240 samples are L2 hits) +
// 4*n_mels*frames bytes written.  No intermediate ever leaves the CU.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "fe_steps.h"
#include "frontend.h"
#include "layers.h"

#define FE_TB_BYTES ((int)((sizeof(FeTables) + 15) & ~15))

__device__ __forceinline__ int fe_lds_bytes_dev(int fc, int hop) {
    return FE_TB_BYTES + fc * 200 * 8 + ((fc * FE_PSTRIDE * 4 + 15) & ~15) + (((hop * (fc - 1) + FE_NFFT) * 2 + 15) & ~15);
}

int fe_lds_bytes(int fc, int hop) {
    return FE_TB_BYTES + fc * 200 * 8 + ((fc * FE_PSTRIDE * 4 + 15) & ~15) + (((hop * (fc - 1) + FE_NFFT) * 2 + 15) & ~15);
}

// 8 consecutive padded-signal samples starting at s (relative to sample 0 of the clip) as one 16-byte value:
// a single coalesced load when the group is interior and aligned, otherwise 2-byte loads through the reflect map.
__device__ __forceinline__ uint4 fe_fetch8(const int16_t* __restrict__ x, int s, int N, int count) {
    const int16_t* src = x + s;
    if (s >= 0 && s + 8 <= N && count == 8 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0))
        return *reinterpret_cast<const uint4*>(src);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < count) w[e >> 1] |= (uint32_t)(uint16_t)x[fe_reflect(s + e, N)] << (16 * (e & 1));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

#define FE_MAXG 4   // 16-byte groups per lane per chunk held in registers for the next chunk

__global__ void __launch_bounds__(256)
fe_stft_mel_db_kernel(const int16_t* __restrict__ pcm, size_t row_stride, int B, int N, int T, int nchunks, int fc,
                      int hop, int pad, int n_mels, float amin, float db_mult,
                      const FeTables* __restrict__ gtb, float* __restrict__ out_db,
                      float* __restrict__ out_mel, int frames_major, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FeTables* tb = reinterpret_cast<FeTables*>(smem);
    nww_c32* yz = reinterpret_cast<nww_c32*>(smem + FE_TB_BYTES);
    float* pw = reinterpret_cast<float*>(smem + FE_TB_BYTES + fc * 1600);
    int16_t* span = reinterpret_cast<int16_t*>(smem + FE_TB_BYTES + fc * 1600 + ((fc * FE_PSTRIDE * 4 + 15) & ~15));

    const int tid = threadIdx.x, nthr = blockDim.x;
    {   // tables -> LDS (8 KB, once per workgroup; the workgroup then walks chunks persistently)
        const uint4* src = reinterpret_cast<const uint4*>(gtb);
        uint4* dst = reinterpret_cast<uint4*>(tb);
        for (int i = tid; i < FE_TB_BYTES / 16; i += nthr) dst[i] = src[i];
    }
    const int total = B * nchunks;
    // Prefetch pipeline: while chunk i is being transformed, chunk i+1's PCM (incl. reflected edge samples) is
    // already in flight into registers; it is written to the LDS span at the top of the next iteration.
    uint4 pre[FE_MAXG];
    auto chunk_geom = [&](int work, int& b, int& t0, int& nf, int& len, int& s0) {
        b = work / nchunks;
        const int c = work - b * nchunks;
        t0 = c * fc;
        nf = min(fc, T - t0);
        len = hop * (nf - 1) + FE_NFFT;
        s0 = hop * t0 - pad;
    };
    auto prefetch = [&](int work) {
        int b, t0, nf, len, s0;
        chunk_geom(work, b, t0, nf, len, s0);
        const int16_t* x = pcm + (size_t)b * row_stride;
#pragma unroll
        for (int q = 0; q < FE_MAXG; ++q) {
            const int i0 = (tid + q * nthr) * 8;
            if (i0 < len) pre[q] = fe_fetch8(x, s0 + i0, N, min(8, len - i0));
        }
    };
    if ((int)blockIdx.x < total) prefetch(blockIdx.x);
    for (int work = blockIdx.x; work < total; work += gridDim.x) {
        int b, t0, nf, len, s0;
        chunk_geom(work, b, t0, nf, len, s0);
        // ---- S0: registers -> LDS span (16 B per lane per group)
#pragma unroll
        for (int q = 0; q < FE_MAXG; ++q) {
            const int i0 = (tid + q * nthr) * 8;
            if (i0 < len) *reinterpret_cast<uint4*>(span + i0) = pre[q];
        }
        __syncthreads();
        if (work + (int)gridDim.x < total) prefetch(work + gridDim.x);
        // ---- S1
        for (int task = (dbg & 1) ? nf * 25 : tid; task < nf * 25; task += nthr) {
            const int f = task / 25;
            fe_s1(f, task - f * 25, hop, span, tb, yz);
        }
        __syncthreads();
        // ---- S2
        for (int task = (dbg & 2) ? nf * 8 : tid; task < nf * 8; task += nthr) fe_s2(task >> 3, task & 7, yz);
        __syncthreads();
        // ---- S3
        if ((nthr & 127) == 0) {
            // lane -> bin k (fixed for the whole chunk: positions and twiddle live in registers), frames strided over
            // the nthr/128 lane groups; 101 of 128 lanes busy, ~20 VALU per (frame, bin) instead of 46
            const int k3 = tid & 127;
            if (k3 <= 100 && !(dbg & 4)) {
                const int ia = fe_zpos(k3), ib = fe_zpos(k3 ? 200 - k3 : 0);
                const nww_c32 tw = tb->tw400[k3];
                for (int f = tid >> 7; f < nf; f += nthr >> 7) {
                    const nww_c32* zf = yz + f * 200;
                    float* p = pw + f * FE_PSTRIDE;
                    fe_s3_core(zf[ia], zf[ib], tw, &p[k3], &p[200 - k3]);
                }
            }
        } else {
            for (int task = (dbg & 4) ? nf * 101 : tid; task < nf * 101; task += nthr) {
                const int f = task / 101;
                fe_s3(f, task - f * 101, tb, yz, pw);
            }
        }
        __syncthreads();
        // ---- S4: sparse mel + dB.  Tasks are filter-major (lane = frame of one filter) so all lanes of a wave run
        // the same trip count, the filter weights are LDS broadcasts and the power rows are conflict-free.
        const int ntask = (dbg & 8) ? 0 : nf * n_mels;
        if (frames_major) {          // out[b][t][j]: stage [f][j] in LDS (yz is free after S3), then store coalesced
            float* stage = reinterpret_cast<float*>(yz);
            float* stage_m = stage + fc * n_mels;
            if (fc <= 16) {
                // lane -> (filter slot jj = lane>>4, frame f = lane&15): the frame's row pointer is fixed, a wave walks
                // filter quads 4*(wave + nw*it) .. +3 - no division, 4 filters (similar tap counts) per wave instruction
                const int f = tid & 15, jj = (tid >> 4) & 3, wv = tid >> 6, nwv = nthr >> 6;
                if (f < nf && ntask) {
                    for (int j = 4 * wv + jj; j < n_mels; j += 4 * nwv) {
                        const float m = fe_s4(f, j, tb, pw);
                        stage[f * n_mels + j] = fe_db(m, amin, db_mult);
                        if (out_mel) stage_m[f * n_mels + j] = m;
                    }
                }
            } else {
                for (int task = tid; task < ntask; task += nthr) {
                    const int j = task / nf, f = task - j * nf;
                    const float m = fe_s4(f, j, tb, pw);
                    stage[f * n_mels + j] = fe_db(m, amin, db_mult);
                    if (out_mel) stage_m[f * n_mels + j] = m;
                }
            }
            __syncthreads();
            float* ob = out_db ? out_db + ((size_t)b * T + t0) * n_mels : nullptr;
            float* om = out_mel ? out_mel + ((size_t)b * T + t0) * n_mels : nullptr;
            for (int k = tid; k < ntask; k += nthr) {
                if (ob) ob[k] = stage[k];
                if (om) om[k] = stage_m[k];
            }
            __syncthreads();         // stage aliases yz: the next chunk's S1 must not overwrite it early
        } else {                     // out[b][j][t]
            for (int task = tid; task < ntask; task += nthr) {
                const int j = task / nf, f = task - j * nf;
                const float m = fe_s4(f, j, tb, pw);
                const size_t o = ((size_t)b * n_mels + j) * T + t0 + f;
                if (out_db) out_db[o] = fe_db(m, amin, db_mult);
                if (out_mel) out_mel[o] = m;
            }
        }
        // no barrier needed here: the next iteration's S0 writes only `span` (last read in S1) and
        // its S1 writes `yz` (last read in S3, two barriers ago); `pw` is rewritten after 3 barriers.
    }
}

// Choose frames-per-chunk so chunks are balanced: nchunks = ceil(T/FC_MAX), fc = ceil(T/nchunks).
void fe_plan(int T, int fc_max, int* fc, int* nchunks) {
    int nc = (T + fc_max - 1) / fc_max;
    if (nc < 1) nc = 1;
    *nchunks = nc;
    *fc = (T + nc - 1) / nc;
}

hipError_t fe_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p, const FeTables* d_tables,
                     float* d_db, float* d_mel, int frames_major, int fc_max, int block, int max_grid,
                     hipStream_t stream) {
    int fc, nchunks;
    fe_plan(T, fc_max, &fc, &nchunks);
    // the register prefetch holds FE_MAXG 16-byte groups per lane: shrink the chunk if its span would not fit
    while (fc > 1 && (p.hop * (fc - 1) + FE_NFFT + 7) / 8 > FE_MAXG * block) { --fc; nchunks = (T + fc - 1) / fc; }
    if ((p.hop * (fc - 1) + FE_NFFT + 7) / 8 > FE_MAXG * block) return hipErrorInvalidValue;
    const int lds = fe_lds_bytes(fc, p.hop);
    {
        hipError_t e = nww_allow_lds(reinterpret_cast<const void*>(fe_stft_mel_db_kernel), (size_t)lds);
        if (e != hipSuccess) return e;
    }
    static const int dbg = [] { const char* e = getenv("NWW_FE_DBG"); return e ? atoi(e) : 0; }();   // ablation only
    long long total = (long long)B * nchunks;
    int grid = (int)(total < max_grid ? total : max_grid);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fe_stft_mel_db_kernel, dim3(grid), dim3(block), lds, stream, d_pcm, row_stride, B, N, T, nchunks, fc,
                       p.hop, p.center ? FE_NFFT / 2 : 0, p.n_mels, p.amin, p.db_mult, d_tables, d_db, d_mel,
                       frames_major, dbg);
    return hipGetLastError();
}
