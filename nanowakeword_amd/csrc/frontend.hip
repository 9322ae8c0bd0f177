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
#include "fe_steps.h"
#include "frontend.h"

#define FE_TB_BYTES ((int)((sizeof(FeTables) + 15) & ~15))

__device__ __forceinline__ int fe_lds_bytes_dev(int fc, int hop) {
    return FE_TB_BYTES + fc * 200 * 8 + ((fc * FE_PSTRIDE * 4 + 15) & ~15) + (((hop * (fc - 1) + FE_NFFT) * 2 + 15) & ~15);
}

int fe_lds_bytes(int fc, int hop) {
    return FE_TB_BYTES + fc * 200 * 8 + ((fc * FE_PSTRIDE * 4 + 15) & ~15) + (((hop * (fc - 1) + FE_NFFT) * 2 + 15) & ~15);
}

__global__ void __launch_bounds__(256)
fe_stft_mel_db_kernel(const int16_t* __restrict__ pcm, int B, int N, int T, int nchunks, int fc,
                      int hop, int pad, int n_mels, float amin, float db_mult,
                      const FeTables* __restrict__ gtb, float* __restrict__ out_db,
                      float* __restrict__ out_mel, int frames_major) {
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
    for (int work = blockIdx.x; work < total; work += gridDim.x) {
        const int b = work / nchunks, c = work - b * nchunks;
        const int t0 = c * fc;
        const int nf = min(fc, T - t0);
        const int len = hop * (nf - 1) + FE_NFFT;
        const int s0 = hop * t0 - pad;
        const int16_t* x = pcm + (size_t)b * N;
        // ---- S0: stage span. 8 samples (16 B) per step.
        for (int g = tid; g * 8 < len; g += nthr) {
            const int i0 = g * 8, s = s0 + i0;
            const int16_t* src = x + s;
            if (s >= 0 && s + 8 <= N && i0 + 8 <= len && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                *reinterpret_cast<uint4*>(span + i0) = *reinterpret_cast<const uint4*>(src);
            } else {
                for (int e = 0; e < 8 && i0 + e < len; ++e) span[i0 + e] = x[fe_reflect(s + e, N)];
            }
        }
        __syncthreads();
        // ---- S1
        for (int task = tid; task < nf * 25; task += nthr) {
            const int f = task / 25;
            fe_s1(f, task - f * 25, hop, span, tb, yz);
        }
        __syncthreads();
        // ---- S2
        for (int task = tid; task < nf * 8; task += nthr) fe_s2(task >> 3, task & 7, yz);
        __syncthreads();
        // ---- S3
        for (int task = tid; task < nf * 101; task += nthr) {
            const int f = task / 101;
            fe_s3(f, task - f * 101, tb, yz, pw);
        }
        __syncthreads();
        // ---- S4 (writes HBM; consecutive lanes -> consecutive addresses in the chosen layout)
        const int ntask = nf * n_mels;
        if (frames_major) {          // out[b][t][j]
            float* ob = out_db ? out_db + ((size_t)b * T + t0) * n_mels : nullptr;
            float* om = out_mel ? out_mel + ((size_t)b * T + t0) * n_mels : nullptr;
            for (int task = tid; task < ntask; task += nthr) {
                const int f = task / n_mels, j = task - f * n_mels;
                const float m = fe_s4(f, j, tb, pw);
                if (ob) ob[task] = fe_db(m, amin, db_mult);
                if (om) om[task] = m;
            }
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

hipError_t fe_launch(const int16_t* d_pcm, int B, int N, int T, const FeParams& p, const FeTables* d_tables,
                     float* d_db, float* d_mel, int frames_major, int fc_max, int block, int max_grid,
                     hipStream_t stream) {
    int fc, nchunks;
    fe_plan(T, fc_max, &fc, &nchunks);
    const int lds = fe_lds_bytes(fc, p.hop);
    static int attr_set_for = 0;
    if (lds > attr_set_for) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fe_stft_mel_db_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set_for = lds;
    }
    long long total = (long long)B * nchunks;
    int grid = (int)(total < max_grid ? total : max_grid);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fe_stft_mel_db_kernel, dim3(grid), dim3(block), lds, stream, d_pcm, B, N, T, nchunks, fc,
                       p.hop, p.center ? FE_NFFT / 2 : 0, p.n_mels, p.amin, p.db_mult, d_tables, d_db, d_mel,
                       frames_major);
    return hipGetLastError();
}
