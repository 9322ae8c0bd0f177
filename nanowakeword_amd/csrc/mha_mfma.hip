// mha_mfma.hip - scaled-dot-product attention core of the Conformer's nn.MultiheadAttention on the matrix cores
// (architectures.py:471-493, torch.nn.functional.multi_head_attention_forward: q scaled by 1/sqrt(dh), softmax(q k^T) v),
// float32 in, float32 products and sums (v_mfma_f32_32x32x2_f32), replacing the one-lane-per-query VALU kernel
// (layers.hip: mha_core_kernel, 0.50 ms at B = 2048, T = 101, 4 heads of 36).
//
// One workgroup = UW (clip, head) units of two waves (UW = 2 for the reference's 36-wide heads: 74 KB of LDS, two
// workgroups per CU, so that one's row fetch runs under the other's MFMAs); a workgroup walks groups of UW units.  A
// unit's K and V rows ([128][dh], dense, zero from row T on) go global -> LDS directly (global_load_lds_dwordx4); a wave
// takes the query tiles (32 queries) qt = w, w + 2.  Both products are computed transposed so that the probabilities
// never leave the registers:
//     St [32 keys x 32 queries] = K tile . Qt        A = K[key][2s + half] (LDS), B = q[2s + half] of the lane's query (registers)
//     softmax over the keys = over the lane's registers of the four key tiles and its partner half-wave (one shuffle)
//     Ot [dh x 32 queries]     += Vt . Pt            A = V[key][m] (LDS); B = Pt: lane (query, half) of the C layout holds key
//                                                    8g + 4 half + q in register 4g + q - exactly a B operand whose k slot `half` is
//                                                    that key, so step (tile, register r) multiplies V rows key(r, 0), key(r, 1)
// and the output rows of a lane are 4 consecutive head dims of its query: 16-byte stores.  Steps whose keys are all >= T
// are skipped.  Per unit and query tile: 4 x dh / 2 + dh_tiles x (valid key pairs) MFMAs of 64 clocks.
// Measured (B = 2048, T = 101, 4 heads of 36): 0.50 ms (VALU kernel) -> 0.32 ms.  Ablation: the MFMAs account for 0.17 ms
// (their nominal 0.15), the row traffic (K, V in, Q in, out: 476 MB as 144-byte row pieces) for the rest; variants measured
// and dropped: one workgroup of four units per CU with the copy as its own phase 0.35 ms; double-buffered LDS with one
// wave per SIMD 0.57 ms (a lone wave's dependent float32-MFMA chains leave the pipe idle); next group's rows prefetched
// into registers: spills; the key rows read 16 bytes at a time (conflict-free, where the dword reads at stride 36 hit 8 banks
// four ways - 52 % of this kernel's LDS cycles are conflict cycles): 0.318 vs 0.316 ms, the conflicts hide under the MFMAs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int DH, int UW>
__global__ void __launch_bounds__(128 * UW, UW == 2 ? 2 : 1) mha_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int units, int T, int D,
                                                       int n_head, float scale, int head_major) {
    constexpr int LD = DH;                                     // LDS rows are dense: the rows arrive by LDS-DMA in 16-byte pieces
    constexpr int MT = (DH + 31) / 32;                         // output tiles along the head dim
    constexpr int REGION = 2 * 128 * DH;                       // floats of one unit's K [128][DH] + V [128][DH]
    constexpr int NDMA = (128 * (DH / 4) + 127) / 128;         // 16-byte pieces per thread and matrix, T = 128
    extern __shared__ __attribute__((aligned(16))) float lds_mha[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = wave >> 1, w = wave & 1;                    // unit inside the workgroup, wave inside the unit
    const int n = lane & 31, h = lane >> 5;
    const int NT = (T + 31) / 32;
    const int groups = (units + UW - 1) / UW;
    // rows from T on are never written: zero them (all of LDS) once - V rows up to the end of the last key step are
    // multiplied by probability 0 and must be finite
    for (int i = tid; i < UW * REGION; i += 128 * UW) lds_mha[i] = 0.0f;
    __syncthreads();
    // K, V rows of a unit go global -> LDS directly (global_load_lds_dwordx4, lane i of a wave lands at base + 16 i), into
    // the buffer the workgroup is NOT computing on: as a separate phase through registers the copy ran at 2.7 TB/s and took
    // half of the kernel's time.
    const int t2 = tid & 127, pieces = T * (DH / 4);
    // head_major: qkv = [q|k|v][clip][head][T][DH] (lin_x3's qkv store) - a unit's K and V are one contiguous run each;
    // else the rows of nn.Linear's [clip][T][3 D] output
    const size_t hm_which = (size_t)units * T * DH;            // floats per q / k / v plane
    auto fetch_kv = [&](int g, int buf) {
        const int u = g * UW + ul;
        if (u >= units) return;
        const int ub = u / n_head, uh = u - ub * n_head;
        const float* src = head_major ? qkv + hm_which + (size_t)u * T * DH : qkv + (size_t)ub * T * 3 * D + uh * DH;
        float* dst = lds_mha + (size_t)(buf * UW + ul) * REGION + w * 256;      // + 64 lanes x 4 floats per wave; wave-uniform
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int i = t2 + 128 * j;
            if (i < pieces) {
                const int t = i / (DH / 4), c = 4 * (i - t * (DH / 4));
                const float* p = head_major ? src + 4 * i : src + (size_t)t * 3 * D + D + c;
                const float* pv = head_major ? p + hm_which : p + D;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p,
                                                 (void __attribute__((address_space(3)))*)(dst + 512 * j), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)pv,
                                                 (void __attribute__((address_space(3)))*)(dst + 128 * DH + 512 * j), 16, 0, 0);
            }
        }
    };
    constexpr int buf = 0;
    for (int grp = (int)blockIdx.x; grp < groups; grp += (int)gridDim.x) {
    __syncthreads();                                           // everyone has left the previous group's rows
    fetch_kv(grp, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int unit = grp * UW + ul;
    if (unit < units) {
    const int b = unit / n_head, head = unit - b * n_head;
    const float* base = head_major ? qkv + (size_t)unit * T * DH : qkv + (size_t)b * T * 3 * D + head * DH;
    const int qstride = head_major ? DH : 3 * D;
    const float* Ks = lds_mha + (size_t)(buf * UW + ul) * REGION;
    const float* Vs = Ks + 128 * DH;
    for (int qt = w; qt < NT; qt += 2) {
        // ---- the lane's query row, scaled; qs[s] = element 2s + half (the B operand of step s)
        const int query = 32 * qt + n;
        const float* qrow = base + (size_t)min(query, T - 1) * qstride;
        float qs[DH / 2];
#pragma unroll
        for (int c4 = 0; c4 < DH / 4; ++c4) {
            const float4 q4 = *reinterpret_cast<const float4*>(qrow + 4 * c4);
            qs[2 * c4] = (h ? q4.y : q4.x) * scale;
            qs[2 * c4 + 1] = (h ? q4.w : q4.z) * scale;
        }
        // ---- St tiles
        f32x16 st[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.0f;
            if (kt < NT) {
                const float* kp = Ks + (32 * kt + n) * LD + h;
#pragma unroll
                for (int s = 0; s < DH / 2; ++s) st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qs[s], st[kt], 0, 0, 0);
            }
        }
        // ---- softmax over the keys (register 4g + q of tile kt = key 32 kt + 8 g + 4 half + q)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3);
                st[kt][r] = key < T ? st[kt][r] : -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kt][r] = __builtin_amdgcn_exp2f((st[kt][r] - mx) * 1.4426950408889634f);
                den += st[kt][r];
            }
        den += __shfl_xor(den, 32, 64);
        const float inv = 1.0f / den;
        // ---- Ot tiles
        f32x16 ot[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[mt][r] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key0 = 32 * kt + 8 * (r >> 2) + (r & 3);           // the half-0 key of this step; half 1 is key0 + 4
                if (key0 < T) {                                              // uniform: keys beyond T carry probability 0
                    const float* vp = Vs + (key0 + 4 * h) * LD + n;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        ot[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * mt], st[kt][r], ot[mt], 0, 0, 0);
                }
            }
        // ---- out[query][head dims 32 mt + 8 g + 4 half + 0..3]
        if (query < T) {
            float* op = out + ((size_t)b * T + query) * D + head * DH;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 32 * mt + 8 * g + 4 * h;
                    if (c < DH)                                              // DH % 4 == 0: the four dims are in or out together
                        *reinterpret_cast<float4*>(op + c) = make_float4(ot[mt][4 * g] * inv, ot[mt][4 * g + 1] * inv,
                                                                         ot[mt][4 * g + 2] * inv, ot[mt][4 * g + 3] * inv);
                }
        }
    }
    }                                                          // unit < units
    }                                                          // groups
}

}  // namespace

#define NWW_MHA_MFMA_DIMS(X) X(4) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(36) X(40) X(48) X(64)

static size_t mha_mfma_lds(int dh, int uw) { return (size_t)uw * 2 * 128 * dh * sizeof(float); }
static int mha_mfma_units(int dh) { return mha_mfma_lds(dh, 4) <= 80 * 1024 ? 4 : mha_mfma_lds(dh, 2) <= 80 * 1024 ? 2 : mha_mfma_lds(dh, 1) <= 160 * 1024 ? 1 : 0; }

bool mha_mfma_supported(int T, int D, int n_head) {
    if (n_head <= 0 || D % n_head || T > 128 || T < 1 || D % 4) return false;
    switch (D / n_head) {
#define MHA_OK(DHV) case DHV: return mha_mfma_units(DHV) != 0;
        NWW_MHA_MFMA_DIMS(MHA_OK)
#undef MHA_OK
        default: return false;
    }
}

hipError_t launch_mha_mfma(const float* qkv, float* out, int B, int T, int D, int n_head, hipStream_t s, int head_major) {
    if (!mha_mfma_supported(T, D, n_head)) return hipErrorInvalidValue;
    const int dh = D / n_head, units = B * n_head;
    if (units <= 0) return hipSuccess;
    const float scale = 1.0f / sqrtf((float)dh);
    const int uw = mha_mfma_units(dh);
    const size_t lds = mha_mfma_lds(dh, uw);
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int groups = (units + uw - 1) / uw;
    const int slots = 2 * cus;                                 // two workgroups per CU: one's row fetch under the other's MFMAs
    const dim3 grid(groups < slots ? groups : slots);
#define MHA_GO(DHV, UWV)                                                                                           \
    {                                                                                                              \
        hipError_t ea = nww_allow_lds(reinterpret_cast<const void*>(mha_mfma_kernel<DHV, UWV>), lds);              \
        if (ea != hipSuccess) return ea;                                                                           \
        hipLaunchKernelGGL((mha_mfma_kernel<DHV, UWV>), grid, dim3(128 * UWV), lds, s, qkv, out, units, T, D, n_head, scale, head_major); \
    }
#define MHA_CASE(DHV) case DHV: if (uw == 4) MHA_GO(DHV, 4) else if (uw == 2) MHA_GO(DHV, 2) else MHA_GO(DHV, 1) break;
    switch (dh) {
        NWW_MHA_MFMA_DIMS(MHA_CASE)
        default: return hipErrorInvalidValue;
    }
#undef MHA_CASE
#undef MHA_GO
    return hipGetLastError();
}
