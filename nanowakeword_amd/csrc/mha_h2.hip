// mha_h2.hip - scaled-dot-product attention core of the Conformer's nn.MultiheadAttention on the binary16 matrix cores by
// two-term operand splitting (the "f16x3" arithmetic, split_h2.h; architectures.py:471-493, F.multi_head_attention_forward:
// q scaled by 1/sqrt(dh), softmax(q k^T) v).  Replaces mha_mfma.hip's v_mfma_f32_32x32x2_f32 core (0.32 ms at B = 2048,
// T = 101, 4 heads of 36: MFMAs 0.17, 52 % of its LDS cycles bank conflicts) under NWW_ARITH_F16X3.
//
// One workgroup (4 waves) = one (clip, head) unit; wave w takes query tile w (32 queries).  Everything transposed, as in
// mha_mfma.hip, so that the probabilities never leave the registers:
//     St [32 keys x 32 queries] = K tile . Qt     A = K rows (LDS, binary16 hi / lo, [key][DHP] + 16 B pad: conflict-free 16-byte
//                                                 fragment reads), B = the lane's query row, split in registers
//     softmax over the keys = over the lane's 64 registers and its partner half-wave (one shuffle)
//     Ot [dh x 32 queries]   += Vt . Pt           A = Vt rows (LDS, [head dim][key], keys PERMUTED inside every 16-block so that the
//                                                 eight k slots of a lane - the keys its B operand holds - are 16 contiguous bytes),
//                                                 B = Pt: registers 8 j .. 8 j + 7 of a score tile ARE a B fragment of 16-key block j
// Scales (powers of two, exact): K and V by the unit's own maxima (one wave reduction + one LDS word each per unit - a unit's
// result depends on nothing but the unit, so batch invariance holds), every query row by its own maximum (a column of St: undone
// per lane), the probabilities by 2^14 (their maximum is exactly 1).  No plan-time bound is involved: q, k, v are Linear outputs
// of the residual stream, whose interval bound grows with every block.
// Per unit and query tile: 4 x DHP/16 x 3 + (T/16) x MT x 3 MFMAs of 32 clocks (78 at T = 101, dh = 36) against 130 of 64.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "split_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef NWW_TRACE      // tools/ubench/mha_trace.hip: s_memtime of workgroup 0's waves at the phase boundaries of its first 16 units
__device__ unsigned long long g_mha_trace[4 * 16 * 8];
#define MHA_STAMP(k) if (blockIdx.x == 0 && mha_unit < 16 && lane == 0) g_mha_trace[(wave * 16 + mha_unit) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define MHA_STAMP(k)
#endif

namespace {

// largest power of two s with m s <= 2^14 (m > 0; a unit of zeros gets 1)
__device__ __forceinline__ float pow2_scale_to_2p14(float m) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;              // zeros, and a non-finite maximum (frexpf's exponent is unspecified there)
    int e;
    (void)frexpf(m, &e);                                       // m = f 2^e, f in [0.5, 1)
    return ldexpf(1.0f, min(14 - e, 100));                     // a denormal-sized maximum must not scale to +inf (0 x inf = NaN downstream)
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Head dims above 40 keep one wave per SIMD (the 64-wide instance needs ~330 registers - its K / V prefetch alone is 64 - and spilled 52 of
// them under the two-waves-per-SIMD budget; its 72 KB of LDS still lets two workgroups share a CU when the registers allow)
template <int DH>
__global__ void __launch_bounds__(256, (DH > 40 ? 1 : 2)) mha_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int units, int T, int D,
                                                        int n_head, float scale, int head_major) {
    constexpr int NKB = (DH + 15) / 16, DHP = 16 * NKB;        // k-blocks of the score product, padded head dim
    constexpr int MT = (DH + 31) / 32;                         // output tiles along the head dim
    constexpr int KROW = DHP * 2 + 16;                         // bytes per K row (an odd number of 16-byte slots)
    constexpr int VROW = 128 * 2 + 16;                         // bytes per Vt row
    constexpr int K_BYTES = 128 * KROW, V_BYTES = 32 * MT * VROW;
    constexpr int NPC = (128 * (DH / 4) + 255) / 256;          // 16-byte pieces per thread and matrix at T = 128
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_h2[];
    unsigned char* const Kh = lds_h2;
    unsigned char* const Kl = Kh + K_BYTES;
    unsigned char* const Vh = Kl + K_BYTES;
    unsigned char* const Vl = Vh + V_BYTES;
    float* const red = reinterpret_cast<float*>(Vl + V_BYTES);   // [2][4] wave maxima of K, V
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int NTq = (T + 31) / 32;
    // rows / columns from T on and head dims from DH on are never written: zero everything once (finite operands for the
    // products whose probability is 0)
    for (int i = tid; i < (2 * K_BYTES + 2 * V_BYTES) / 4; i += 256) reinterpret_cast<uint32_t*>(lds_h2)[i] = 0u;
    const size_t hm_which = (size_t)units * T * DH;            // floats per q / k / v plane (head-major)
    const int pieces = T * (DH / 4);

    // K, V rows of a unit -> registers (4-byte aligned 16-byte pieces); requested one unit ahead, under the previous unit's products
    float4 kr[NPC], vr[NPC];
    auto unit_ptrs = [&](int u, const float*& qb, const float*& kb, const float*& vb) {
        const int b = u / n_head, head = u - b * n_head;
        // head_major: qkv = [q|k|v][unit][T][DH] (lin_x3's qkv store); else the rows of nn.Linear's [clip][T][3 D] output
        qb = head_major ? qkv + (size_t)u * T * DH : qkv + (size_t)b * T * 3 * D + head * DH;
        kb = head_major ? qb + hm_which : qb + D;
        vb = head_major ? kb + hm_which : kb + D;
    };
    const int rstride = head_major ? DH : 3 * D;
    auto fetch_kv = [&](int u) {
        const float *qb, *kb, *vb;
        unit_ptrs(u, qb, kb, vb);
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = tid + 256 * j;
            kr[j] = make_float4(0.f, 0.f, 0.f, 0.f); vr[j] = kr[j];
            if (i < pieces) {
                const int t = i / (DH / 4), c = 4 * (i - t * (DH / 4));
                kr[j] = *reinterpret_cast<const float4*>(kb + (size_t)t * rstride + c);
                vr[j] = *reinterpret_cast<const float4*>(vb + (size_t)t * rstride + c);
            }
        }
    };
    // the thread's share of max |K|, max |V| of the rows in its registers.  Taken (= the loads waited for) BEFORE the previous
    // unit's output stores are issued: hipcc waits for loop-carried loads with s_waitcnt vmcnt(0), which on gfx9 covers stores too
    float mk = 0.0f, mv = 0.0f;
    auto local_max = [&]() {
        mk = 0.0f; mv = 0.0f;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            mk = fmaxf(mk, fmaxf(fmaxf(fabsf(kr[j].x), fabsf(kr[j].y)), fmaxf(fabsf(kr[j].z), fabsf(kr[j].w))));
            mv = fmaxf(mv, fmaxf(fmaxf(fabsf(vr[j].x), fabsf(vr[j].y)), fmaxf(fabsf(vr[j].z), fabsf(vr[j].w))));
        }
    };
    if ((int)blockIdx.x < units) { fetch_kv((int)blockIdx.x); local_max(); }
    [[maybe_unused]] int mha_unit = -1;
    for (int unit = (int)blockIdx.x; unit < units; unit += (int)gridDim.x) {
        ++mha_unit;
        MHA_STAMP(0)
        const int b = unit / n_head, head = unit - b * n_head;
        const float *qb, *kb, *vb;
        unit_ptrs(unit, qb, kb, vb);
        // the lane's query row (dims 16 kb + 8 h + e) is requested first: it lands while K and V are staged
        const int qt = wave;                                   // T <= 128: at most four query tiles
        const int query = 32 * qt + n;
        float qv[NKB][8];
        if (qt < NTq) {
            const float* qrow = qb + (size_t)min(query, T - 1) * rstride;
#pragma unroll
            for (int k2 = 0; k2 < NKB; ++k2)
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const int c = 16 * k2 + 8 * h + 4 * e4;
                    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < DH) q4 = *reinterpret_cast<const float4*>(qrow + c);      // DH % 4 == 0: all four in or out
                    qv[k2][4 * e4] = q4.x; qv[k2][4 * e4 + 1] = q4.y; qv[k2][4 * e4 + 2] = q4.z; qv[k2][4 * e4 + 3] = q4.w;
                }
        }
        // ---- the unit's maxima, then K, V scaled, split and stored
        mk = wave_max(mk); mv = wave_max(mv);
        __syncthreads();                                       // everyone has left the previous unit's rows (and `red`)
        if (lane == 0) { red[wave] = mk; red[4 + wave] = mv; }
        __syncthreads();
        const float sK = pow2_scale_to_2p14(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        const float sV = pow2_scale_to_2p14(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        MHA_STAMP(1)
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = tid + 256 * j;
            if (i < pieces) {
                const int t = i / (DH / 4), c = 4 * (i - t * (DH / 4));
                uint32_t h0, l0, h1, l1;
                nww_split2h(kr[j].x * sK, kr[j].y * sK, h0, l0);
                nww_split2h(kr[j].z * sK, kr[j].w * sK, h1, l1);
                *reinterpret_cast<uint2*>(Kh + t * KROW + c * 2) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(Kl + t * KROW + c * 2) = make_uint2(l0, l1);
                // Vt[m][pos(t)]: key t = 16 J + 8 a + 4 hh + bq sits at position 16 J + 8 hh + 4 a + bq of its row
                const int pos = (t & ~15) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3);
                nww_split2h(vr[j].x * sV, vr[j].y * sV, h0, l0);
                nww_split2h(vr[j].z * sV, vr[j].w * sV, h1, l1);
                unsigned char* vh = Vh + c * VROW + pos * 2;
                unsigned char* vl = Vl + c * VROW + pos * 2;
                *reinterpret_cast<uint16_t*>(vh) = (uint16_t)h0;
                *reinterpret_cast<uint16_t*>(vh + VROW) = (uint16_t)(h0 >> 16);
                *reinterpret_cast<uint16_t*>(vh + 2 * VROW) = (uint16_t)h1;
                *reinterpret_cast<uint16_t*>(vh + 3 * VROW) = (uint16_t)(h1 >> 16);
                *reinterpret_cast<uint16_t*>(vl) = (uint16_t)l0;
                *reinterpret_cast<uint16_t*>(vl + VROW) = (uint16_t)(l0 >> 16);
                *reinterpret_cast<uint16_t*>(vl + 2 * VROW) = (uint16_t)l1;
                *reinterpret_cast<uint16_t*>(vl + 3 * VROW) = (uint16_t)(l1 >> 16);
            }
        }
        MHA_STAMP(2)
        __syncthreads();
        MHA_STAMP(3)
        const bool has_next = unit + (int)gridDim.x < units;
        if (qt >= NTq && has_next) { fetch_kv(unit + (int)gridDim.x); local_max(); }
        if (qt < NTq) {
            // ---- the query row scaled by 1/sqrt(dh) and by the row's own power of two, split (waits for the row: the only loads
            // in flight), THEN the next unit's K, V rows are requested
            float mq = 0.0f;
#pragma unroll
            for (int k2 = 0; k2 < NKB; ++k2)
#pragma unroll
                for (int e = 0; e < 8; ++e) { qv[k2][e] *= scale; mq = fmaxf(mq, fabsf(qv[k2][e])); }
            asm volatile("" ::: "memory");
            if (has_next) fetch_kv(unit + (int)gridDim.x);
            mq = fmaxf(mq, __shfl_xor(mq, 32, 64));
            const float sQ = pow2_scale_to_2p14(mq);
            u32x4 qh[NKB], ql[NKB];
#pragma unroll
            for (int k2 = 0; k2 < NKB; ++k2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t hh, ll;
                    nww_split2h(qv[k2][2 * j] * sQ, qv[k2][2 * j + 1] * sQ, hh, ll);
                    qh[k2][j] = hh; ql[k2][j] = ll;
                }
            // ---- St tiles: three products per k-block, small terms first
            f32x16 st[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] = 0.0f;
                if (kt < NTq) {
                    const unsigned char* kp = Kh + (32 * kt + n) * KROW + 16 * h;
#pragma unroll
                    for (int k2 = 0; k2 < NKB; ++k2) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(kp + 32 * k2), al = *reinterpret_cast<const f16x8*>(kp + K_BYTES + 32 * k2);
                        const f16x8 bh = __builtin_bit_cast(f16x8, qh[k2]), bl = __builtin_bit_cast(f16x8, ql[k2]);
                        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, st[kt], 0, 0, 0);
                        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, st[kt], 0, 0, 0);
                        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, st[kt], 0, 0, 0);
                    }
                }
            }
            // ---- softmax over the keys (register 4g + q of tile kt = key 32 kt + 8 g + 4 half + q), in the exp2 domain.  The raw
            // accumulators are sK sQ times the scores: the maximum is taken on them (the factor is positive), and one fma per score
            // brings it to the true scale, subtracts the maximum and adds 14 - the probabilities leave exp2 already times 2^14, the
            // scale of their binary16 split (the denominator carries the same factor: exact, undone at the end).
            MHA_STAMP(4)
            const float unS = 1.4426950408889634f / (sK * sQ);
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (32 * kt + 32 > T) {                          // (uniform) only the tile that straddles T has keys to mask
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3);
                        st[kt][r] = key < T ? st[kt][r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float off = 14.0f - mx * unS;
            float den = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[kt][r] = __builtin_amdgcn_exp2f(fmaf(st[kt][r], unS, off));
                    den += st[kt][r];
                }
            den += __shfl_xor(den, 32, 64);
            // ---- Ot tiles: per 16-key block the lane's eight probabilities are the B fragment
            f32x16 ot[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[mt][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (32 * kt + 16 * j < T) {                 // uniform: blocks of keys beyond T carry probability 0
                        u32x4 ph, pl;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            uint32_t hh, ll;
                            nww_split2h(st[kt][8 * j + 2 * e2], st[kt][8 * j + 2 * e2 + 1], hh, ll);
                            ph[e2] = hh; pl[e2] = ll;
                        }
                        const f16x8 bh = __builtin_bit_cast(f16x8, ph), bl = __builtin_bit_cast(f16x8, pl);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const unsigned char* vp = Vh + (32 * mt + n) * VROW + (32 * kt + 16 * j + 8 * h) * 2;
                            const f16x8 ah = *reinterpret_cast<const f16x8*>(vp), al = *reinterpret_cast<const f16x8*>(vp + V_BYTES);
                            ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ot[mt], 0, 0, 0);
                            ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ot[mt], 0, 0, 0);
                            ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ot[mt], 0, 0, 0);
                        }
                    }
                }
            MHA_STAMP(5)
            if (has_next) local_max();
            asm volatile("" ::: "memory");
            MHA_STAMP(6)
            // ---- out[query][head dims 32 mt + 8 g + 4 half + 0..3]
            if (query < T) {
                const float inv = 1.0f / (den * sV);             // (den and the products both carry the probabilities' 2^14)
                float* op = out + ((size_t)b * T + query) * D + head * DH;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = 32 * mt + 8 * g + 4 * h;
                        if (c < DH)                              // DH % 4 == 0: the four dims are in or out together
                            *reinterpret_cast<float4*>(op + c) = make_float4(ot[mt][4 * g] * inv, ot[mt][4 * g + 1] * inv,
                                                                             ot[mt][4 * g + 2] * inv, ot[mt][4 * g + 3] * inv);
                    }
            }
        }
    }
}

}  // namespace

#define NWW_MHA_H2_DIMS(X) X(4) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(36) X(40) X(48) X(64)

static size_t mha_h2_lds(int dh) {
    const int dhp = 16 * ((dh + 15) / 16), mt = (dh + 31) / 32;
    return (size_t)2 * 128 * (dhp * 2 + 16) + (size_t)2 * 32 * mt * (128 * 2 + 16) + 64;
}

bool mha_h2_supported(int T, int D, int n_head) {
    if (n_head <= 0 || D % n_head || T > 128 || T < 1 || D % 4) return false;
    switch (D / n_head) {
#define MHA_OK(DHV) case DHV: return mha_h2_lds(DHV) <= 80 * 1024;
        NWW_MHA_H2_DIMS(MHA_OK)
#undef MHA_OK
        default: return false;
    }
}

hipError_t launch_mha_h2(const float* qkv, float* out, int B, int T, int D, int n_head, hipStream_t s, int head_major) {
    if (!mha_h2_supported(T, D, n_head)) return hipErrorInvalidValue;
    const int dh = D / n_head, units = B * n_head;
    if (units <= 0) return hipSuccess;
    const float scale = 1.0f / sqrtf((float)dh);
    const size_t lds = mha_h2_lds(dh);
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int slots = (dh > 40 ? 1 : 2) * cus;                 // two workgroups per CU (one's staging under the other's MFMAs) where the registers allow
    const dim3 grid(units < slots ? units : slots);
#define MHA_GO(DHV)                                                                                                \
    case DHV: {                                                                                                    \
        hipError_t ea = nww_allow_lds(reinterpret_cast<const void*>(mha_h2_kernel<DHV>), lds);                     \
        if (ea != hipSuccess) return ea;                                                                           \
        hipLaunchKernelGGL((mha_h2_kernel<DHV>), grid, dim3(256), lds, s, qkv, out, units, T, D, n_head, scale, head_major); \
        break;                                                                                                     \
    }
    switch (dh) {
        NWW_MHA_H2_DIMS(MHA_GO)
        default: return hipErrorInvalidValue;
    }
#undef MHA_GO
    return hipGetLastError();
}
