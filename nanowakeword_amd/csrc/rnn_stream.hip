// rnn_stream.hip - GRU / LSTM recurrences wider than the register-resident ones (128 < layer_dim <= 256) on the f16 matrix cores, W_hh
// STREAMED from L2 every step (gfx950).
//
// Cell semantics as rnn_x3.hip / layers.hip state them (nanowakeword/modules/architectures.py:129-145, 238-254).  rnn_x3 keeps W_hh in
// registers for all steps; at H = 256 the two binary16 terms of a GRU's W_hh are 786 KB - more than the registers and LDS of a CU together
// (512 + 160 KB) - so the general kernel of layers.hip took these widths at ~4 ms per layer (B = 2048, T = 101).  Here a workgroup owns
// 16 or 32 clips x all hidden units; W_hh is packed once at plan time into the MFMA B-fragment order of the (k-block, wave, gate, column block, term)
// loop nest, so that every fetch is one 16-byte load per lane and 1 KB contiguous per wave, and each step reads the whole pack through a
// register ring RING units ahead of the matrix pipe.  The pack (0.8 - 1 MB) stays in each XCD's 4 MB L2; a CU takes 64 bytes per clock
// from L2, i.e. ~12 k clocks per step against ~9 k clocks of MFMA issue: the kernel is L2-bandwidth bound by design, about 5 us per step.
//
// Arithmetic: the two-term form of rnn_x3 (NP = 3): h x 2^14 and W_hh x w_scale as hi + lo binary16 terms, products lo.hi, hi.lo, hi.hi on
// v_mfma_f32_16x16x32_f16, the scaling back inside the fma that adds b_hh.  Widths between the instances (160, 200 ...) run as the next
// instance with zero rows / columns in the pack; all global addressing uses the real width (rnn_x3.hip PAD).
// Clips are independent rows of every product: results do not depend on batch size or position.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "layers.h"
#include "split_h2.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ float sigmoid_s(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float tanh_s(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * v)); }

constexpr int RS_NB = 2;                                      // 16-wide column blocks per wave and gate: a wave owns 32 hidden units of every gate

// pack: unit (ks, wave, q, bl) -> [term 2][lane 64] uint4; lane (n = lane & 15, g = lane >> 4) holds row q H + 32 wave + 16 bl + n,
// k = 32 ks + 8 g .. + 7 (the B fragment of v_mfma_f32_16x16x32_f16); rows / columns beyond H are zero
__global__ void __launch_bounds__(256) rnn_stream_pack_kernel(const float* __restrict__ w, uint4* __restrict__ out, int G, int H, int HP, float w_scale) {
    const int NWV = HP / 32, KS = HP / 32;
    const long total = (long)KS * NWV * G * RS_NB * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long u = idx >> 6;
    const int bl = (int)(u % RS_NB); u /= RS_NB;
    const int q = (int)(u % G); u /= G;
    const int wv = (int)(u % NWV);
    const int ks = (int)(u / NWV);
    const int n = lane & 15, g = lane >> 4;
    const int j = 32 * wv + 16 * bl + n, k0 = 32 * ks + 8 * g;
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + 2 * e;
        const float a = (j < H && k < H) ? w[(size_t)(q * H + j) * H + k] * w_scale : 0.0f;
        const float b = (j < H && k + 1 < H) ? w[(size_t)(q * H + j) * H + k + 1] * w_scale : 0.0f;
        nww_split2h(a, b, hh[e], ll[e]);
    }
    const size_t unit = (((size_t)ks * NWV + wv) * G + q) * RS_NB + bl;
    out[(unit * 2 + 0) * 64 + lane] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    out[(unit * 2 + 1) * 64 + lane] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// G = 3: GRU (r, z, n), G = 4: LSTM (i, f, g, o); HP = 192 / 256: the instance's width (a.H <= HP the real one); RING: units fetched ahead
// RS_MT: 16-clip row tiles per workgroup.  A step costs the stream of the whole pack (L2 bound, the same for 16 or 32 clips) plus the gate
// arithmetic of its cells, which nothing overlaps (every wave is in the same phase between the step's two barriers) and which is as long as
// the products at these widths: 1 tile while the batch leaves compute units idle (shortest step), 2 beyond that (half the L2 traffic per clip).
// Results do not depend on the choice: a clip's arithmetic is the same in either.
template <int G, int HP, int RING, int RS_MT>
__global__ void __launch_bounds__(64 * (HP / 32)) rnn_stream_kernel(GruArgs a) {
    constexpr int KS = HP / 32, NWV = HP / 32;
    constexpr int LDP = HP + 16;                              // binary16 per LDS row: + 32 bytes - conflict-free fragment reads for the real ds_read_b128 lane groups (rnn_x3.hip)
    constexpr int UPK = G * RS_NB;                            // units per k-block and wave
    constexpr int NU = KS * UPK;                              // units per step and wave
    static_assert(NU % RING == 0, "the ring closes on the step");
    const int HR = a.H;
    const float s_h = 16384.0f, un = 1.0f / (16384.0f * a.w_scale);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    uint16_t* hp = reinterpret_cast<uint16_t*>(smem_s);       // [2 sets][term 2][clips][LDP]: step t reads set t & 1 and writes the other (one barrier per step, as rnn_x3)
    constexpr int PLANE = 16 * RS_MT * LDP * 2;               // bytes per term plane
    constexpr int SET = 2 * PLANE;                            // bytes per set
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * 16 * RS_MT;
    const int j0 = 32 * wave + n;                             // this lane's hidden units j0 + 16 bl
    for (int idx = threadIdx.x; idx < 2 * 2 * 16 * RS_MT * LDP / 2; idx += blockDim.x) reinterpret_cast<uint32_t*>(hp)[idx] = 0u;
    float bh[G][RS_NB];
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int bl = 0; bl < RS_NB; ++bl) bh[q][bl] = j0 + 16 * bl < HR ? a.b_hh[q * HR + j0 + 16 * bl] : 0.0f;
    float hprev[RS_MT][RS_NB][4], cprev[RS_MT][RS_NB][4];
#pragma unroll
    for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
        for (int bl = 0; bl < RS_NB; ++bl)
#pragma unroll
            for (int r = 0; r < 4; ++r) { hprev[mt][bl][r] = 0.0f; cprev[mt][bl][r] = 0.0f; }
    // the wave's slice of the pack: unit u = ks * UPK + (q * NB + bl) at wbase + (ks * NWV * UPK + q * NB + bl) * 128 uint4
    // (a uniform base plus ONE 32-bit lane offset: every fetch is a scalar-base load with a compile-time displacement - 96 separate
    // 64-bit lane addresses would not fit beside the accumulators)
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w_packed);
    uint32_t voff = (uint32_t)((wave * UPK * 128 + lane) * 16);
    auto fetch_unit = [&](int u, uint4& hi, uint4& lo) {
        const unsigned char* pu = wp + ((size_t)(u / UPK) * NWV * UPK + (u % UPK)) * 2048;
        hi = *reinterpret_cast<const uint4*>(pu + voff);
        lo = *reinterpret_cast<const uint4*>(pu + 1024 + voff);
    };
    uint4 ring[RING][2];
#pragma unroll
    for (int u = 0; u < RING; ++u) fetch_unit(u, ring[u][0], ring[u][1]);
    const unsigned char* arow0 = smem_s + (size_t)(n * LDP + 8 * g) * 2;         // A fragment: clip n (+ 16 mt), k = 32 ks + 8 g .. + 7
    // gate pre-activations xg [B][T][G HR]: the workgroup's 32 clips from a uniform base, each of the lane's 8 rows by one 32-bit byte offset
    // that moves by a row per step (scalar base + lane offset loads again; rows beyond B repeat the last clip).  Lanes of padded units read a
    // few floats past their gate's columns - inside the buffer, whose clips carry one row more than T (nww_plan.hip: add_bigru_last) - and
    // the value is discarded.
    const unsigned char* xg_wg = reinterpret_cast<const unsigned char*>(a.xg + (size_t)b0 * a.T * G * HR);
    const int row_bytes = G * HR * 4, t_first = a.reverse ? a.T - 1 : 0, t_delta = a.reverse ? -row_bytes : row_bytes;
    uint32_t xoff[RS_MT][4];
#pragma unroll
    for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = min(b0 + 16 * mt + 4 * g + r, a.B - 1) - b0;
            xoff[mt][r] = (uint32_t)((c * a.T + t_first) * row_bytes + j0 * 4);
        }
    const bool jok[RS_NB] = {j0 < HR, j0 + 16 < HR};
    // the layer's output sequence (when another layer follows): the same addressing, rows of ld_seq floats
    unsigned char* seq_wg = reinterpret_cast<unsigned char*>(a.seq_out ? a.seq_out + (size_t)b0 * a.T * a.ld_seq + a.col_off : nullptr);
    const int s_delta = (a.reverse ? -a.ld_seq : a.ld_seq) * 4;
    uint32_t soff[RS_MT][4];
#pragma unroll
    for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) soff[mt][r] = (uint32_t)(((16 * mt + 4 * g + r) * a.T + t_first) * a.ld_seq * 4 + j0 * 4);
    __syncthreads();
    for (int step = 0; step < a.steps; ++step) {
        // input-side pre-activations of this step, used behind the products.  Requested inside the product loop, behind the last unit whose
        // weights were fetched in the PREVIOUS step: hipcc waits for loads carried round the loop with s_waitcnt vmcnt(0), so a request ahead of
        // those units made every step begin with a trip to HBM (the ablation's "no xq: -2 us per step")
        float xq[RS_MT][G][RS_NB][4];
        auto fetch_xq = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("" : "+v"(xoff[mt][r]));     // (opaque, like voff below: no hoisted lane address pairs)
#pragma unroll
                    for (int q = 0; q < G; ++q)
#pragma unroll
                        for (int bl = 0; bl < RS_NB; ++bl)
#ifdef NWW_ABLATION
                            xq[mt][q][bl][r] = (a.dbg & 4) ? 0.0f : *reinterpret_cast<const float*>(xg_wg + (size_t)(q * HR + 16 * bl) * 4 + xoff[mt][r]);
#else
                            xq[mt][q][bl][r] = *reinterpret_cast<const float*>(xg_wg + (size_t)(q * HR + 16 * bl) * 4 + xoff[mt][r]);
#endif
                    xoff[mt][r] += (uint32_t)t_delta;
                }
        };
        constexpr int XQ_UNIT = RING - 1;                     // the last unit of the step that was fetched in the previous one
        const unsigned char* arow = arow0 + (step & 1) * SET;
        uint16_t* hpw = hp + ((step & 1) ^ 1) * (SET / 2);
        f32x4 acc[RS_MT][G][RS_NB];
#pragma unroll
        for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int bl = 0; bl < RS_NB; ++bl) acc[mt][q][bl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // (the lane offset is made opaque per k-block: addresses derived from it are not loop-invariant, so the compiler forms them here
            // from the scalar base instead of hoisting ~100 lane address pairs out of the step loop and spilling them)
            asm volatile("" : "+v"(voff));
            f16x8 ah[RS_MT], al[RS_MT];
#pragma unroll
            for (int mt = 0; mt < RS_MT; ++mt) {
                ah[mt] = *reinterpret_cast<const f16x8*>(arow + mt * 16 * LDP * 2 + 64 * ks);
                al[mt] = *reinterpret_cast<const f16x8*>(arow + PLANE + mt * 16 * LDP * 2 + 64 * ks);
            }
#pragma unroll
            for (int v = 0; v < UPK; ++v) {
                const int u = ks * UPK + v, slot = u % RING;
                const f16x8 wh = __builtin_bit_cast(f16x8, ring[slot][0]), wl = __builtin_bit_cast(f16x8, ring[slot][1]);
                // refill the slot with the unit RING ahead; behind the last unit of the step that is the next step's head (same pack every step)
#ifdef NWW_ABLATION
                if (!(a.dbg & 1))
#endif
                fetch_unit((u + RING) % NU, ring[slot][0], ring[slot][1]);
                const int q = v / RS_NB, bl = v % RS_NB;
#ifdef NWW_ABLATION
                if (!(a.dbg & 2))
#endif
#pragma unroll
                for (int mt = 0; mt < RS_MT; ++mt) {          // small terms first, the dominant hi*hi last (rnn_x3's order)
                    acc[mt][q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], wh, acc[mt][q][bl], 0, 0, 0);
                    acc[mt][q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], wl, acc[mt][q][bl], 0, 0, 0);
                    acc[mt][q][bl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], wh, acc[mt][q][bl], 0, 0, 0);
                }
                if (u == XQ_UNIT) fetch_xq();
                __builtin_amdgcn_sched_barrier(0);            // keep the stream in order: no fetch rises above the ring's depth (registers)
            }
        }
#pragma unroll
        for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
            for (int bl = 0; bl < RS_NB; ++bl) {
                const int j = j0 + 16 * bl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * mt + 4 * g + r, b = b0 + c;
                    // rows beyond B repeat the last clip (clamped xg row): straight-line gate arithmetic, only the stores are predicated
                    float hn, cn = 0.0f;
                    if constexpr (G == 3) {
                        const float a0 = fmaf(acc[mt][0][bl][r], un, bh[0][bl]);
                        const float a1 = fmaf(acc[mt][1][bl][r], un, bh[1][bl]);
                        const float a2 = fmaf(acc[mt][2][bl][r], un, bh[2][bl]);
                        const float rg = sigmoid_s(xq[mt][0][bl][r] + a0);
                        const float zg = sigmoid_s(xq[mt][1][bl][r] + a1);
                        const float ng = tanh_s(xq[mt][2][bl][r] + rg * a2);
                        hn = (1.0f - zg) * ng + zg * hprev[mt][bl][r];
                    } else {
                        const float ig = sigmoid_s(xq[mt][0][bl][r] + fmaf(acc[mt][0][bl][r], un, bh[0][bl]));
                        const float fg = sigmoid_s(xq[mt][1][bl][r] + fmaf(acc[mt][1][bl][r], un, bh[1][bl]));
                        const float gg = tanh_s(xq[mt][2][bl][r] + fmaf(acc[mt][2][bl][r], un, bh[2][bl]));
                        const float og = sigmoid_s(xq[mt][G - 1][bl][r] + fmaf(acc[mt][G - 1][bl][r], un, bh[G - 1][bl]));
                        cn = fg * cprev[mt][bl][r] + ig * gg;
                        hn = og * tanh_s(cn);
                    }
                    if (!jok[bl]) hn = cn = 0.0f;              // padded unit: its pre-activation loads were not its own
                    if (seq_wg && b < a.B && jok[bl]) *reinterpret_cast<float*>(seq_wg + 64 * bl + soff[mt][r]) = hn;
                    hprev[mt][bl][r] = hn; cprev[mt][bl][r] = cn;
                    uint32_t hh, ll;
                    nww_split2h(hn * s_h, 0.0f, hh, ll);
                    uint16_t* d = hpw + c * LDP + j;
                    d[0] = (uint16_t)hh; d[16 * RS_MT * LDP] = (uint16_t)ll;
                }
            }
#pragma unroll
        for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { asm volatile("" : "+v"(soff[mt][r])); soff[mt][r] += (uint32_t)s_delta; }
        __syncthreads();
    }
    // the direction's last state -> last_out (rnn_out[:, -1]'s half), and the opposite direction's FIRST step beside it (h = c = 0: no recurrent product)
    if (!a.last_out) return;
#pragma unroll
    for (int mt = 0; mt < RS_MT; ++mt)
#pragma unroll
        for (int bl = 0; bl < RS_NB; ++bl) {
            const int j = j0 + 16 * bl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + 16 * mt + 4 * g + r;
                if (b >= a.B || !jok[bl]) continue;
                a.last_out[(size_t)b * a.ld_last + a.col_off + j] = hprev[mt][bl][r];
                if (!a.xg2) continue;
                const float* x2 = a.xg2 + (size_t)b * a.xg2_bstride + j;
                float h2;
                if (G == 3) {
                    const float rg = sigmoid_s(x2[0] + a.b_hh2[j]);
                    const float zg = sigmoid_s(x2[HR] + a.b_hh2[HR + j]);
                    const float ng = tanh_s(x2[2 * HR] + rg * a.b_hh2[2 * HR + j]);
                    h2 = (1.0f - zg) * ng;
                } else {
                    const float ig = sigmoid_s(x2[0] + a.b_hh2[j]);
                    const float gg = tanh_s(x2[2 * HR] + a.b_hh2[2 * HR + j]);
                    const float og = sigmoid_s(x2[(G - 1) * HR] + a.b_hh2[(G - 1) * HR + j]);
                    h2 = og * tanh_s(ig * gg);
                }
                a.last_out[(size_t)b * a.ld_last + a.col_off2 + j] = h2;
            }
        }
}
}  // namespace

static int rnn_stream_width(int H) { return H <= 192 ? 192 : 256; }

// 128 < H <= 256, a multiple of 4 (the xg rows' alignment), two-term form
bool rnn_stream_usable(const GruArgs& a) {
    static const int on = 1;
    return on && a.products == 3 && a.H > 128 && a.H <= 256 && a.H % 4 == 0 && a.fin == 0;
}

size_t rnn_stream_packed_bytes(int gates, int H) {
    const int HP = rnn_stream_width(H);
    return (size_t)(HP / 32) * (HP / 32) * gates * RS_NB * 2 * 64 * sizeof(uint4);
}

hipError_t launch_rnn_stream_pack(const float* w_hh, void* packed, int gates, int H, float w_scale, hipStream_t s) {
    const int HP = rnn_stream_width(H);
    const long total = (long)(HP / 32) * (HP / 32) * gates * RS_NB * 64;
    hipLaunchKernelGGL(rnn_stream_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_hh, reinterpret_cast<uint4*>(packed), gates, H, HP, w_scale);
    return hipGetLastError();
}

hipError_t launch_rnn_stream(const GruArgs& a, int gates, hipStream_t s) {
    if (!rnn_stream_usable(a) || !a.w_packed || (gates != 3 && gates != 4) || !(a.w_scale > 0.0f)) return hipErrorInvalidValue;
    const int HP = rnn_stream_width(a.H);
    static const int force_mt = 0;
    static const int n_cu = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n; }();
    const int mt = force_mt == 1 || force_mt == 2 ? force_mt : (a.B <= 16 * n_cu ? 1 : 2);
    const dim3 grid((a.B + 16 * mt - 1) / (16 * mt)), block(64 * (HP / 32));
    const size_t lds = (size_t)2 * 2 * 16 * mt * (HP + 16) * sizeof(uint16_t);     // two sets of two term planes
#ifdef NWW_ABLATION
    GruArgs ad = a;
    { const char* e = getenv("NWW_RNN_DBG"); ad.dbg = e ? atoi(e) : 0; }
#define a ad
#endif
    // ring depths: what fits the 256 registers of a wave (two per SIMD) without scratch beside 24 / 48 (GRU) or 32 / 64 (LSTM) accumulators
#define RS_GO(GV, HV, R1, R2)                                                                            \
    {                                                                                                    \
        const void* fn = mt == 1 ? reinterpret_cast<const void*>(rnn_stream_kernel<GV, HV, R1, 1>)      \
                                 : reinterpret_cast<const void*>(rnn_stream_kernel<GV, HV, R2, 2>);      \
        const hipError_t ea = nww_allow_lds(fn, lds);           /* 67.6 KB at HP = 256, two tiles */        \
        if (ea != hipSuccess) return ea;                                                                 \
        if (mt == 1) hipLaunchKernelGGL((rnn_stream_kernel<GV, HV, R1, 1>), grid, block, lds, s, a);     \
        else hipLaunchKernelGGL((rnn_stream_kernel<GV, HV, R2, 2>), grid, block, lds, s, a);             \
    }
    if (gates == 3) {
        if (HP == 192) { RS_GO(3, 192, 12, 4) } else { RS_GO(3, 256, 12, 4) }
    } else {
        if (HP == 192) { RS_GO(4, 192, 8, 2) } else { RS_GO(4, 256, 8, 2) }
    }
#undef RS_GO
#ifdef NWW_ABLATION
#undef a
#endif
    return hipGetLastError();
}
