// fe_steps.h - per-lane task bodies of the fused frontend kernel (framing -> Hann -> 400-point
// real FFT -> |X|^2 -> sparse mel -> 10 log10).  Written as host+device inline functions so the
// same arithmetic is compiled by hipcc into the gfx950 kernel (frontend2.hip) and by g++ into the
// CPU emulator used by the non-GPU tests (tests/hostemu/): index maps, twiddles and butterflies are
// validated without a GPU.
//
// Replaces, for one frame x[0..399] (already scaled by 1/32768):
//   conv1d(x, real_basis) / conv1d(x, imag_basis), real^2+imag^2, matmul(., mel_fb)
//   (reference: nanowakeword/_export/onnx.py:66-83) followed by AmplitudeToDB
//   (nanowakeword/modules/architectures.py:837,875).
//
// Algorithm.  N = 400 real points -> M = 200 complex z[m] = x[2m] + i x[2m+1].
//   FFT200 = 8 x 25 Cooley-Tukey:  m = 25 n1 + n2,  k = k1 + 8 k2
//     S1  (25 tasks/frame): Y[k1][n2] = W200^(n2 k1) * sum_n1 z[25 n1 + n2] W8^(n1 k1)
//     S2  ( 8 tasks/frame): Z[k1 + 8 k2] = sum_n2 Y[k1][n2] W25^(n2 k2)      (25 = 5 x 5 in registers)
//     S3 (101 tasks/frame): E = (Z[k] + conj Z[200-k])/2, O = -i (Z[k] - conj Z[200-k])/2,
//                           P[k] = |E + w_k O|^2, P[200-k] = |E - w_k O|^2,  w_k = exp(-2 pi i k/400)
//     S4 (n_mels tasks/frame): mel[j] = sum_{k in support(j)} P[k] fb[k][j]; dB = mult*log10(max(mel, amin))
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NWW_HD __host__ __device__ __forceinline__
typedef float2 nww_c32;
#else
#define NWW_HD inline
struct nww_c32 { float x, y; };
#endif

#define FE_NFFT 400
#define FE_M 200          // complex points
#define FE_BINS 201
#define FE_PSTRIDE 209    // odd stride of the power rows in LDS (bank spread)
#define FE_MAX_MELS 128
#define FE_MAX_MELW 640   // >= total nonzeros of the filterbank (each bin feeds <= 2 filters, + slack)

NWW_HD nww_c32 c_make(float x, float y) { nww_c32 r; r.x = x; r.y = y; return r; }
NWW_HD nww_c32 c_add(nww_c32 a, nww_c32 b) { return c_make(a.x + b.x, a.y + b.y); }
NWW_HD nww_c32 c_sub(nww_c32 a, nww_c32 b) { return c_make(a.x - b.x, a.y - b.y); }
NWW_HD nww_c32 c_mul(nww_c32 a, nww_c32 b) { return c_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
NWW_HD nww_c32 c_mul_negi(nww_c32 a) { return c_make(a.y, -a.x); }   // a * (-i)
NWW_HD nww_c32 c_scale(nww_c32 a, float s) { return c_make(a.x * s, a.y * s); }

// Tables, built on the host in double precision (fe_tables.cpp), resident in LDS in the kernel.
struct FeTables {
    nww_c32 win2[FE_M];          // (w[2m], w[2m+1]) / 65536  (int16 -> unit scale and fe_s3_core's 1/2 folded in; exact, powers of 2)
    nww_c32 tw200[8 * 25];       // [k1][n2] = exp(-2 pi i n2 k1 / 200)  (lane index n2 contiguous: no bank conflicts)
    nww_c32 tw400[101];          // exp(-2 pi i k / 400), k = 0..100
    int32_t mel_lo[FE_MAX_MELS];   // first FFT bin of filter j
    int32_t mel_cnt[FE_MAX_MELS];  // number of bins in its support
    int32_t mel_off[FE_MAX_MELS];  // offset of its weights in melw
    float melw[FE_MAX_MELW];
};

// ---- radix-8 forward DFT (decimation in frequency), in place: y[k] = sum_n z[n] exp(-2 pi i n k/8)
NWW_HD void dft8(nww_c32 z[8]) {
    const float r = 0.70710678118654752440f;
    nww_c32 a0 = c_add(z[0], z[4]), a1 = c_add(z[1], z[5]), a2 = c_add(z[2], z[6]), a3 = c_add(z[3], z[7]);
    nww_c32 b0 = c_sub(z[0], z[4]);
    nww_c32 d1 = c_sub(z[1], z[5]), d2 = c_sub(z[2], z[6]), d3 = c_sub(z[3], z[7]);
    nww_c32 b1 = c_make((d1.x + d1.y) * r, (d1.y - d1.x) * r);      // d1 * (1 - i)/sqrt2
    nww_c32 b2 = c_mul_negi(d2);                                     // d2 * (-i)
    nww_c32 b3 = c_make((d3.y - d3.x) * r, -(d3.x + d3.y) * r);     // d3 * (-1 - i)/sqrt2
    // even outputs: DFT4(a)
    nww_c32 p0 = c_add(a0, a2), p1 = c_sub(a0, a2), q0 = c_add(a1, a3), q1 = c_mul_negi(c_sub(a1, a3));
    z[0] = c_add(p0, q0); z[4] = c_sub(p0, q0); z[2] = c_add(p1, q1); z[6] = c_sub(p1, q1);
    // odd outputs: DFT4(b)
    nww_c32 s0 = c_add(b0, b2), s1 = c_sub(b0, b2), t0 = c_add(b1, b3), t1 = c_mul_negi(c_sub(b1, b3));
    z[1] = c_add(s0, t0); z[5] = c_sub(s0, t0); z[3] = c_add(s1, t1); z[7] = c_sub(s1, t1);
}

// ---- radix-5 forward DFT: X[k] = sum_n x[n] exp(-2 pi i n k/5)
NWW_HD void dft5(nww_c32& x0, nww_c32& x1, nww_c32& x2, nww_c32& x3, nww_c32& x4) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;   // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;    // sin(2pi/5), sin(4pi/5)
    nww_c32 a1 = c_add(x1, x4), a2 = c_add(x2, x3), b1 = c_sub(x1, x4), b2 = c_sub(x2, x3);
    nww_c32 m1 = c_make(x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y);
    nww_c32 m2 = c_make(x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y);
    nww_c32 n1 = c_make(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    nww_c32 n2 = c_make(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    nww_c32 j1 = c_mul_negi(n1), j2 = c_mul_negi(n2);
    x0 = c_make(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    x1 = c_add(m1, j1); x4 = c_sub(m1, j1);
    x2 = c_add(m2, j2); x3 = c_sub(m2, j2);
}

// exp(-2 pi i p / 25) for p = 0..16 (products b*c with b,c in 0..4)
NWW_HD nww_c32 w25(int p) {
    const float C[17] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f,
                         0.53582679497899666f, 0.30901699437494742f, 0.06279051952931337f,
                         -0.18738131458572463f, -0.42577929156507272f, -0.63742398974868975f,
                         -0.80901699437494742f, -0.92977648588825146f, -0.99211470131447788f,
                         -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494742f,
                         -0.63742398974868975f};
    const float S[17] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868873f,
                         0.84432792550201508f, 0.95105651629515357f, 0.99802672842827156f,
                         0.98228725072868872f, 0.90482705246601958f, 0.77051324277578925f,
                         0.58778525229247313f, 0.36812455268467797f, 0.12533323356430426f,
                         -0.12533323356430426f, -0.36812455268467797f, -0.58778525229247313f,
                         -0.77051324277578925f};
    return c_make(C[p], -S[p]);
}

// ---- 25-point forward DFT in registers: n = 5a + b, k = c + 5d.
//   U[b][c] = W25^(b c) sum_a y[5a+b] W5^(a c);  Z[c+5d] = sum_b U[b][c] W5^(b d)
#if defined(__HIP_DEVICE_COMPILE__)
#define NWW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NWW_SCHED_FENCE() ((void)0)
#endif
// SEQ: keep the five row transforms (and the five column transforms) in program order on the GPU - without it the
// scheduler interleaves them for ILP and the 25-point transform alone takes ~170 VGPRs; in order it needs the 50 of
// the data plus one radix-5 butterfly's temporaries (same arithmetic either way).
template <bool SEQ = false, typename LoadF, typename StoreF>
NWW_HD void dft25(LoadF load, StoreF store) {
    nww_c32 u[5][5];   // u[b][a] then u[b][c]
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int a = 0; a < 5; ++a) u[b][a] = load(5 * a + b);
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        if (SEQ) NWW_SCHED_FENCE();
        dft5(u[b][0], u[b][1], u[b][2], u[b][3], u[b][4]);
        if (b > 0) {
#pragma unroll
            for (int c = 1; c < 5; ++c) u[b][c] = c_mul(u[b][c], w25(b * c));
        }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        if (SEQ) NWW_SCHED_FENCE();
        dft5(u[0][c], u[1][c], u[2][c], u[3][c], u[4][c]);   // over b -> index d
#pragma unroll
        for (int d = 0; d < 5; ++d) store(c + 5 * d, u[d][c]);
    }
    if (SEQ) NWW_SCHED_FENCE();
}

// S1: task (frame f, column n2).  span = int16 samples of this chunk (frame f starts at hop*f),
// yz = scratch [FC][8][25] complex.
NWW_HD void fe_s1(int f, int n2, int hop, const int16_t* span, const FeTables* tb, nww_c32* yz) {
    // hop is even and span is 4-byte aligned: one 32-bit LDS read fetches the (even, odd) sample pair
    const uint32_t* fr = (const uint32_t*)(span + hop * f);
    nww_c32 z[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int m = 25 * n1 + n2;
        const nww_c32 w = tb->win2[m];
        const uint32_t v = fr[m];
        z[n1] = c_make((float)(int16_t)(v & 0xffffu) * w.x, (float)(int16_t)(v >> 16) * w.y);
    }
    dft8(z);
    nww_c32* out = yz + (f * 8) * 25 + n2;
    out[0] = z[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) out[k1 * 25] = c_mul(z[k1], tb->tw200[k1 * 25 + n2]);
}

// S2: task (frame f, row k1): 25-point DFT along n2, in place (index n2 -> k2).
NWW_HD void fe_s2(int f, int k1, nww_c32* yz) {
    nww_c32* row = yz + (f * 8 + k1) * 25;
    dft25([&](int i) { return row[i]; }, [&](int i, nww_c32 v) { row[i] = v; });
}

// S3 arithmetic: Z[k] = A, Z[200-k] = B of the 200-point complex FFT -> power of bins k and 200-k of the 400-point
// real FFT (tw = exp(-2 pi i k / 400)).  A and B arrive times 1/2 (FeTables::win2), which is the split's own factor.
NWW_HD void fe_s3_core(nww_c32 A, nww_c32 B, nww_c32 tw, float* pa, float* pb) {
    const nww_c32 E = c_make(A.x + B.x, A.y - B.y);
    const nww_c32 O = c_make(A.y + B.y, B.x - A.x);
    const nww_c32 wO = c_mul(tw, O);
    const nww_c32 Xa = c_add(E, wO), Xb = c_sub(E, wO);
    *pa = Xa.x * Xa.x + Xa.y * Xa.y;
    *pb = Xb.x * Xb.x + Xb.y * Xb.y;
}
// position of Z[k] in the S2 output order
NWW_HD int fe_zpos(int k) { return (k & 7) * 25 + (k >> 3); }

// S3: task (frame f, bin k in 0..100): power of bins k and 200-k into pw[f*FE_PSTRIDE + ...].
NWW_HD void fe_s3(int f, int k, const FeTables* tb, const nww_c32* yz, float* pw) {
    const nww_c32* zf = yz + f * 200;
    const int kb = (k == 0) ? 0 : 200 - k;
    float* p = pw + f * FE_PSTRIDE;
    fe_s3_core(zf[fe_zpos(k)], zf[fe_zpos(kb)], tb->tw400[k], &p[k], &p[200 - k]);
}

// S4: task (frame f, mel j): sparse triangular contraction; returns mel power.
NWW_HD float fe_s4(int f, int j, const FeTables* tb, const float* pw) {
    const float* p = pw + f * FE_PSTRIDE + tb->mel_lo[j];
    const float* w = tb->melw + tb->mel_off[j];
    const int n = tb->mel_cnt[j];
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc = fmaf(p[i], w[i], acc);
    return acc;
}

NWW_HD float fe_db(float mel, float amin, float mult) {
    return mult * log10f(fmaxf(mel, amin));
}

// Source index of padded position s (s relative to sample 0, may be <0 or >=N): torch 'reflect'.
NWW_HD int fe_reflect(int s, int N) {
    if (s < 0) s = -s;
    if (s >= N) s = 2 * (N - 1) - s;
    return s;
}

// =====================================================================================================================
// v2 "wave-private" schedule (frontend2.hip): one wave owns FE2_G consecutive frames of a clip and runs S1..S4 on
// them without any workgroup barrier.  The arithmetic is the same 8 x 25 factorisation; what changes is where the
// data lives: the lane's window/twiddle factors stay in registers for the whole launch, samples come straight from
// global memory (L1/L2), and one 400-dword LDS region per frame is reused in place by every stage:
//   S1 writes Y[k1][n2] (complex, [k1*25+n2]);  S2 reads its row and writes Z in natural order (Z[k] at complex k);
//   S3 reads Z[k], Z[200-k] and writes the 201 powers as floats at dword FE2_PSHIFT(f) + k of the same region;
//   S4 contracts them with the mel filterbank and stages the dB values at dword FE2_STAGE_OFF + FE2_PSHIFT(f) + j.
#define FE2_G 8                       // frames per wave item
#define FE2_FRAME_DW 400              // dwords of LDS per frame
#define FE2_PSHIFT(f) (4 * ((f) >> 1))   // per-frame shift of the power row: spreads the 8 frames over distinct LDS banks
#define FE2_STAGE_OFF 216
#define FE2_MAX_TILES 8               // 16-filter tiles (n_mels <= 128)
#define FE2_MAX_STEPS 512             // >= total MFMA steps of the mel plan (dense worst case 8 * 56)

// Mel contraction plan for v_mfma_f32_16x16x4_f32: D[frame][filter] += P[frame][k] * fb[k][filter], one 16-filter tile at
// a time over the tile's own bin range [k0, k0 + 4 nsteps).  Steps come in chunks of FE2_CHUNK (operands of the next
// chunk are fetched while the current one is on the matrix pipe), so every tile is padded to whole chunks with
// all-zero B rows.  b holds the B operand of every step in lane order; chunk_meta[c] = first bin of the chunk |
// tile << 12 | (last chunk of its tile) << 16.
#define FE2_CHUNK 8
#define FE2_MAX_CHUNKS 64
struct Fe2MelPlan {
    int32_t ntiles;
    int32_t total_steps;
    int32_t nchunks;
    int32_t tile_k0[FE2_MAX_TILES];
    int32_t tile_nsteps[FE2_MAX_TILES];   // multiple of FE2_CHUNK (two accumulator chains: even / odd steps)
    int32_t tile_first[FE2_MAX_TILES];
    uint32_t chunk_meta[FE2_MAX_CHUNKS];
    float b[FE2_MAX_STEPS * 64];          // [step][lane]: fb[k0 + 4 s + lane/16][16 tile + lane%16] (0 outside the table)
};

// S1 body: 8 (even, odd) int16 sample pairs of column n2 -> windowed radix-8 DFT -> twiddled Y[k1][n2], k1 = 0..7
NWW_HD void fe2_s1(const uint32_t s[8], const nww_c32 win[8], const nww_c32 tw[7], nww_c32 z[8]) {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
        z[n1] = c_make((float)(int16_t)(s[n1] & 0xffffu) * win[n1].x, (float)(int16_t)(s[n1] >> 16) * win[n1].y);
    dft8(z);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) z[k1] = c_mul(z[k1], tw[k1 - 1]);
}

// mel power of (frame, filter j) in the summation order of the MFMA plan: two fmaf chains (even / odd steps, 4 bins per
// step in ascending order), added at the end.  p = the frame's power row indexed by FFT bin (readable up to k0+4*nsteps).
NWW_HD float fe2_mel_planned(const Fe2MelPlan* pl, const float* p, int j) {
    const int t = j >> 4, n = j & 15;
    const int k0 = pl->tile_k0[t], ns = pl->tile_nsteps[t];
    const float* b = pl->b + (size_t)pl->tile_first[t] * 64 + n;
    float acc[2] = {0.0f, 0.0f};
    for (int s = 0; s < ns; ++s)
        for (int kk = 0; kk < 4; ++kk) acc[s & 1] = fmaf(p[k0 + 4 * s + kk], b[s * 64 + kk * 16], acc[s & 1]);
    return acc[0] + acc[1];
}
