// fe3.h - the matrix-pipe frontend (frontend3.hip): index maps, plan tables and binary16 helpers shared by the gfx950 kernel,
// the host-side plan builder (fe_tables.cpp) and the CPU emulator of the non-GPU tests (tests/hostemu/fe_emu.cpp).
//
// Replaces, per 400-sample frame, the dense windowed DFT the reference's CPU interpreter executes as two conv1d's
// (nanowakeword/_export/onnx.py:42-63 bases, :66-83 forward) by a PRIME-FACTOR 25 x 16 pair of small dense products on
// v_mfma_f32_16x16x32_f16, in the two-term binary16 arithmetic of DESIGN 2:
//   sample n = 16 j + c of the frame  (c = n mod 16 "class", j = 0..24);  bin k <-> (k1, k2) = (k mod 16, k mod 25)
//   n1 = 9 n mod 16, n2 = 11 n mod 25  =>  exp(-2 pi i n k / 400) = W16^(n1 k1) W25^(n2 k2)      (no twiddles between the stages)
//   stage 1, per class c:  Z_c[k2] = sum_j x[16 j + c] w[16 j + c] W25^(n2 k2),  k2 = 0..12 (real input: the rest are conjugates)
//            M = 26 real rows (re / im of k2), K = 25 samples, N = 16 frames; the WINDOW is folded into the 16 matrices (plan time)
//   stage 2, per k2:       X[k1, k2] = sum_c Z_c[k2] W16^(n1(c) k1)
//            M = 32 real rows (re / im of k1 = 0..15), K = 32 (re / im of the 16 classes), N = 16 frames; ONE matrix for all k2
//   bin of (k1, k2): k = (225 k1 + 176 k2) mod 400, and 400 - k when that exceeds 200 (conjugate: same power)
// Operands: an int16 sample IS two binary16 terms exactly (hi = RN16(x), lo = x - hi, |lo| <= 8); matrix entries times a power of
// two are hi + lo at plan time; the stage-1 accumulators leave as hi = RN16(acc 2^-12), lo' = RN16(acc - hi 2^12) (lo' keeps its
// own 11 bits at every magnitude: it meets the matrix times 2^-12).  Three partial products per stage, each exact, float32 sums.
// hop_length must be 160 = 16 x 10: then the class of a sample does not depend on the frame, the 25 samples of (frame t, class c)
// are the CONTIGUOUS entries 10 t .. 10 t + 24 of the clip's class-c plane, and one staging pass serves all frames.
#pragma once
#include <stdint.h>
#include <string.h>
#include "fe_steps.h"

#define FE3_F 16                 // frames per work item (the N of the matrix instruction)
#define FE3_HOP 160
#define FE3_LPF (FE3_HOP / 16)   // plane entries a frame advances
#define FE3_PL 184               // plane entries staged per item: 10 * 15 + 32 (the K = 32 fragment of the last frame), rounded to 8
#define FE3_NK2 13               // k2 = 0..12
#define FE3_PP 212               // float pitch of a power row in LDS (multiple of 4: 16-byte reads of the mel stage)
#define FE3_XP_BYTES (2 * 16 * FE3_PL * 2)
#define FE3_ZT_BYTES (FE3_NK2 * FE3_F * 64)          // one term of Z: [k2][frame] rows of 32 binary16 = (class, re / im)
#define FE3_P_BYTES ((FE3_F * FE3_PP + 32) * 4)      // + a zeroed tail the last row's padded taps read
#define FE3_LDS_BYTES (FE3_XP_BYTES + 2 * FE3_ZT_BYTES + FE3_P_BYTES)

NWW_HD int fe3_n1_of(int n) { return (9 * n) & 15; }
NWW_HD int fe3_n2_of(int n) { return (11 * n) % 25; }
// power-row index of (k1, k2), k2 = 0..12; -1 for the k2 = 0 rows that duplicate a conjugate
NWW_HD int fe3_bin_of(int k1, int k2) {
    const int k = (225 * k1 + 176 * k2) % 400;
    if (k <= 200) return k;
    return k2 == 0 ? -1 : 400 - k;
}
// Z row (k2, frame f) holds four 16-byte chunks (chunk g = classes 4 g .. 4 g + 3, re / im interleaved); chunk g sits at position
// a ^ (3 g & 3), a = (f >> 2) & 3: with that map the 16 lanes of every ds_read_b128 group (whose membership mixes g) and of every
// ds_write_b128 group fall on 16 distinct bank slots
NWW_HD int fe3_z_off(int k2, int f, int g) { return (k2 * FE3_F + f) * 64 + 16 * ((((f >> 2) & 3) ^ (3 * g)) & 3); }

// ---- binary16 on the host (gcc 11 has no _Float16 on x86): round to nearest even, subnormals kept
NWW_HD uint16_t fe3_f32_to_f16(float v) {
    uint32_t x;
    memcpy(&x, &v, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));       // overflow -> inf, nan
    if (x < 0x38800000u) {                      // below 2^-14: subnormal or zero
        if (x < 0x33000000u) return (uint16_t)sign;                                                // < 2^-25: rounds to zero
        const int e = (int)(x >> 23);           // biased exponent, 102..112
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;              // 14..24: m / 2^shift in units of 2^-24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t rem = x & 0x1fffu;
    uint32_t h = (x - 0x38000000u) >> 13;       // exponent rebias, 10-bit mantissa
    h += (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ? 1u : 0u;
    return (uint16_t)(sign | h);
}
NWW_HD float fe3_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) { x = sign; }
        else {
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; ++s; }
            x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float v;
    memcpy(&v, &x, 4);
    return v;
}

// Plan tables (device memory; read once per workgroup into registers, the bin map per stage-2 tile from L1 / L2).
// Fragment element e of lane l: row = l & 15 of the 16-row tile, k slot = 8 (l >> 4) + e (v_mfma_f32_16x16x32_f16's A operand).
struct Fe3Plan {
    // stage 1 [class c][row tile][term: hi, lo][lane][e]: row rho = 16 tile + (l & 15) = 2 k2 + part (part 0 = re, 1 = im), k slot =
    // j: the matrix entry w[16 j + c] cos / sin(-2 pi n2(16 j + c) k2 / 25) times m_scale; rows >= 26 and slots >= 25 are zero
    uint16_t a1[16][2][2][64][8];
    // stage 2 [row tile][variant: hi, lo, hi 2^-12][lane][e]: row sigma = 2 k1 + part, k slot = 2 c + part_in:
    // (re, re) = cos t, (re, im) = -sin t, (im, re) = sin t, (im, im) = cos t, t = -2 pi n1(c) k1 / 16, times 2^14
    uint16_t a2[2][3][64][8];
    // [k2][lane][q]: power-row index of the lane's four bins of tile k2 - row tile q >> 1, k1 = 8 (q >> 1) + 2 (l >> 4) + (q & 1); -1: skip
    int16_t bin[FE3_NK2][64][4];
    float m_scale;       // power of two the stage-1 matrices carry (2^8 for the Hann window)
    float p_scale;       // power = (re^2 + im^2) p_scale;  p_scale = (m_scale 2^15 2^-12 2^14)^-2
};
#define FE3_Z_DOWN 0.000244140625f      // 2^-12
#define FE3_Z_UP 4096.0f
#define FE3_D_SCALE 16384.0f            // 2^14
