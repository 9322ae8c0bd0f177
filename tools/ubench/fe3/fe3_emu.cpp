// fe3_emu.cpp - lane-by-lane CPU emulation of the matrix-pipe frontend experiment (moved out of tests/hostemu in round 6)
// build: g++ -O2 -std=c++17 -shared -fPIC -I tools/ubench/fe3 -I nanowakeword_amd/csrc -o tools/ubench/fe3/libfe3_emu.so tools/ubench/fe3/fe3_emu.cpp tools/ubench/fe3/fe3_tables.cpp nanowakeword_amd/csrc/fe_tables.cpp
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fe_steps.h"
#include "fe_tables.h"
#include "fe3.h"
#include "fe3_tables.h"

// ---- v3: the matrix-pipe frontend (frontend3.hip / fe3.h), lane by lane: the plan's register images, the LDS planes / Z rows / power
// rows at the kernel's own addresses, v_mfma_f32_16x16x32_f16 as a k-ordered float32 sum of exact binary16 products (the hardware's
// internal order is not specified: results agree to float32 rounding, not bit for bit), the fma_mix splits as single roundings.
#ifndef FE3_EMU_KBLOCK
#define FE3_EMU_KBLOCK 8
#endif
namespace {
struct Frag { uint16_t e[64][8]; };
struct Acc { float r[64][4]; };
// D = A B + C: A[i][k] = lane i + 16 (k / 8) element k % 8, B[k][j] = lane j + 16 (k / 8) element k % 8, C[i][j] = lane j + 16 (i / 4) reg i % 4
void mfma16(const uint16_t a[64][8], const Frag& b, Acc& c) {
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            // products are exact; the adder tree is modelled as exact sums over FE3_EMU_KBLOCK consecutive k, each block rounded into
            // the float32 accumulator (the hardware's tree is not documented: the GPU tests, not this model, bound its rounding)
            float s = c.r[j + 16 * (i >> 2)][i & 3];
            for (int k0 = 0; k0 < 32; k0 += FE3_EMU_KBLOCK) {
                double blk = 0.0;
                for (int k = k0; k < k0 + FE3_EMU_KBLOCK; ++k)
                    blk += (double)fe3_f16_to_f32(a[i + 16 * (k >> 3)][k & 7]) * (double)fe3_f16_to_f32(b.e[j + 16 * (k >> 3)][k & 7]);
                s = (float)((double)s + blk);
            }
            c.r[j + 16 * (i >> 2)][i & 3] = s;
        }
}
}  // namespace

extern "C" int emu_frontend3(const int16_t* pcm, int B, int N, int n_mels, int center, int hop,
                             const float* window, const float* fb, float* mel_out, float* db_out /* [B][n_mels][T] */) {
    FeParams p; p.n_mels = n_mels; p.center = center; p.hop = hop;
    std::vector<float> w, f;
    if (!window) { fe_default_window(p.win_length, w); window = w.data(); }
    if (!fb) { fe_default_melfb(p, f); fb = f.data(); }
    FeTables tb;
    if (!fe_build_tables(p, window, fb, &tb).empty()) return -2;
    std::vector<Fe3Plan> plv(1);
    if (!fe3_build_plan(p, window, plv.data()).empty()) return -3;
    const Fe3Plan& pl = plv[0];
    const int T = fe_num_frames(p, N);
    if (T < 0) return -1;
    const int pad = center ? FE_NFFT / 2 : 0;
    std::vector<uint16_t> xp((size_t)2 * 16 * FE3_PL);
    std::vector<uint8_t> z((size_t)2 * FE3_ZT_BYTES);
    std::vector<float> P((size_t)FE3_F * FE3_PP + 32, 0.0f);
    for (int b = 0; b < B; ++b) {
        const int16_t* x = pcm + (size_t)b * N;
        for (int t0 = 0; t0 < T; t0 += FE3_F) {
            const int nf = (T - t0 < FE3_F) ? T - t0 : FE3_F;
            // staging: padded position p0 + 16 li + c -> plane c, entry li, as hi = RN16(x), lo = x - hi (exact)
            const int p0 = t0 * hop;
            for (int c = 0; c < 16; ++c)
                for (int li = 0; li < FE3_PL; ++li) {
                    int s = fe_reflect(p0 + 16 * li + c - pad, N);
                    s = s < 0 ? 0 : (s >= N ? N - 1 : s);
                    const float xf = (float)x[s];
                    const uint16_t hi = fe3_f32_to_f16(xf);
                    xp[(0 * 16 + c) * FE3_PL + li] = hi;
                    xp[(1 * 16 + c) * FE3_PL + li] = fe3_f32_to_f16(xf - fe3_f16_to_f32(hi));
                }
            // stage 1: wave wv owns classes 4 wv .. 4 wv + 3
            for (int wv = 0; wv < 4; ++wv)
                for (int ci = 0; ci < 4; ++ci) {
                    const int c = 4 * wv + ci;
                    Frag bh, bl;
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int li = FE3_LPF * (l & 15) + 8 * (l >> 4) + e;
                            bh.e[l][e] = xp[(0 * 16 + c) * FE3_PL + li];
                            bl.e[l][e] = xp[(1 * 16 + c) * FE3_PL + li];
                        }
                    for (int mt = 0; mt < 2; ++mt) {
                        Acc acc;
                        memset(&acc, 0, sizeof(acc));
                        mfma16(pl.a1[c][mt][0], bh, acc);
                        mfma16(pl.a1[c][mt][0], bl, acc);
                        mfma16(pl.a1[c][mt][1], bh, acc);
                        for (int l = 0; l < 64; ++l)
                            for (int r = 0; r < 4; ++r) {
                                const int fi = l & 15, g = l >> 4, k2 = 8 * mt + 2 * g + (r >> 1), part = r & 1;
                                if (k2 >= FE3_NK2) continue;
                                const float a = acc.r[l][r];
                                const uint16_t hi = fe3_f32_to_f16(a * FE3_Z_DOWN);
                                const uint16_t lo = fe3_f32_to_f16(a - fe3_f16_to_f32(hi) * FE3_Z_UP);
                                const int off = fe3_z_off(k2, fi, wv) + (2 * ci + part) * 2;
                                memcpy(&z[off], &hi, 2);
                                memcpy(&z[FE3_ZT_BYTES + off], &lo, 2);
                            }
                    }
                }
            // stage 2: one tile per k2
            for (int k2 = 0; k2 < FE3_NK2; ++k2) {
                Frag bh, bl;
                for (int l = 0; l < 64; ++l) {
                    const int off = fe3_z_off(k2, l & 15, l >> 4);
                    memcpy(bh.e[l], &z[off], 16);
                    memcpy(bl.e[l], &z[FE3_ZT_BYTES + off], 16);
                }
                for (int mt = 0; mt < 2; ++mt) {
                    Acc acc;
                    memset(&acc, 0, sizeof(acc));
                    mfma16(pl.a2[mt][0], bh, acc);
                    mfma16(pl.a2[mt][1], bh, acc);
                    mfma16(pl.a2[mt][2], bl, acc);
                    for (int l = 0; l < 64; ++l)
                        for (int half = 0; half < 2; ++half) {
                            const int idx = pl.bin[k2][l][2 * mt + half];
                            if (idx < 0) continue;
                            const float re = acc.r[l][2 * half], im = acc.r[l][2 * half + 1];
                            P[(size_t)(l & 15) * FE3_PP + idx] = fmaf(im, im, re * re) * pl.p_scale;
                        }
                }
            }
            // mel + dB (register filters: ascending taps, one fmaf chain)
            for (int fi = 0; fi < nf; ++fi) {
                const float* prow = P.data() + (size_t)fi * FE3_PP;
                for (int j = 0; j < n_mels; ++j) {
                    const float* pp = prow + tb.mel_lo[j];
                    const float* ww = tb.melw + tb.mel_off[j];
                    float m = 0.0f;
                    for (int i = 0; i < tb.mel_cnt[j]; ++i) m = fmaf(pp[i], ww[i], m);
                    const size_t o = ((size_t)b * n_mels + j) * T + t0 + fi;
                    if (mel_out) mel_out[o] = m;
                    if (db_out) db_out[o] = fe_db(m, p.amin, p.db_mult);
                }
            }
        }
    }
    return T;
}
