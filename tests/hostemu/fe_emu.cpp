// CPU emulation of the frontend kernel's task schedule (tests only; NOT a product fallback).
// Runs the exact per-lane task bodies of nanowakeword_amd/csrc/fe_steps.h sequentially, chunk by chunk,
// with plain arrays standing in for LDS.  Built by tests/test_hostemu.py with g++.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fe_steps.h"
#include "fe_tables.h"

extern "C" int emu_frontend(const int16_t* pcm, int B, int N, int n_mels, int center, int hop,
                            const float* window, const float* fb,   /* may be NULL -> defaults */
                            int fc, float* mel_out, float* db_out /* [B][n_mels][T] */) {
    FeParams p; p.n_mels = n_mels; p.center = center; p.hop = hop;
    std::vector<float> w, f;
    if (!window) { fe_default_window(p.win_length, w); window = w.data(); }
    if (!fb) { fe_default_melfb(p, f); fb = f.data(); }
    FeTables tb;
    if (!fe_build_tables(p, window, fb, &tb).empty()) return -2;
    const int T = fe_num_frames(p, N);
    if (T < 0) return -1;
    const int pad = center ? FE_NFFT / 2 : 0;
    std::vector<int16_t> span((size_t)hop * (fc - 1) + FE_NFFT);
    std::vector<nww_c32> yz((size_t)fc * 200);
    std::vector<float> pw((size_t)fc * FE_PSTRIDE);
    for (int b = 0; b < B; ++b) {
        const int16_t* x = pcm + (size_t)b * N;
        for (int t0 = 0; t0 < T; t0 += fc) {
            const int nf = (T - t0 < fc) ? T - t0 : fc;
            const int len = hop * (nf - 1) + FE_NFFT;
            for (int i = 0; i < len; ++i) span[i] = x[fe_reflect(hop * t0 - pad + i, N)];
            for (int task = 0; task < nf * 25; ++task) fe_s1(task / 25, task % 25, hop, span.data(), &tb, yz.data());
            for (int task = 0; task < nf * 8; ++task) fe_s2(task / 8, task % 8, yz.data());
            for (int task = 0; task < nf * 101; ++task) fe_s3(task / 101, task % 101, &tb, yz.data(), pw.data());
            for (int task = 0; task < nf * n_mels; ++task) {
                const int fi = task / n_mels, j = task % n_mels;
                const float m = fe_s4(fi, j, &tb, pw.data());
                const size_t o = ((size_t)b * n_mels + j) * T + t0 + fi;
                if (mel_out) mel_out[o] = m;
                if (db_out) db_out[o] = fe_db(m, p.amin, p.db_mult);
            }
        }
    }
    return T;
}

// ---- v2: the wave-private schedule of frontend2.hip (items of FE2_G frames, one 400-dword region per frame reused in
// place by every stage, samples read per column, powers shifted by FE2_PSHIFT, mel in the MFMA plan's summation
// order or the sparse loop).  Lanes are run one after another; within a stage all reads of a wave instruction
// happen before its writes on the GPU, which the emulator reproduces by buffering where a stage works in place.
extern "C" int emu_frontend2(const int16_t* pcm, int B, int N, int n_mels, int center, int hop,
                             const float* window, const float* fb, int mfma_mel,
                             float* mel_out, float* db_out /* [B][n_mels][T] */) {
    FeParams p; p.n_mels = n_mels; p.center = center; p.hop = hop;
    std::vector<float> w, f;
    if (!window) { fe_default_window(p.win_length, w); window = w.data(); }
    if (!fb) { fe_default_melfb(p, f); fb = f.data(); }
    FeTables tb;
    if (!fe_build_tables(p, window, fb, &tb).empty()) return -2;
    std::vector<Fe2MelPlan> plan(1);
    if (!fe2_build_mel_plan(p, fb, plan.data()).empty()) return -3;
    const int T = fe_num_frames(p, N);
    if (T < 0) return -1;
    const int pad = center ? FE_NFFT / 2 : 0;
    std::vector<float> slab((size_t)FE2_G * FE2_FRAME_DW);
    for (int b = 0; b < B; ++b) {
        const int16_t* x = pcm + (size_t)b * N;
        for (int t0 = 0; t0 < T; t0 += FE2_G) {
            const int nf = (T - t0 < FE2_G) ? T - t0 : FE2_G;
            // S1: lane = (slot, n2), iteration it -> frame 2 it + slot
            for (int fi = 0; fi < nf; ++fi)
                for (int n2 = 0; n2 < 25; ++n2) {
                    uint32_t s[8];
                    nww_c32 win[8], tw[7], z[8];
                    const int base = (t0 + fi) * hop - pad;
                    for (int n1 = 0; n1 < 8; ++n1) {
                        const int q = base + 50 * n1 + 2 * n2;
                        s[n1] = (uint32_t)(uint16_t)x[fe_reflect(q, N)] | ((uint32_t)(uint16_t)x[fe_reflect(q + 1, N)] << 16);
                        win[n1] = tb.win2[25 * n1 + n2];
                    }
                    for (int k1 = 1; k1 < 8; ++k1) tw[k1 - 1] = tb.tw200[k1 * 25 + n2];
                    fe2_s1(s, win, tw, z);
                    nww_c32* y = reinterpret_cast<nww_c32*>(slab.data() + fi * FE2_FRAME_DW) + n2;
                    for (int k1 = 0; k1 < 8; ++k1) y[k1 * 25] = z[k1];
                }
            // S2: lane = (frame, k1); all 64 lanes read their rows before any lane stores (one wave instruction stream)
            for (int fi = 0; fi < nf; ++fi) {
                nww_c32* zf = reinterpret_cast<nww_c32*>(slab.data() + fi * FE2_FRAME_DW);
                nww_c32 out[200];
                for (int k1 = 0; k1 < 8; ++k1) {
                    const nww_c32* row = zf + k1 * 25;
                    dft25<true>([&](int i) { return row[i]; }, [&](int i, nww_c32 v) { out[k1 + 8 * i] = v; });
                }
                for (int k = 0; k < 200; ++k) zf[k] = out[k];
            }
            // S3: lane = bin k (and 64 + k); the frame's reads precede its writes
            for (int fi = 0; fi < nf; ++fi) {
                const nww_c32* zf = reinterpret_cast<const nww_c32*>(slab.data() + fi * FE2_FRAME_DW);
                float pw[201];
                for (int k = 0; k <= 100; ++k) {
                    float pa, pb;
                    fe_s3_core(zf[k], zf[k ? FE_M - k : 0], tb.tw400[k], &pa, &pb);
                    pw[k] = pa;
                    pw[FE_M - k] = pb;        // k = 100: the second store wins, as on the GPU
                }
                float* prow = slab.data() + fi * FE2_FRAME_DW + FE2_PSHIFT(fi);
                for (int k = 0; k <= 200; ++k) prow[k] = pw[k];
            }
            // S4
            for (int fi = 0; fi < nf; ++fi) {
                const float* prow = slab.data() + fi * FE2_FRAME_DW + FE2_PSHIFT(fi);
                for (int j = 0; j < n_mels; ++j) {
                    float m;
                    if (mfma_mel) {
                        m = fe2_mel_planned(plan.data(), prow, j);
                    } else {
                        const float* pp = prow + tb.mel_lo[j];
                        const float* ww = tb.melw + tb.mel_off[j];
                        m = 0.0f;
                        for (int i = 0; i < tb.mel_cnt[j]; ++i) m = fmaf(pp[i], ww[i], m);
                    }
                    const size_t o = ((size_t)b * n_mels + j) * T + t0 + fi;
                    if (mel_out) mel_out[o] = m;
                    if (db_out) db_out[o] = fe_db(m, p.amin, p.db_mult);
                }
            }
        }
    }
    return T;
}
