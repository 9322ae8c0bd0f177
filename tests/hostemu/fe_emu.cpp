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
