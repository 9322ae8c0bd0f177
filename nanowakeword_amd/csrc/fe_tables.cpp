#include "fe_tables.h"
#include <cmath>
#include <cstring>

static const double kPi = 3.14159265358979323846;

void fe_default_window(int win_length, std::vector<float>& window) {
    window.resize(win_length);
    for (int n = 0; n < win_length; ++n)
        window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * kPi * n / win_length));   // torch.hann_window(periodic=True)
}

void fe_default_melfb(const FeParams& p, std::vector<float>& fb) {
    // torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk")
    const int n_freqs = p.n_fft / 2 + 1;
    fb.assign((size_t)n_freqs * p.n_mels, 0.f);
    const double m_min = 2595.0 * std::log10(1.0 + (double)p.f_min / 700.0);
    const double m_max = 2595.0 * std::log10(1.0 + (double)p.f_max / 700.0);
    std::vector<double> f_pts(p.n_mels + 2);
    for (int i = 0; i < p.n_mels + 2; ++i) {
        const double m = m_min + (m_max - m_min) * i / (p.n_mels + 1);
        f_pts[i] = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    }
    const double nyq = (double)(p.sample_rate / 2);
    for (int k = 0; k < n_freqs; ++k) {
        const double f = nyq * k / (n_freqs - 1);
        for (int j = 0; j < p.n_mels; ++j) {
            const double down = (f - f_pts[j]) / (f_pts[j + 1] - f_pts[j]);          // -slopes[:, :-2]/f_diff[:-1]
            const double up = (f_pts[j + 2] - f) / (f_pts[j + 2] - f_pts[j + 1]);    //  slopes[:, 2:]/f_diff[1:]
            const double v = std::fmax(0.0, std::fmin(down, up));
            fb[(size_t)k * p.n_mels + j] = (float)v;
        }
    }
}

std::string fe_build_tables(const FeParams& p, const float* window, const float* fb, FeTables* t) {
    if (p.n_fft != FE_NFFT) return "only n_fft=400 is implemented";
    if (p.win_length > p.n_fft || p.win_length <= 0) return "win_length must be in 1..n_fft";
    if (p.n_mels <= 0 || p.n_mels > FE_MAX_MELS) return "n_mels must be in 1..128";
    if (p.hop <= 0 || (p.hop & 1)) return "hop_length must be positive and even";
    std::memset(t, 0, sizeof(*t));
    // centre-padded window (onnx.py:51-53), int16 normalisation 1/32768 folded in (exact scaling) - and the 1/2 of the real-FFT split
    // (fe_s3_core: E = (A + conj B) / 2, O = (A - conj B) / 2i): the whole transform carries half the value, a power of two, so every
    // rounding is the one the unscaled arithmetic makes and the powers come out bit for bit the same with two multiplies less per bin pair
    std::vector<float> wp(p.n_fft, 0.f);
    const int pad_left = (p.n_fft - p.win_length) / 2;
    for (int i = 0; i < p.win_length; ++i) wp[pad_left + i] = window[i];
    for (int m = 0; m < FE_M; ++m) {
        t->win2[m].x = wp[2 * m] * (1.0f / 65536.0f);
        t->win2[m].y = wp[2 * m + 1] * (1.0f / 65536.0f);
    }
    for (int n2 = 0; n2 < 25; ++n2)
        for (int k1 = 0; k1 < 8; ++k1) {
            const double a = -2.0 * kPi * (double)(n2 * k1) / 200.0;
            t->tw200[k1 * 25 + n2].x = (float)std::cos(a);
            t->tw200[k1 * 25 + n2].y = (float)std::sin(a);
        }
    for (int k = 0; k <= 100; ++k) {
        const double a = -2.0 * kPi * (double)k / 400.0;
        t->tw400[k].x = (float)std::cos(a);
        t->tw400[k].y = (float)std::sin(a);
    }
    int off = 0;
    for (int j = 0; j < p.n_mels; ++j) {
        int lo = -1, hi = -1;
        for (int k = 0; k < FE_BINS; ++k)
            if (fb[(size_t)k * p.n_mels + j] != 0.f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { lo = 0; hi = -1; }     // empty filter (possible for tiny n_mels/f ranges): contributes 0
        const int cnt = hi - lo + 1;
        if (off + cnt > FE_MAX_MELW) return "mel filterbank has too many non-zeros for the LDS table";
        t->mel_lo[j] = lo; t->mel_cnt[j] = cnt; t->mel_off[j] = off;
        for (int i = 0; i < cnt; ++i) t->melw[off + i] = fb[(size_t)(lo + i) * p.n_mels + j];
        off += cnt;
    }
    return "";
}

int fe_num_frames(const FeParams& p, int n) {
    if (p.center) {
        if (n <= p.n_fft / 2) return -1;            // reflect padding needs pad < N
        return 1 + n / p.hop;                       // (n + 2*(n_fft/2) - n_fft)/hop + 1
    }
    if (n < p.n_fft) return -1;
    return 1 + (n - p.n_fft) / p.hop;
}

std::string fe2_build_mel_plan(const FeParams& p, const float* fb, Fe2MelPlan* pl) {
    std::memset(pl, 0, sizeof(*pl));
    const int ntiles = (p.n_mels + 15) / 16;
    if (ntiles > FE2_MAX_TILES) return "n_mels must be <= 128";
    pl->ntiles = ntiles;
    int first = 0;
    for (int t = 0; t < ntiles; ++t) {
        int lo = FE_BINS, hi = -1;
        for (int k = 0; k < FE_BINS; ++k)
            for (int n = 0; n < 16; ++n) {
                const int j = 16 * t + n;
                if (j < p.n_mels && fb[(size_t)k * p.n_mels + j] != 0.f) { if (k < lo) lo = k; if (k > hi) hi = k; }
            }
        if (hi < 0) { lo = 0; hi = 0; }                      // a tile of empty filters: one all-zero pair of steps
        int ns = (hi - lo + 1 + 3) / 4;
        ns = (ns + FE2_CHUNK - 1) / FE2_CHUNK * FE2_CHUNK;    // whole chunks (even: two accumulator chains)
        if (first + ns > FE2_MAX_STEPS || (first + ns) / FE2_CHUNK > FE2_MAX_CHUNKS) return "mel filterbank is too dense for the MFMA plan";
        for (int c = 0; c < ns / FE2_CHUNK; ++c)
            pl->chunk_meta[first / FE2_CHUNK + c] =
                (uint32_t)(lo + 4 * FE2_CHUNK * c) | ((uint32_t)t << 12) | ((c == ns / FE2_CHUNK - 1) ? 1u << 16 : 0u);
        pl->tile_k0[t] = lo; pl->tile_nsteps[t] = ns; pl->tile_first[t] = first;
        for (int s = 0; s < ns; ++s)
            for (int l = 0; l < 64; ++l) {
                const int k = lo + 4 * s + (l >> 4), j = 16 * t + (l & 15);
                pl->b[(size_t)(first + s) * 64 + l] = (k < FE_BINS && j < p.n_mels) ? fb[(size_t)k * p.n_mels + j] : 0.f;
            }
        first += ns;
    }
    pl->total_steps = first;
    pl->nchunks = first / FE2_CHUNK;
    return "";
}
