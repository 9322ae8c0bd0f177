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

// ---- matrix-pipe frontend (fe3.h)
static void fe3_split(double v, uint16_t* hi, uint16_t* lo) {       // v (already scaled) = hi + lo in binary16, both round-to-nearest
    const uint16_t h = fe3_f32_to_f16((float)v);
    *hi = h;
    *lo = fe3_f32_to_f16((float)(v - (double)fe3_f16_to_f32(h)));
}

std::string fe3_build_plan(const FeParams& p, const float* window, Fe3Plan* pl) {
    if (p.n_fft != FE_NFFT || p.hop != FE3_HOP) return "the matrix-pipe frontend needs n_fft = 400 and hop_length = 160";
    if (p.win_length > p.n_fft || p.win_length <= 0) return "win_length must be in 1..n_fft";
    std::memset(pl, 0, sizeof(*pl));
    std::vector<double> wp(p.n_fft, 0.0);
    const int pad_left = (p.n_fft - p.win_length) / 2;
    for (int i = 0; i < p.win_length; ++i) wp[pad_left + i] = (double)window[i];
    // Scale of the stage-1 matrices: the accumulator of (class c, any row) is bounded by 32768 m_scale sum_j |w[16 j + c]| and must
    // stay below 2^27, so that hi = RN16(acc 2^-12) <= 2^15 and lo' = RN16(acc - hi 2^12) <= 2^15 are binary16 numbers
    double bound = 0.0;
    for (int c = 0; c < 16; ++c) {
        double s = 0.0;
        for (int j = 0; j < 25; ++j) s += std::fabs(wp[16 * j + c]);
        bound = std::fmax(bound, s);
    }
    if (!(bound > 0.0) || !std::isfinite(bound)) return "window is all zero or not finite";
    int e = (int)std::floor(std::log2(134217728.0 * 0.999 / (32768.0 * bound)));
    if (e > 14) e = 14;                                   // the largest entry stays below 2^15 in binary16 whatever the window
    double wmax = 0.0;
    for (int n = 0; n < p.n_fft; ++n) wmax = std::fmax(wmax, std::fabs(wp[n]));
    while (e > -20 && std::ldexp(wmax, e) >= 32768.0) --e;
    if (e <= -20) return "window too large for the binary16 plan";
    const double ms = std::ldexp(1.0, e);
    pl->m_scale = (float)ms;
    const double xs = ms * 32768.0 * (double)FE3_Z_DOWN * (double)FE3_D_SCALE;      // what a stage-2 accumulator carries of X
    pl->p_scale = (float)(1.0 / (xs * xs));
    for (int c = 0; c < 16; ++c)
        for (int mt = 0; mt < 2; ++mt)
            for (int l = 0; l < 64; ++l)
                for (int el = 0; el < 8; ++el) {
                    const int rho = 16 * mt + (l & 15), j = 8 * (l >> 4) + el, k2 = rho >> 1, part = rho & 1;
                    double v = 0.0;
                    if (rho < 2 * FE3_NK2 && j < 25) {
                        const int n = 16 * j + c;
                        const double a = -2.0 * kPi * (double)((fe3_n2_of(n) * k2) % 25) / 25.0;
                        v = wp[n] * (part ? std::sin(a) : std::cos(a)) * ms;
                    }
                    fe3_split(v, &pl->a1[c][mt][0][l][el], &pl->a1[c][mt][1][l][el]);
                }
    for (int mt = 0; mt < 2; ++mt)
        for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 8; ++el) {
                const int sigma = 16 * mt + (l & 15), slot = 8 * (l >> 4) + el, k1 = sigma >> 1, part = sigma & 1, c = slot >> 1, pin = slot & 1;
                const double a = -2.0 * kPi * (double)((fe3_n1_of(c) * k1) & 15) / 16.0;
                // (Zre + i Zim)(cos a + i sin a): re = Zre cos a - Zim sin a, im = Zre sin a + Zim cos a
                double v = part == 0 ? (pin == 0 ? std::cos(a) : -std::sin(a)) : (pin == 0 ? std::sin(a) : std::cos(a));
                if (std::fabs(v) < 1e-12) v = 0.0;        // exact zeros of the 16-point matrix (cos / sin of multiples of pi / 2)
                uint16_t hi, lo;
                fe3_split(v * (double)FE3_D_SCALE, &hi, &lo);
                pl->a2[mt][0][l][el] = hi;
                pl->a2[mt][1][l][el] = lo;
                pl->a2[mt][2][l][el] = fe3_f32_to_f16(fe3_f16_to_f32(hi) * FE3_Z_DOWN);      // exact: a power of two inside the normal range
            }
    for (int k2 = 0; k2 < FE3_NK2; ++k2)
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 4; ++q) pl->bin[k2][l][q] = (int16_t)fe3_bin_of(8 * (q >> 1) + 2 * (l >> 4) + (q & 1), k2);
    return "";
}
