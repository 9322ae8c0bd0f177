// fe3_tables.cpp - plan of the matrix-pipe frontend experiment (fe3.h; moved out of the product library in round 6)
#include "fe_tables.h"
#include <cmath>
#include <cstring>
#include "fe3.h"
#include "fe3_tables.h"

static const double kPi = 3.14159265358979323846;

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
