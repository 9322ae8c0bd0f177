// fe_tables.h - host-side construction of the frontend tables (double precision, rounded once).
#pragma once
#include <string>
#include <vector>
#include "fe_steps.h"

struct FeParams {
    int sample_rate = 16000, n_fft = 400, win_length = 400, hop = 160, n_mels = 64, center = 1;
    float f_min = 0.f, f_max = 8000.f, amin = 1e-10f, db_mult = 10.f;
};

// torchaudio defaults (reference: nanowakeword/modules/architectures.py:830-836): Hann periodic window,
// HTK mel scale, norm=None triangular filterbank [n_fft/2+1][n_mels] row-major.
void fe_default_window(int win_length, std::vector<float>& window);
void fe_default_melfb(const FeParams& p, std::vector<float>& fb);

// Build LDS tables from a window [win_length] and filterbank [201][n_mels]. Returns "" or an error.
std::string fe_build_tables(const FeParams& p, const float* window, const float* fb, FeTables* out);

// MFMA mel-contraction plan of the wave-private kernel (frontend2.hip) from the same filterbank
std::string fe2_build_mel_plan(const FeParams& p, const float* fb, Fe2MelPlan* out);

// frame law; -1 if the clip is too short
int fe_num_frames(const FeParams& p, int n_samples);
