// fe3_tables.h - plan builder (host) + launcher (hipcc only) of the matrix-pipe frontend experiment (frontend3.hip / fe3.h)
#pragma once
#include <string>
#include "fe_tables.h"
#include "fe3.h"

// Plan of the matrix-pipe frontend: the window-folded stage-1 matrices, the 16-point matrix and the bin map as the register images the
// kernel's MFMAs consume.  "" or why this configuration has no such plan.
std::string fe3_build_plan(const FeParams& p, const float* window, Fe3Plan* out);

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include "frontend.h"
// the same contract as fe2_launch for hop_length = 160, n_mels <= 64, filters of <= 25 taps
bool fe3_supported(const FeParams& p, int max_taps);
hipError_t fe3_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p, const FeTables* d_tables,
                      const Fe3Plan* d_plan, float* d_db, float* d_mel, int frames_major, int max_taps, int max_grid,
                      hipStream_t stream, const Fe2Sub* subset = nullptr);
#endif
