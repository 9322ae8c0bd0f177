#pragma once
#include <hip/hip_runtime.h>
#include "fe_tables.h"

// d_db / d_mel: [B][n_mels][T] (frames_major = 0) or [B][T][n_mels] (frames_major = 1); either may be null.
// row_stride: samples between consecutive clips (N for a dense [B][N] batch; 2*window for the streaming rings)
//
// Wave-private kernel (frontend2.hip).  mel_mode 2: lane = filter with register-resident
// weights (needs n_mels <= 64 and max_taps <= 28, else mode 1), 1: mel contraction on the matrix cores (d_plan from
// fe2_build_mel_plan), 0: sparse loop over LDS tables.  max_taps = longest filter support.  block = 256 (4 waves).
int fe2_lds_bytes(int waves, int mel_mode);
hipError_t fe2_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p,
                      const FeTables* d_tables, const Fe2MelPlan* d_plan, float* d_db, float* d_mel, int frames_major,
                      int mel_mode, int max_taps, int block, int max_grid, hipStream_t stream);
