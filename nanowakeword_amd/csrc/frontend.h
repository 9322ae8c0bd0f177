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
// Streaming hop (nww_stream.hip): only the frames of up to four ranges [t0, t1) of every clip's window are computed (nr = 0: all
// T frames), and with ring_rows > 0 frame t is written to row row0 + t of a per-clip ring of 2 * ring_rows rows of n_mels floats
// (and again ring_rows rows away, so that any window of T rows starting below ring_rows is contiguous); out_clip_stride = floats
// between the rings of consecutive clips.  Frames-major log-mel output only.
struct Fe2Sub {
    int nr = 0;
    int t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
    int gend[4] = {0, 0, 0, 0};        // filled by fe2_launch: groups of ranges 0..r
    int ring_rows = 0, row0 = 0;
    size_t out_clip_stride = 0;
};
bool fe2_subset_supported(const FeParams& p, int mel_mode);
hipError_t fe2_launch(const int16_t* d_pcm, size_t row_stride, int B, int N, int T, const FeParams& p,
                      const FeTables* d_tables, const Fe2MelPlan* d_plan, float* d_db, float* d_mel, int frames_major,
                      int mel_mode, int max_taps, int block, int max_grid, hipStream_t stream, const Fe2Sub* subset = nullptr);

