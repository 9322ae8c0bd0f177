// emb_stream.h - device state of the embedding-mode preprocessor (emb_stream.hip); see there for the reference map.
#pragma once
#include <hip/hip_runtime.h>

#define EMB_WINDOW 76     // frames per embedding window (AudioFeatures.py:168)
#define EMB_STEP 8        // frames between windows = one 80 ms chunk

struct EmbState {
    int S = 0, bins = 32, D = 96, mel_cap = 970, feat_cap = 120;
    int mel_len = 0, mel_pos = 0;       // valid frames, next write row (lock-step: the same for every stream)
    int feat_len = 0, feat_pos = 0;
    float* mel_ring = nullptr;          // [S][mel_cap][bins]
    float* feat_ring = nullptr;         // [S][feat_cap][D]
    float* stage = nullptr;             // staging for host-pointer entry points and the head's input
    size_t stage_floats = 0;
};

hipError_t emb_alloc(EmbState* e);
void emb_free(EmbState* e);
hipError_t emb_reset(EmbState* e, hipStream_t s);
hipError_t emb_push_mel(EmbState* e, const float* d_src, int n_frames, int raw, hipStream_t s);
hipError_t emb_push_feat(EmbState* e, const float* d_src, int k, hipStream_t s);
int emb_valid_windows(const EmbState* e, int n_chunks);
hipError_t emb_windows(const EmbState* e, int n_windows, float* d_out, hipStream_t s);
hipError_t emb_tail_features(const EmbState* e, int n, float* d_out, hipStream_t s);
hipError_t emb_window_batch(const float* d_mel, int B, int F, int bins, float* d_out, hipStream_t s);
hipError_t emb_pad_batch(const float* d_packed, const int* d_start, const int* d_frames, float* d_out, int B, int Fmax, int bins,
                         float pad, int raw, hipStream_t s);
