// emb_stream.hip - device-resident state of the reference's embedding-mode preprocessor for S lock-step streams
// (reference: nanowakeword/data/AudioFeatures.py).  The two ONNX models the reference runs there
// (melspectrogram.onnx, embedding_model.onnx) are un-vendored binaries and stay pluggable; everything BETWEEN them
// is reference code and lives here:
//   mel ring      [S][mel_cap = 970][bins = 32]   melspectrogram_buffer: starts as ones((76, 32)) (:107,119), grows by
//                                                 the mel model's frames with x/10 + 2 applied (:124,146), keeps the
//                                                 newest 970 frames (:397-398)
//   windows       76 frames every 8, newest last  (:434-440 streaming; :168-179, 261-272 batch)
//   feature ring  [S][feat_cap = 120][D = 96]     feature_buffer, newest 120 rows (:446-447); get_features(n) = last n rows (:451-457)
// All kernels are pure data movement (HBM-bound, a few KB per stream and hop): one lane per float, coalesced along
// the innermost (bins / embedding) dimension.
#include <hip/hip_runtime.h>
#include "emb_stream.h"

namespace {

__device__ __forceinline__ int ring_index(int pos, int len, int cap, int logical) {
    int p = pos - len + logical;      // logical 0 = oldest valid row
    p %= cap;
    return p < 0 ? p + cap : p;
}

// src [S][n][bins] -> ring rows (pos + f) % cap; raw != 0 applies the reference's melspec_transform x/10 + 2
__global__ void __launch_bounds__(256) emb_push_rows_kernel(float* __restrict__ ring, const float* __restrict__ src, int S, int cap,
                                                            int width, int pos, int n, int skip, int raw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)(n - skip) * width;
    if (idx >= (size_t)S * per) return;
    const int s = (int)(idx / per);
    const size_t r = idx - (size_t)s * per;
    const int f = (int)(r / width) + skip, b = (int)(r % width);
    float v = src[((size_t)s * n + f) * width + b];
    if (raw) v = v / 10.0f + 2.0f;                                   // AudioFeatures.py:124 (float32, true division)
    ring[((size_t)s * cap + (pos + f - skip) % cap) * width + b] = v;
}

__global__ void __launch_bounds__(256) emb_fill_kernel(float* __restrict__ p, size_t n, float v) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) p[idx] = v;
}

// out [S][nw][76][bins]: window w (oldest first) ends 8 * (nw - 1 - w) frames before the newest frame
__global__ void __launch_bounds__(256) emb_windows_kernel(const float* __restrict__ ring, float* __restrict__ out, int S, int cap,
                                                          int bins, int pos, int len, int nw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_w = (size_t)EMB_WINDOW * bins;
    if (idx >= (size_t)S * nw * per_w) return;
    const int s = (int)(idx / (nw * per_w));
    size_t r = idx - (size_t)s * nw * per_w;
    const int w = (int)(r / per_w);
    r -= (size_t)w * per_w;
    const int row = (int)(r / bins), b = (int)(r % bins);
    const int logical = len - EMB_STEP * (nw - 1 - w) - EMB_WINDOW + row;
    out[idx] = ring[((size_t)s * cap + ring_index(pos, len, cap, logical)) * bins + b];
}

// out [S][n][D] = the newest n rows of the ring, oldest first
__global__ void __launch_bounds__(256) emb_tail_kernel(const float* __restrict__ ring, float* __restrict__ out, int S, int cap, int width,
                                                       int pos, int len, int n) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)n * width;
    if (idx >= (size_t)S * per) return;
    const int s = (int)(idx / per);
    const size_t r = idx - (size_t)s * per;
    const int row = (int)(r / width), b = (int)(r % width);
    out[idx] = ring[((size_t)s * cap + ring_index(pos, len, cap, len - n + row)) * width + b];
}

// batch path: mel [B][F][bins] -> windows [B][W][76][bins], W = (F - 76) / 8 + 1 (windows that would run past F are dropped)
__global__ void __launch_bounds__(256) emb_window_batch_kernel(const float* __restrict__ mel, float* __restrict__ out, int B, int F,
                                                               int bins, int W) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_w = (size_t)EMB_WINDOW * bins;
    if (idx >= (size_t)B * W * per_w) return;
    const int bi = (int)(idx / (W * per_w));
    size_t r = idx - (size_t)bi * W * per_w;
    const int w = (int)(r / per_w);
    r -= (size_t)w * per_w;
    out[idx] = mel[((size_t)bi * F + EMB_STEP * w) * bins + r];      // rows of a window are contiguous in mel
}

// ragged mel spectrograms (packed back to back, row offsets in `start`, frames in `frames`) -> [B][Fmax][bins], padded with
// `pad` (-80: AudioFeatures.py:221); raw != 0 applies x/10 + 2 first
__global__ void __launch_bounds__(256) emb_pad_batch_kernel(const float* __restrict__ packed, const int* __restrict__ start,
                                                            const int* __restrict__ frames, float* __restrict__ out, int B,
                                                            int Fmax, int bins, float pad, int raw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)Fmax * bins;
    if (idx >= (size_t)B * per) return;
    const int bi = (int)(idx / per);
    const size_t r = idx - (size_t)bi * per;
    const int f = (int)(r / bins), b = (int)(r % bins);
    float v = pad;
    if (f < frames[bi]) {
        v = packed[((size_t)start[bi] + f) * bins + b];
        if (raw) v = v / 10.0f + 2.0f;
    }
    out[idx] = v;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t emb_alloc(EmbState* e) {
    hipError_t rc = hipMalloc(&e->mel_ring, (size_t)e->S * e->mel_cap * e->bins * sizeof(float));
    if (rc != hipSuccess) return rc;
    return hipMalloc(&e->feat_ring, (size_t)e->S * e->feat_cap * e->D * sizeof(float));
}

void emb_free(EmbState* e) {
    if (e->mel_ring) (void)hipFree(e->mel_ring);
    if (e->feat_ring) (void)hipFree(e->feat_ring);
    if (e->stage) (void)hipFree(e->stage);
    e->mel_ring = e->feat_ring = e->stage = nullptr;
    e->stage_floats = 0;
}

hipError_t emb_reset(EmbState* e, hipStream_t s) {
    // melspectrogram_buffer = np.ones((76, 32)) (AudioFeatures.py:107,119); the caller re-seeds the feature ring
    // (the reference fills it with embeddings of 4 s of random noise, :112,121)
    const size_t n = (size_t)e->S * e->mel_cap * e->bins;
    hipLaunchKernelGGL(emb_fill_kernel, dim3(blocks(n)), dim3(256), 0, s, e->mel_ring, n, 1.0f);
    e->mel_len = EMB_WINDOW < e->mel_cap ? EMB_WINDOW : e->mel_cap;
    e->mel_pos = e->mel_len % e->mel_cap;
    e->feat_len = e->feat_pos = 0;
    return hipGetLastError();
}

static hipError_t push_rows(float* ring, const float* d_src, int S, int cap, int width, int* pos, int* len, int n, int raw, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int skip = n > cap ? n - cap : 0;                           // only the newest `cap` rows can survive
    const size_t total = (size_t)S * (n - skip) * width;
    hipLaunchKernelGGL(emb_push_rows_kernel, dim3(blocks(total)), dim3(256), 0, s, ring, d_src, S, cap, width, *pos, n, skip, raw);
    *pos = (*pos + (n - skip)) % cap;
    *len = (*len + n > cap) ? cap : *len + n;
    return hipGetLastError();
}

hipError_t emb_push_mel(EmbState* e, const float* d_src, int n, int raw, hipStream_t s) {
    return push_rows(e->mel_ring, d_src, e->S, e->mel_cap, e->bins, &e->mel_pos, &e->mel_len, n, raw, s);
}

hipError_t emb_push_feat(EmbState* e, const float* d_src, int k, hipStream_t s) {
    return push_rows(e->feat_ring, d_src, e->S, e->feat_cap, e->D, &e->feat_pos, &e->feat_len, k, 0, s);
}

int emb_valid_windows(const EmbState* e, int n_chunks) {
    if (e->mel_len < EMB_WINDOW) return 0;
    const int most = (e->mel_len - EMB_WINDOW) / EMB_STEP + 1;
    return n_chunks < most ? n_chunks : most;
}

hipError_t emb_windows(const EmbState* e, int nw, float* d_out, hipStream_t s) {
    if (nw <= 0) return hipSuccess;
    const size_t total = (size_t)e->S * nw * EMB_WINDOW * e->bins;
    hipLaunchKernelGGL(emb_windows_kernel, dim3(blocks(total)), dim3(256), 0, s, e->mel_ring, d_out, e->S, e->mel_cap, e->bins, e->mel_pos,
                       e->mel_len, nw);
    return hipGetLastError();
}

hipError_t emb_tail_features(const EmbState* e, int n, float* d_out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const size_t total = (size_t)e->S * n * e->D;
    hipLaunchKernelGGL(emb_tail_kernel, dim3(blocks(total)), dim3(256), 0, s, e->feat_ring, d_out, e->S, e->feat_cap, e->D, e->feat_pos,
                       e->feat_len, n);
    return hipGetLastError();
}

hipError_t emb_window_batch(const float* d_mel, int B, int F, int bins, float* d_out, hipStream_t s) {
    const int W = (F - EMB_WINDOW) / EMB_STEP + 1;
    const size_t total = (size_t)B * W * EMB_WINDOW * bins;
    hipLaunchKernelGGL(emb_window_batch_kernel, dim3(blocks(total)), dim3(256), 0, s, d_mel, d_out, B, F, bins, W);
    return hipGetLastError();
}

hipError_t emb_pad_batch(const float* d_packed, const int* d_start, const int* d_frames, float* d_out, int B, int Fmax, int bins,
                         float pad, int raw, hipStream_t s) {
    const size_t total = (size_t)B * Fmax * bins;
    hipLaunchKernelGGL(emb_pad_batch_kernel, dim3(blocks(total)), dim3(256), 0, s, d_packed, d_start, d_frames, d_out, B, Fmax, bins, pad, raw);
    return hipGetLastError();
}
