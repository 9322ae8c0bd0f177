// nww_stream.hip - batched streaming: S lock-step device rings, one score per stream and hop (nww_stream_*).
#include "nww_internal.h"
#define prof_mark nww_prof_mark
#define prof_begin nww_prof_begin
#define ensure_ws nww_ensure_ws
#define run_head nww_run_head
#define check_run nww_check_run
#define frontend_dev nww_frontend_on_dev
#define forward_pcm_dev nww_forward_pcm_on_dev
#define h2d_small nww_h2d_small
#define copy_out nww_copy_out

// ------------------------------------------------------------------------------------------ streaming
__global__ void __launch_bounds__(256)
stream_push_kernel(int16_t* __restrict__ ring, const int16_t* __restrict__ chunk, int S, int W, int hop, int pos) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)S * hop) return;
    const int s = (int)(idx / hop), j = (int)(idx - (size_t)s * hop);
    int p = pos + j;
    if (p >= W) p -= W;
    const int16_t v = chunk[idx];
    int16_t* r = ring + (size_t)s * 2 * W;
    r[p] = v;
    r[p + W] = v;
}

extern "C" int nww_stream_close(nww_handle* h) {
    if (!h) return NWW_ERR_INVALID;
    (void)hipSetDevice(h->cfg.device);
    if (h->d_ring) (void)hipFree(h->d_ring);
    if (h->d_chunk) (void)hipFree(h->d_chunk);
    h->d_ring = nullptr; h->d_chunk = nullptr; h->ring_S = h->ring_W = h->ring_hop = h->ring_pos = 0; h->ring_filled = 0;
    return NWW_OK;
}

extern "C" int nww_stream_open(nww_handle* h, int32_t S, int32_t W, int32_t hop) {
    int rc = check_run(h, S);
    if (rc) return rc;
    if (W <= 0 || hop <= 0 || hop > W || (W % 8) || (hop % 8))
        return fail(h, NWW_ERR_INVALID, "window and hop must be positive multiples of 8 samples with hop <= window");
    const nww_config& c = h->cfg;
    const int T = fe_num_frames(h->fe, W);
    const int rows = c.mel_major_features ? c.n_mels : T, cols = c.mel_major_features ? T : c.n_mels;
    if (T <= 0 || rows != c.in_rows || cols != c.in_cols)
        return fail(h, NWW_ERR_SHAPE, "a %d-sample window gives (%d,%d) features but the head expects (%d,%d)", W, rows, cols, c.in_rows, c.in_cols);
    nww_stream_close(h);
    HIP_TRY(h, hipSetDevice(c.device));
    HIP_TRY(h, hipMalloc(&h->d_ring, (size_t)S * 2 * W * sizeof(int16_t) + 16));
    HIP_TRY(h, hipMemset(h->d_ring, 0, (size_t)S * 2 * W * sizeof(int16_t)));
    HIP_TRY(h, hipMalloc(&h->d_chunk, (size_t)S * hop * sizeof(int16_t) + 16));
    h->ring_S = S; h->ring_W = W; h->ring_hop = hop; h->ring_pos = 0; h->ring_filled = 0;
    return ensure_ws(h, S, W);
}

extern "C" int nww_stream_reset(nww_handle* h) {
    if (!h || !h->d_ring) return fail(h, NWW_ERR_STATE, "no open stream batch");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemset(h->d_ring, 0, (size_t)h->ring_S * 2 * h->ring_W * sizeof(int16_t)));
    h->ring_pos = 0; h->ring_filled = 0;
    return NWW_OK;
}

extern "C" int64_t nww_stream_filled(const nww_handle* h) { return h ? h->ring_filled : 0; }

static int stream_push_dev(nww_handle* h, const int16_t* d_chunk, float* d_logits, float* d_probs, hipStream_t s) {
    const int S = h->ring_S, W = h->ring_W, hop = h->ring_hop;
    const size_t total = (size_t)S * hop;
    hipLaunchKernelGGL(stream_push_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->d_ring, d_chunk, S, W, hop, h->ring_pos);
    HIP_TRY(h, hipGetLastError());
    h->ring_pos = (h->ring_pos + hop) % W;
    h->ring_filled += hop;
    // the last W samples of every stream are contiguous at ring + pos (double-written ring)
    if (h->ring_filled < W) {       // window not full yet: the reference reports 0.0 (nanointerpreter.py:785-786)
        if (d_logits) HIP_TRY(h, hipMemsetAsync(d_logits, 0, (size_t)S * sizeof(float), s));
        if (d_probs) HIP_TRY(h, hipMemsetAsync(d_probs, 0, (size_t)S * sizeof(float), s));
        return NWW_OK;
    }
    return forward_pcm_dev(h, h->d_ring + h->ring_pos, S, W, d_logits, d_probs, s, (size_t)2 * W);
}

extern "C" int nww_stream_push_dev(nww_handle* h, const int16_t* d_chunk, float* d_logits, float* d_probs, void* stream) {
    if (!h || !h->d_ring) return fail(h, NWW_ERR_STATE, "no open stream batch (nww_stream_open)");
    if (!d_chunk) return fail(h, NWW_ERR_INVALID, "null chunk pointer");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return stream_push_dev(h, d_chunk, d_logits, d_probs, stream ? (hipStream_t)stream : h->own_stream);
}

extern "C" int nww_stream_push(nww_handle* h, const int16_t* chunk, float* logits, float* probs) {
    if (!h || !h->d_ring) return fail(h, NWW_ERR_STATE, "no open stream batch (nww_stream_open)");
    if (!chunk) return fail(h, NWW_ERR_INVALID, "Input audio must be a non-null int16 array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = h->own_stream;
    const int S = h->ring_S;
    { int rcs = h2d_small(h, h->d_chunk, chunk, (size_t)S * h->ring_hop * sizeof(int16_t), s); if (rcs) return rcs; }
    int rc = stream_push_dev(h, h->d_chunk, h->d_logits, h->d_probs, s);
    if (rc) return rc;
    return copy_out(h, S, logits, probs, nullptr, s);
}

