// nww_emb.hip - embedding-mode preprocessor state on the device (nww_emb_*; kernels in emb_stream.hip).
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

// ------------------------------------------------------------------------------------------ embedding-mode state
// C-ABI over emb_stream.hip.  Host-pointer arguments are staged through e->stage; device-pointer arguments are used
// in place.  Everything runs on the handle's own stream and the host-pointer forms synchronise before returning.
static int emb_stage(nww_handle* h, size_t floats) {
    EmbState* e = h->emb;
    if (floats <= e->stage_floats) return NWW_OK;
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    if (e->stage) (void)hipFree(e->stage);
    e->stage = nullptr; e->stage_floats = 0;
    HIP_TRY(h, hipMalloc(&e->stage, floats * sizeof(float) + 16));
    e->stage_floats = floats;
    return NWW_OK;
}
#define EMB_CHECK(h)                                                                         \
    if (!(h) || !(h)->emb) return fail(h, NWW_ERR_STATE, "no embedding-mode state (nww_emb_open)"); \
    HIP_TRY(h, hipSetDevice((h)->cfg.device));

extern "C" int nww_emb_close(nww_handle* h) {
    if (!h) return NWW_ERR_INVALID;
    if (h->emb) {
        (void)hipSetDevice(h->cfg.device);
        emb_free(h->emb);
        delete h->emb;
        h->emb = nullptr;
    }
    return NWW_OK;
}

extern "C" int nww_emb_open(nww_handle* h, int32_t n_streams, int32_t mel_bins, int32_t emb_dim, int32_t mel_cap, int32_t feat_cap) {
    int rc = check_run(h, n_streams);
    if (rc) return rc;
    if (mel_bins <= 0 || emb_dim <= 0 || mel_cap < EMB_WINDOW || feat_cap <= 0)
        return fail(h, NWW_ERR_INVALID, "nww_emb_open: mel_bins, emb_dim, feat_cap must be positive and mel_cap >= 76");
    if (h->cfg.in_cols != emb_dim || h->cfg.in_rows > feat_cap)
        return fail(h, NWW_ERR_SHAPE, "the head expects (%d,%d) features; the embedding stream provides (<=%d, %d)", h->cfg.in_rows,
                    h->cfg.in_cols, feat_cap, emb_dim);
    nww_emb_close(h);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->emb = new EmbState();
    EmbState* e = h->emb;
    e->S = n_streams; e->bins = mel_bins; e->D = emb_dim; e->mel_cap = mel_cap; e->feat_cap = feat_cap;
    hipError_t er = emb_alloc(e);
    if (er == hipSuccess) er = emb_reset(e, h->own_stream);
    if (er != hipSuccess) { nww_emb_close(h); return fail(h, NWW_ERR_HIP, "nww_emb_open: %s", hipGetErrorString(er)); }
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return ensure_ws(h, n_streams, 0);
}

extern "C" int nww_emb_reset(nww_handle* h) {
    EMB_CHECK(h);
    HIP_TRY(h, emb_reset(h->emb, h->own_stream));
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return NWW_OK;
}

extern "C" int nww_emb_state(const nww_handle* h, int32_t* mel_frames, int32_t* feature_rows) {
    if (!h || !h->emb) return NWW_ERR_STATE;
    if (mel_frames) *mel_frames = h->emb->mel_len;
    if (feature_rows) *feature_rows = h->emb->feat_len;
    return NWW_OK;
}

static int emb_in(nww_handle* h, const float* src, size_t floats, int on_device, const float** d_src) {
    if (!src) return fail(h, NWW_ERR_INVALID, "null input pointer");
    if (on_device) { *d_src = src; return NWW_OK; }
    int rc = emb_stage(h, floats);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->emb->stage, src, floats * sizeof(float), hipMemcpyHostToDevice, h->own_stream));
    *d_src = h->emb->stage;
    return NWW_OK;
}

extern "C" int nww_emb_push_mel(nww_handle* h, const float* mel, int32_t n_frames, int32_t on_device, int32_t raw) {
    EMB_CHECK(h);
    if (n_frames <= 0) return fail(h, NWW_ERR_INVALID, "n_frames must be positive");
    EmbState* e = h->emb;
    const float* d = nullptr;
    int rc = emb_in(h, mel, (size_t)e->S * n_frames * e->bins, on_device, &d);
    if (rc) return rc;
    HIP_TRY(h, emb_push_mel(e, d, n_frames, raw, h->own_stream));
    if (!on_device) HIP_TRY(h, hipStreamSynchronize(h->own_stream));      // the staging buffer is reused by the next call
    return NWW_OK;
}

extern "C" int nww_emb_push_features(nww_handle* h, const float* emb, int32_t k, int32_t on_device) {
    EMB_CHECK(h);
    if (k <= 0) return fail(h, NWW_ERR_INVALID, "k must be positive");
    EmbState* e = h->emb;
    const float* d = nullptr;
    int rc = emb_in(h, emb, (size_t)e->S * k * e->D, on_device, &d);
    if (rc) return rc;
    HIP_TRY(h, emb_push_feat(e, d, k, h->own_stream));
    if (!on_device) HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return NWW_OK;
}

extern "C" int nww_emb_windows(nww_handle* h, int32_t n_chunks, float* windows, int32_t on_device, int32_t* n_valid) {
    EMB_CHECK(h);
    EmbState* e = h->emb;
    if (n_chunks <= 0 || !windows) return fail(h, NWW_ERR_INVALID, "nww_emb_windows: bad arguments");
    const int nw = emb_valid_windows(e, n_chunks);
    if (n_valid) *n_valid = nw;
    if (nw == 0) return NWW_OK;
    const size_t floats = (size_t)e->S * nw * EMB_WINDOW * e->bins;
    if (on_device) { HIP_TRY(h, emb_windows(e, nw, windows, h->own_stream)); return NWW_OK; }
    int rc = emb_stage(h, floats);
    if (rc) return rc;
    HIP_TRY(h, emb_windows(e, nw, e->stage, h->own_stream));
    HIP_TRY(h, hipMemcpyAsync(windows, e->stage, floats * sizeof(float), hipMemcpyDeviceToHost, h->own_stream));
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return NWW_OK;
}

extern "C" int nww_emb_get_features(nww_handle* h, int32_t n_frames, float* out, int32_t on_device, int32_t* n_out) {
    EMB_CHECK(h);
    EmbState* e = h->emb;
    if (n_frames <= 0 || !out) return fail(h, NWW_ERR_INVALID, "nww_emb_get_features: bad arguments");
    const int n = n_frames < e->feat_len ? n_frames : e->feat_len;       // feature_buffer[-n:] of a shorter buffer is the whole buffer
    if (n_out) *n_out = n;
    if (n == 0) return NWW_OK;
    const size_t floats = (size_t)e->S * n * e->D;
    if (on_device) { HIP_TRY(h, emb_tail_features(e, n, out, h->own_stream)); return NWW_OK; }
    int rc = emb_stage(h, floats);
    if (rc) return rc;
    HIP_TRY(h, emb_tail_features(e, n, e->stage, h->own_stream));
    HIP_TRY(h, hipMemcpyAsync(out, e->stage, floats * sizeof(float), hipMemcpyDeviceToHost, h->own_stream));
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return NWW_OK;
}

// head on get_features(in_rows) of every stream; the features never leave the device
extern "C" int nww_emb_forward(nww_handle* h, float* logits, float* probs) {
    EMB_CHECK(h);
    EmbState* e = h->emb;
    const int T = h->cfg.in_rows;
    if (e->feat_len < T) return fail(h, NWW_ERR_STATE, "the feature buffer holds %d rows, the head needs %d", e->feat_len, T);
    int rc = ensure_ws(h, e->S, 0);
    if (rc) return rc;
    hipStream_t s = h->own_stream;
    HIP_TRY(h, emb_tail_features(e, T, h->d_feats, s));
    prof_begin(h);
    rc = run_head(h, h->d_feats, e->S, h->d_logits, probs ? h->d_probs : nullptr, s);
    if (rc) return rc;
    return copy_out(h, e->S, logits, probs, nullptr, s);
}

// batch path (AudioFeatures._get_embeddings_batch, :231-295): mel [B][F][bins] -> windows [B][(F-76)/8+1][76][bins]
extern "C" int nww_emb_window_batch(nww_handle* h, const float* mel, int32_t B, int32_t F, int32_t bins, float* windows,
                                    int32_t on_device, int32_t* n_windows) {
    if (!h) return NWW_ERR_INVALID;
    if (B <= 0 || bins <= 0 || !mel || !windows) return fail(h, NWW_ERR_INVALID, "nww_emb_window_batch: bad arguments");
    if (F < EMB_WINDOW) return fail(h, NWW_ERR_INVALID, "Embedding model requires the input melspectrograms to have at least 76 frames");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int W = (F - EMB_WINDOW) / EMB_STEP + 1;
    if (n_windows) *n_windows = W;
    hipStream_t s = h->own_stream;
    if (on_device) { HIP_TRY(h, emb_window_batch(mel, B, F, bins, windows, s)); return NWW_OK; }
    const size_t n_in = (size_t)B * F * bins, n_out = (size_t)B * W * EMB_WINDOW * bins;
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(h, hipMalloc(&d_in, n_in * sizeof(float)));
    hipError_t er = hipMalloc(&d_out, n_out * sizeof(float));
    if (er == hipSuccess) er = hipMemcpyAsync(d_in, mel, n_in * sizeof(float), hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = emb_window_batch(d_in, B, F, bins, d_out, s);
    if (er == hipSuccess) er = hipMemcpyAsync(windows, d_out, n_out * sizeof(float), hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (er != hipSuccess) return fail(h, NWW_ERR_HIP, "nww_emb_window_batch: %s", hipGetErrorString(er));
    return NWW_OK;
}

// batch mel shaping (AudioFeatures._get_melspectrogram_batch, :188-227): B ragged spectrograms packed back to back ->
// [B][Fmax][bins] padded with `pad` (-80 in the reference); raw != 0 applies x/10 + 2 first.  Host pointers.
extern "C" int nww_emb_pad_batch(nww_handle* h, const float* packed, const int32_t* frames, int32_t B, int32_t bins, int32_t Fmax,
                                 float pad, int32_t raw, float* out) {
    if (!h) return NWW_ERR_INVALID;
    if (B <= 0 || bins <= 0 || Fmax <= 0 || !packed || !frames || !out) return fail(h, NWW_ERR_INVALID, "nww_emb_pad_batch: bad arguments");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    std::vector<int> start(B), fr(B);
    size_t total = 0;
    for (int i = 0; i < B; ++i) {
        if (frames[i] < 0 || frames[i] > Fmax) return fail(h, NWW_ERR_INVALID, "frames[%d] = %d outside 0..Fmax", i, frames[i]);
        start[i] = (int)total; fr[i] = frames[i]; total += frames[i];
    }
    hipStream_t s = h->own_stream;
    float *d_in = nullptr, *d_out = nullptr;
    int* d_idx = nullptr;
    const size_t n_out = (size_t)B * Fmax * bins;
    hipError_t er = hipMalloc(&d_in, (total * bins + 4) * sizeof(float));
    if (er == hipSuccess) er = hipMalloc(&d_out, n_out * sizeof(float));
    if (er == hipSuccess) er = hipMalloc(&d_idx, 2 * (size_t)B * sizeof(int));
    if (er == hipSuccess && total) er = hipMemcpyAsync(d_in, packed, total * bins * sizeof(float), hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipMemcpyAsync(d_idx, start.data(), B * sizeof(int), hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipMemcpyAsync(d_idx + B, fr.data(), B * sizeof(int), hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = emb_pad_batch(d_in, d_idx, d_idx + B, d_out, B, Fmax, bins, pad, raw, s);
    if (er == hipSuccess) er = hipMemcpyAsync(out, d_out, n_out * sizeof(float), hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_idx) (void)hipFree(d_idx);
    if (er != hipSuccess) return fail(h, NWW_ERR_HIP, "nww_emb_pad_batch: %s", hipGetErrorString(er));
    return NWW_OK;
}

