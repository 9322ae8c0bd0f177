/*
 * nww.h - C-ABI of libnwwhip.so: the MI355X (gfx950) implementation of nanowakeword's
 * batched keyword-spotting hot path  int16 PCM -> STFT -> mel -> dB -> classifier head -> logit.
 *
 * The reference (arcosoph/nanowakeword v3.0.0) is pure Python and has no FFI of its own; its
 * hot path sits behind a duck-typed onnxruntime.InferenceSession:
 *     session.get_inputs()[0].{name,shape}        nanointerpreter.py:165-167,177-178
 *     session.run(None, {"input": x}) -> [probs(B,1,1)]   nanointerpreter.py:677,681,783
 * with `_RemoteSession` (remote_verifier.py:490-648) as the in-tree precedent of a non-ORT
 * backend.  This header is what a ctypes binding of such a backend binds;
 * nanowakeword_amd/session.py is that binding.
 *
 * Conventions: extern "C", opaque handle, int return codes (0 = ok), no exceptions or C++ types
 * across the boundary.  Buffers are caller-owned.  `*_host` entry points take host pointers and
 * synchronise before returning; `*_dev` entry points take device pointers, enqueue on the given
 * hipStream_t (NULL = the handle's own stream) and do NOT synchronise.  A handle is bound to one
 * GPU and is not re-entrant (the reference interpreter is single-threaded too:
 * nanointerpreter.py:955-959); use one handle per GPU / per thread.
 */
#ifndef NWW_H
#define NWW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NWW_OK 0
#define NWW_ERR_INVALID 1      /* bad argument (ValueError in the reference's terms)          */
#define NWW_ERR_MISSING 2      /* finalize(): a state_dict tensor was never loaded            */
#define NWW_ERR_SHAPE 3        /* load_tensor(): shape differs from Model.state_dict()        */
#define NWW_ERR_HIP 4          /* a HIP runtime call failed; see nww_last_error               */
#define NWW_ERR_STATE 5        /* call order: run before finalize, load after finalize        */
#define NWW_ERR_UNSUPPORTED 6

/* head_type: which nanowakeword/modules/architectures.py class the handle evaluates */
#define NWW_HEAD_DNN 0         /* Net                     architectures.py:102-126 */
#define NWW_HEAD_CNN 1         /* CNNModel                architectures.py:51-80   */
#define NWW_HEAD_CRNN 2        /* CRNNModel (rnn=gru)     architectures.py:209-287 */
#define NWW_HEAD_GRU 3         /* GRUModel                architectures.py:129-145 */
#define NWW_HEAD_BCRESNET 4    /* BcResNetModel           architectures.py:620-687 */
#define NWW_HEAD_CONFORMER 5   /* ConformerModel          architectures.py:441-543 */
#define NWW_HEAD_E2E_DNN 6     /* E2E_MelSpectrogram_CNN  architectures.py:820-889 */

#define NWW_ACT_RELU 0         /* model.py:81-87 activation_function */
#define NWW_ACT_GELU 1
#define NWW_ACT_SILU 2

#define NWW_DTYPE_F32 0

typedef struct nww_handle nww_handle;

typedef struct nww_config {
    int32_t device;            /* HIP device ordinal                                          */
    /* frontend: T.MelSpectrogram(...) + T.AmplitudeToDB() of architectures.py:830-837,
       evaluated as the exported ONNXSafeMelSpectrogram (_export/onnx.py:27-83)               */
    int32_t sample_rate;       /* 16000                                                       */
    int32_t n_fft;             /* 400 (only 400 = 8*25*2 is implemented)                      */
    int32_t win_length;        /* 400                                                         */
    int32_t hop_length;        /* 160                                                         */
    int32_t n_mels;            /* <= 128                                                      */
    int32_t center;            /* 1: reflect-pad n_fft/2 (reference e2e), 0: no padding       */
    float f_min, f_max;        /* 0, sample_rate/2                                            */
    float amin;                /* 1e-10 clamp floor (-100 dB)                                 */
    float db_multiplier;       /* 10 (power spectrogram)                                      */
    /* head: kwargs/config keys of Model() (model.py:67-296)                                  */
    int32_t head_type;         /* NWW_HEAD_*                                                  */
    int32_t in_rows, in_cols;  /* Model(input_shape=(in_rows, in_cols))                       */
    int32_t layer_dim, n_blocks, embedding_dim, activation;
    int32_t n_crnn_channels;   /* crnn_cnn_channels (<= 4 stages)                             */
    int32_t crnn_channels[4];
    int32_t conformer_d_model, conformer_n_head;
    /* how nww_forward_pcm feeds the head: 0 = log-mel transposed to (frames, n_mels) =
       Model(input_shape=(frames, n_mels)); 1 = (n_mels, frames) as E2E_MelSpectrogram_CNN.   */
    int32_t mel_major_features;
    /* arithmetic of the large MFMA contractions - conv2 of the fused conv trunk (CNN / CRNN / E2E heads) and Linear
       layers with K >= 4096 (fc1 of the CNN head); every mode computes float32 results from float32 data, they differ
       in how the products are formed:
         NWW_ARITH_F32    v_mfma_f32_32x32x2_f32, one fmaf chain per output
         NWW_ARITH_BF16X9 each float32 operand split into three bf16 terms (exact), all nine partial products on
                          v_mfma_f32_32x32x16_bf16 - exact products, float32 accumulation
         NWW_ARITH_BF16X6 the six largest partial products; the dropped ones are < 2^-23 of a product
         NWW_ARITH_F16X3  each operand, scaled by a power of two fixed at nww_finalize from bounds on the tensors (features
                          are clamped to +-NWW_F16_FEATURE_BOUND = 8192: log-mel dB values lie in [-100, 60]; the accuracy does not depend on how generous the bound is), split into TWO
                          binary16 terms holding 22-23 of its 24 significant bits; the three partial products >= 2^-22 of a
                          product on v_mfma_f32_32x32x16_f16, float32 accumulation - half the matrix instructions of
                          BF16X6 at the float32 MFMA's accuracy against float64.  Layers without an f16x3 instance, or
                          without a bound on their input, run as NWW_ARITH_BF16X6
         NWW_ARITH_DEFAULT lets the library choose: NWW_ARITH_F16X3 (nww_api.hip NWW_DEFAULT_CONV_ARITH).  Under it the
                          head input is clamped to +-NWW_F16_FEATURE_BOUND - also for nww_forward_features* callers whose
                          features are not log-mel dB (the reference applies no clamp: pass NWW_ARITH_BF16X6 for
                          unbounded features)                                                                    */
    int32_t conv_arith;
    /* recurrent backend of the CRNN head: 0 = GRU, 1 = LSTM (the reference's default, modules/model.py:214;
       CRNNModel, modules/architectures.py:238-254)                                                               */
    int32_t crnn_rnn_lstm;
    /* storage type of the activation tensors BETWEEN the head's kernels: NWW_ACT_DTYPE_F32 (default), NWW_ACT_DTYPE_BF16 (round to
       nearest even on store; products and accumulation stay float32) or NWW_ACT_DTYPE_F16 (binary16 of value x a power of two
       fixed per tensor at plan time from a bound on it, round to nearest even, saturating; same bytes and speed as bf16, 11
       significant bits instead of 8).  Both are opt-ins for the BcResNet head (BASELINE.json config 3 "bf16 activations"):
       bf16 logits agree with the float32 reference to ~2e-2 on the test clips except all-zero PCM (0.2); f16 logits to ~5e-3 on
       every clip.  f16 assumes features within +-NWW_F16_FEATURE_BOUND, like NWW_ARITH_F16X3.                               */
    int32_t act_dtype;
    int32_t reserved[4];
} nww_config;
#define NWW_ACT_DTYPE_F32 0
#define NWW_ACT_DTYPE_BF16 1
#define NWW_ACT_DTYPE_F16 2
#define NWW_ARITH_DEFAULT 0
#define NWW_ARITH_F32 1
#define NWW_ARITH_F16X3 3
#define NWW_F16_FEATURE_BOUND 8192.0f
#define NWW_ARITH_BF16X6 6
#define NWW_ARITH_BF16X9 9

/* Fill *cfg with the reference defaults (16 kHz, 400/400/160, 64 mel, center, DNN (16,96)). */
void nww_default_config(nww_config* cfg);

int nww_create(const nww_config* cfg, nww_handle** out);
int nww_destroy(nww_handle* h);
const char* nww_last_error(const nww_handle* h);   /* h may be NULL: last create() error      */

/* Weights. `key` is exactly a Model.state_dict() key ("model.conv1.weight", "classifier.0.bias",
 * BatchNorm "running_mean"/"running_var"; "num_batches_tracked" is accepted and ignored).
 * Optional frontend tables taken from an exported model instead of the built-in torchaudio
 * formulas: "frontend.window" [win_length] and "frontend.mel_fb" [n_fft/2+1, n_mels]
 * (ONNXSafeMelSpectrogram buffers real_basis[0,0,:] and mel_fb, _export/onnx.py:62-64).     */
int nww_load_tensor(nww_handle* h, const char* key, const void* host_data,
                    const int64_t* shape, int32_t ndim, int32_t dtype);
/* Number of tensors finalize() requires, and the i-th key/shape (for loaders/validators).    */
int nww_num_tensors(const nww_handle* h);
int nww_tensor_info(const nww_handle* h, int32_t i, const char** key, int64_t* shape4, int32_t* ndim);
/* Check completeness, fold BatchNorm (alpha = w/sqrt(var+eps), beta = b - mean*alpha, as
 * PyTorch's eval kernel), build FFT/mel tables, upload.                                      */
int nww_finalize(nww_handle* h);

/* Frame law (bit-exact): center ? 1 + N/hop : 1 + (N - n_fft)/hop ; <0 if N is too short.    */
int32_t nww_num_frames(const nww_handle* h, int32_t n_samples);

/* ---- host-pointer entry points (synchronous) --------------------------------------------- */
/* pcm [B,N] int16 -> logmel [B, n_mels, frames] float32 dB (the mel module's own layout).    */
int nww_frontend(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N,
                 float* logmel_out, int32_t* frames_out);
/* same, additionally returning the mel power spectrogram [B, n_mels, frames] (may be NULL).  */
int nww_frontend_ex(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N,
                    float* logmel_out, float* melpower_out, int32_t* frames_out);
/* pcm [B,N] -> logits [B] (Model.forward) and/or probs [B] (InferenceWrapper sigmoid,
 * _export/onnx.py:164-172).  Either output may be NULL.                                      */
int nww_forward_pcm(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N,
                    float* logits, float* probs);
/* feats [B, in_rows, in_cols] float32 -> logits/probs; embedding [B, embedding_dim] optional. */
int nww_forward_features(nww_handle* h, const float* feats, int32_t B,
                         float* logits, float* probs);
int nww_forward_features_ex(nww_handle* h, const float* feats, int32_t B,
                            float* logits, float* probs, float* embedding);

/* ---- device-pointer entry points (asynchronous on `stream`, a hipStream_t) ----------------- */
int nww_frontend_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N,
                     float* d_logmel, int32_t frames_major, void* stream);
int nww_forward_pcm_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N,
                        float* d_logits, float* d_probs, void* stream);
int nww_forward_features_dev(nww_handle* h, const float* d_feats, int32_t B,
                             float* d_logits, float* d_probs, void* stream);
/* Pre-size the workspace for batches up to B clips of N samples (otherwise grown on demand,
 * which synchronises the device).                                                           */
int nww_reserve(nww_handle* h, int32_t B, int32_t N);

/* Names of the launches a forward_pcm performs, in order, one per line (rocprof correlation).  */
int nww_describe_plan(const nww_handle* h, char* buf, int32_t buflen);
/* The bound the finalized plan ASSUMES on the head's input features: +-NWW_F16_FEATURE_BOUND when a layer that reads them runs in the
   two-term binary16 arithmetic or 16-bit storage (values beyond it are clamped - the reference does not clamp), 0 when nothing is
   clamped.  Log-mel dB from nww_forward_pcm* always lies inside it; nww_forward_features* callers with other features can check:
   the Python host layer warns when a host feature array exceeds it.                                                              */
float nww_feature_clamp(const nww_handle* h);
/* Per-launch timing with HIP events ON THE STREAM THE KERNELS RUN ON.  While enabled, every
 * forward records one event per launch boundary (frontend, each head launch, sigmoid).
 * nww_get_profile synchronises those events and returns, per plan entry (same order as
 * nww_describe_plan), the accumulated milliseconds and the number of launches since the last
 * nww_set_profiling(h, 1).  n_inout: capacity in, entries out.  enable = n > 1 samples every n-th
 * forward only (each event costs the stream ~10 us; sampling keeps a timed loop undisturbed).    */
int nww_set_profiling(nww_handle* h, int32_t enable);
int nww_get_profile(nww_handle* h, float* ms_total, int32_t* launches, int32_t* n_inout);

/* ---- batched streaming (the front half of NanoInterpreter.predict in E2E mode, for S lock-step streams) ----
 * Replaces, per stream, the deque(maxlen=clip_samples) + "last clip_samples" window of
 * nanointerpreter.py:181,746-756 with a device-resident ring (double-written so the last `window` samples
 * are always contiguous).  Every push appends `hop` new samples per stream and re-scores the whole window,
 * exactly as the reference recomputes its window on every predict() call.  Scores are 0 until a stream has
 * seen `window` samples (nanointerpreter.py:755,785-786).  Post-filters stay on the host (a18).            */
int nww_stream_open(nww_handle* h, int32_t n_streams, int32_t window_samples, int32_t hop_samples);
/* chunk [n_streams][hop] int16 (host pointer); logits/probs [n_streams] host, either may be NULL.         */
int nww_stream_push(nww_handle* h, const int16_t* chunk, float* logits, float* probs);
/* device variant: d_chunk [n_streams][hop], outputs device pointers, enqueued on `stream`, no sync.       */
int nww_stream_push_dev(nww_handle* h, const int16_t* d_chunk, float* d_logits, float* d_probs, void* stream);
int nww_stream_reset(nww_handle* h);      /* new session for every stream (NanoInterpreter.reset, :719-733)    */
int nww_stream_close(nww_handle* h);
/* samples seen per stream since open/reset                                                               */
int64_t nww_stream_filled(const nww_handle* h);

/* ---- embedding-mode preprocessor state on the device (S lock-step streams) -------------------------------
 * Replaces the buffers and windowing of nanowakeword/data/AudioFeatures.py around its two ONNX models
 * (melspectrogram.onnx / embedding_model.onnx: un-vendored release binaries, pluggable here - the caller runs
 * them and hands their outputs over, host or device pointers):
 *   melspectrogram_buffer (np.ones((76,32)) at reset, newest 970 frames kept)      AudioFeatures.py:107,119,393-398
 *   melspec_transform x/10 + 2 (raw = 1)                                            :124,146
 *   76-frame windows every 8 frames, one per new 80 ms chunk, oldest first          :434-440
 *   feature_buffer (newest 120 rows) and get_features(n) = its last n rows          :446-457
 *   batch path: -80 padding to the longest clip, all windows of a clip              :188-227, 231-295
 * The handle must be a finalized feature-mode head with in_cols == emb_dim; nww_emb_forward scores every
 * stream on get_features(in_rows) without the features leaving the device.                                    */
int nww_emb_open(nww_handle* h, int32_t n_streams, int32_t mel_bins, int32_t emb_dim, int32_t mel_cap, int32_t feat_cap);
int nww_emb_reset(nww_handle* h);              /* mel ring = ones(76, bins), feature ring empty (re-seed it with nww_emb_push_features) */
int nww_emb_close(nww_handle* h);
int nww_emb_state(const nww_handle* h, int32_t* mel_frames, int32_t* feature_rows);
/* mel [n_streams][n_frames][mel_bins] float32: the mel model's output for the newest audio of every stream      */
int nww_emb_push_mel(nww_handle* h, const float* mel, int32_t n_frames, int32_t on_device, int32_t raw);
/* windows [n_streams][n_valid][76][mel_bins]: the windows of the n_chunks newest 80 ms chunks, oldest first       */
int nww_emb_windows(nww_handle* h, int32_t n_chunks, float* windows, int32_t on_device, int32_t* n_valid);
/* emb [n_streams][k][emb_dim]: the embedding model's rows for those windows                                       */
int nww_emb_push_features(nww_handle* h, const float* emb, int32_t k, int32_t on_device);
/* out [n_streams][n_out][emb_dim], n_out = min(n_frames, rows held)                                               */
int nww_emb_get_features(nww_handle* h, int32_t n_frames, float* out, int32_t on_device, int32_t* n_out);
/* logits / probs [n_streams] (host pointers, either may be NULL)                                                 */
int nww_emb_forward(nww_handle* h, float* logits, float* probs);
/* batch path. mel [B][F][bins] -> windows [B][n_windows][76][bins], n_windows = (F - 76) / 8 + 1                 */
int nww_emb_window_batch(nww_handle* h, const float* mel, int32_t B, int32_t F, int32_t bins, float* windows,
                         int32_t on_device, int32_t* n_windows);
/* packed: B ragged spectrograms back to back ([sum(frames)][bins], host) -> out [B][Fmax][bins] (host)           */
int nww_emb_pad_batch(nww_handle* h, const float* packed, const int32_t* frames, int32_t B, int32_t bins, int32_t Fmax,
                      float pad, int32_t raw, float* out);

/* ---- multi-GPU: batch split, one RCCL all-gather of the per-clip logits ------------------------
 * One process (and one handle) per GPU; every rank scores its own contiguous shard of the clips.  The path's only
 * exchange is an all-gather of 4 bytes per clip over RCCL / xGMI, enqueued on the same stream as the kernels.
 * The reference has no counterpart (it is single-process).  RCCL is bound at run time, so single-GPU users never
 * load it.  Rank 0 creates the 128-byte id and hands it to the other ranks by any means (file, socket, MPI,
 * torch.distributed - bench.py broadcasts it); then every rank calls nww_comm_init.                            */
#define NWW_COMM_ID_BYTES 128
int nww_comm_unique_id(void* id128);
int nww_comm_init(nww_handle* h, int32_t rank, int32_t world, const void* id128);
int nww_comm_destroy(nww_handle* h);
/* d_send [count] -> d_recv [world][count] on every rank; asynchronous on `stream`                                */
int nww_all_gather_logits(nww_handle* h, const float* d_send, float* d_recv, int32_t count, void* stream);
/* forward of this rank's B clips + all-gather into d_all_logits [world][B], one stream, no host hop               */
int nww_forward_pcm_gather_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N, float* d_all_logits, void* stream);
/* the same step with the all-gather on the handle's OWN stream behind an event: step k + 1's kernels (on `stream`) never wait for
   step k's RCCL latency.  Alternate two d_all_logits buffers; call k waits (device-side) for gather k - 2 before it reuses them.
   nww_gather_fence makes `stream` wait for every gather issued so far - after it the gathered vectors are valid in stream order.
   nww_gather_overlap_ms (host-synchronising, tests / tools): ms from the latest step's start to the end of the previous step's gather */
int nww_forward_pcm_gather_async_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N, float* d_all_logits, void* stream);
int nww_gather_fence(nww_handle* h, void* stream);
int nww_gather_overlap_ms(nww_handle* h, float* ms);

const char* nww_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NWW_H */
