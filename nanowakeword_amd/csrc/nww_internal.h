// nww_internal.h - what the translation units of the C-ABI share: the handle, a forward's run state, error helpers and the
// device-side entry points they call on each other.  nww_api.hip: create / load / run entry points; nww_plan.hip: state_dict spec
// and the per-head launch plans (nww_finalize); nww_stream.hip: batched streaming (nww_stream_*); nww_comm.hip: RCCL gather;
// nww_emb.hip: embedding-mode state (nww_emb_*).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/nww.h"
#include "fe_tables.h"
#include "frontend.h"
#include "layers.h"
#include "trunk.h"
#include "ffn_x3.h"
#include "lin_x3.h"
#include "attn_x3.h"
#include "dual_x3.h"
#include "emb_stream.h"


struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool loaded = false;
    size_t dev_off = 0;     // float offset in the weight arena
};

struct Step {
    std::string name;
    std::function<hipError_t(struct Run&)> fn;
};

struct Run {
    int B = 0;
    hipStream_t stream = nullptr;
    const float* x = nullptr;   // head input [B][in_rows*in_cols]
    size_t x_stride = 0;        // floats between consecutive clips of x; 0 = dense (only first steps that say so at plan time take a stride)
    // streaming hop (nww_stream.hip; set only for plans with nww_handle::stream_conv): 1 = the fused trunk writes its pooled rows
    // into per-stream rings and the third conv reads them there, 2 = and only the rows a hop invalidates are computed
    int stream_mode = 0;
    float* a2_ring = nullptr; int a2_rows = 0, a2_row0 = 0; size_t a2_ch_stride = 0, a2_clip_stride = 0;
    int a2_nsub = 0, a2_sub_a[4] = {0, 0, 0, 0}, a2_sub_b[4] = {0, 0, 0, 0};
    // the third conv's sequence output lives in two per-stream buffers that alternate from hop to hop: rows [a3_lo, a3_hi] of the new
    // one are rows + a3_shift of the previous one, the others are computed (stream_mode 2); null: the plan's workspace buffer
    float* seq_new = nullptr; const float* seq_prev = nullptr; int a3_lo = 0, a3_hi = -1, a3_shift = 0;
    bool x_frames_major = false; // E2E head on the transposed plane: x came from the frontend as [B][frames][n_mels] already
    float* buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float* emb = nullptr;       // [B][E]
    float* hid = nullptr;       // [B][E/2]
    float* logits = nullptr;    // [B]
    float* probs = nullptr;     // [B] or null; a plan step that writes it clears `need_sigmoid`
    bool need_sigmoid = true;
    unsigned int* done_flag = nullptr; unsigned int done_seq = 0; bool done_armed = false;   // zero-copy small calls (classifier tail's completion word)
    float* splitk_ws = nullptr; size_t splitk_floats = 0; int cu_count = 256;
    // a split-K GEMM that left its partials for the classifier tail to reduce (GemmArgs::defer_reduce)
    struct { bool active = false; int out_id = 0, parts = 0; size_t stride = 0; const float *bias = nullptr, *alpha = nullptr, *beta = nullptr; int act = 0; } deferred;
};


// key of the packed-weight cache: the weight, the image format (0 = bf16 terms, 1 = scaled binary16 terms) and the latter's scale
struct X3Key {
    const float* w; int fmt; float scale;
    bool operator<(const X3Key& o) const { return w != o.w ? w < o.w : fmt != o.fmt ? fmt < o.fmt : scale < o.scale; }
};

struct nww_handle {
    bool e2e_transposed = false;                   // the E2E plan runs on the (frames, n_mels) plane: the frontend writes frames-major for it
    nww_config cfg;
    FeParams fe;
    std::string err;
    std::vector<std::string> keys;                 // required state_dict keys, in order
    std::map<std::string, HostTensor> tensors;     // required + optional + derived
    bool finalized = false;
    float* d_weights = nullptr;
    FeTables* d_tables = nullptr;
    Fe2MelPlan* d_melplan = nullptr;
    int mel_max_taps = 0;          // longest filter support of the mel filterbank
    hipStream_t own_stream = nullptr;
    std::vector<Step> plan;
    size_t buf_per_clip[6] = {0, 0, 0, 0, 0, 0};   // floats per clip of each workspace buffer
    // workspace (grown on demand)
    int cap_B = 0, cap_N = 0;
    bool trunk_blocked = false;    // CNN head: the fused trunk writes fc1's A operand as [128][32] tiles (decided at plan time)
    int cap_rows = 0;              // cap_B rounded up to 128: the blocked trunk -> fc1 buffer is written in 128-clip row blocks
    float* d_ws = nullptr;
    int16_t* d_pcm = nullptr;
    float* d_logmel = nullptr;     // [B][n_mels*frames]
    float* d_feats = nullptr;      // staging for host feature input
    float* d_emb = nullptr;
    float* d_hid = nullptr;
    float* d_logits = nullptr;
    float* d_probs = nullptr;
    float* d_splitk = nullptr;     // split-K partials
    // streaming rings: [S][2*W] int16, sample p of a stream lives at p and p+W
    int16_t* d_ring = nullptr; int16_t* d_chunk = nullptr;
    // incremental hops (nww_stream.hip): log-mel ring [S][2 * lm_rows][n_mels] (frame t of the current window at row lm_pos + t and
    // lm_rows away), pooled conv rows [S][32][a2_rows][W / 4] (row R at (a2_pos + R) % a2_rows); a hop shifts them by lm_shift / a2_shift
    std::string plan_error;        // plan time: a step that cannot be planned in ANY form (nww_finalize fails with it)
    bool clamps_features = false;  // plan time: a step that reads the head input clamps it to +-NWW_F16_FEATURE_BOUND (nww_feature_clamp)
    bool x_stride_ok = false;      // plan time: the first step takes Run::x_stride (fused split-operand trunk, DNN layer1)
    bool stream_conv = false;      // plan time: fused trunk -> conv3_x3 pair that takes Run::stream_mode (CRNN)
    int stream_H = 0, stream_W = 0;   // the plane that pair works on
    bool inc_fe = false, inc_conv = false, primed = false;
    float* d_lm_ring = nullptr; int lm_rows = 0, lm_pos = 0, lm_shift = 0, fe_edge_l = 0, fe_edge_r = 0;
    float* d_a2_ring = nullptr; int a2_rows = 0, a2_pos = 0, a2_shift = 0, a2_lo = 0, a2_hi = 0;
    int a2_nsub = 0, a2_sub_a[4] = {0, 0, 0, 0}, a2_sub_b[4] = {0, 0, 0, 0};   // the strips of pooled rows a hop recomputes, each cut to fit in LDS (plan_incremental)
    bool stream_seq = false;       // plan time: that third conv writes the recurrent layers' sequence layout (its rows can be carried over)
    size_t seq_floats = 0;         // ... floats per clip of it
    float* d_seq[2] = {nullptr, nullptr}; int seq_cur = 0, a3_lo = 0, a3_hi = -1, a3_shift = 0;
    struct { bool on = false; size_t x_stride = 0; int mode = 0; } sr;    // what the next nww_run_head hands to its Run
    EmbState* emb = nullptr;       // embedding-mode preprocessor state (nww_emb_*)
    void* comm = nullptr;          // ncclComm_t of this rank (nww_comm_init)
    unsigned char* pin_in = nullptr; unsigned char* pin_out = nullptr;   // pinned staging for small host-pointer calls
    bool pin_in_busy = false;                                            // an async copy out of pin_in may still be in flight
    unsigned int done_seq = 0;                                           // completion-word sequence of the zero-copy small calls
    int comm_rank = 0, comm_world = 1;
    // the gather's own stream (nww_forward_pcm_gather_async_dev): step k's all-gather runs there behind ev_ready[k & 1] while step
    // k + 1's kernels run on the caller's stream; ev_gathered[k & 1] closes it, ev_start[k & 1] stamps the step's start (overlap probe)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_gathered[2] = {nullptr, nullptr}, ev_start[2] = {nullptr, nullptr};
    unsigned long long gather_seq = 0;
    const float* gather_buf[2] = {nullptr, nullptr};   // the d_all_logits each slot's gather last wrote
    long long gather_test_delay_us = 0;                // test hook, read ONCE at nww_comm_init (NWW_GATHER_TEST_DELAY_US)
    int ring_S = 0, ring_W = 0, ring_hop = 0, ring_pos = 0; long long ring_filled = 0;
    size_t splitk_per_clip = 0;    // floats per clip (max over the plan's split GEMMs)
    std::map<X3Key, void*> x3_weights;             // GEMM weights pre-split into bf16 terms / scaled binary16 terms (gemm_x3.hip)
    std::vector<void*> packed_weights;             // other plan-time weight packings (ffn_x3.hip)
    int conv_products = 0;                         // fused trunk: 0 = float32 MFMA, 6 | 9 = bf16 split products
    bool f16 = false;                              // NWW_ARITH_F16X3: conv_products = 6, and the layers that have a two-term binary16 instance and a bound on their input use it
    int cu_count = 256;
    // profiling: per forward, events[0..n] bracket the n launches; accumulated on nww_get_profile
    bool profiling = false;
    int prof_period = 1, prof_counter = 0;   // sampling: only every prof_period-th forward records events
    bool prof_active = false;
    std::vector<std::vector<hipEvent_t>> prof_runs;   // one event list per recorded forward
    std::vector<std::vector<int>> prof_ids;           // plan-entry id of each interval
    std::vector<hipEvent_t> event_pool;
    std::vector<double> prof_ms;                      // size plan+2 : [0]=frontend, [1..n]=plan, [n+1]=sigmoid
    std::vector<int> prof_cnt;
};


// ---- shared helpers (defined in nww_api.hip unless noted)
int nww_fail(nww_handle* h, int code, const char* fmt, ...);
std::string& nww_create_err();                 // nww_create / communicator errors: per thread
void nww_prof_mark(nww_handle* h, hipStream_t s, int id_of_next);
void nww_prof_begin(nww_handle* h);
int nww_check_run(nww_handle* h, int B);
int nww_ensure_ws(nww_handle* h, int B, int N);
int nww_run_head(nww_handle* h, const float* d_x, int B, float* d_logits, float* d_probs, hipStream_t s, unsigned int* done_flag = nullptr,
                 unsigned int done_seq = 0, bool* done_armed = nullptr, bool x_frames_major = false);
int nww_frontend_on_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_db, float* d_mel, int frames_major, hipStream_t s,
                        int* frames_out, size_t row_stride = 0, const Fe2Sub* sub = nullptr);
int nww_forward_pcm_on_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_logits, float* d_probs, hipStream_t s,
                           size_t row_stride = 0, unsigned int* done_flag = nullptr, unsigned int done_seq = 0, bool* done_armed = nullptr);
int nww_h2d_small(nww_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s);
int nww_copy_out(nww_handle* h, int B, float* logits, float* probs, float* emb, hipStream_t s);
void nww_build_spec(nww_handle* h);            // nww_plan.hip

#define fail nww_fail
#define HIP_TRY(h, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(h, NWW_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

