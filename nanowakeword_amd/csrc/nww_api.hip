// nww_api.hip - the C-ABI of include/nww.h: handle life cycle, weights, workspace, the forward entry points.
#include "nww_internal.h"

static hipEvent_t prof_event(nww_handle* h) {
    hipEvent_t e = nullptr;
    if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void nww_prof_mark(nww_handle* h, hipStream_t s, int id_of_next) {
    if (!h->profiling || !h->prof_active) return;
    hipEvent_t e = prof_event(h);
    if (!e) return;
    (void)hipEventRecord(e, s);
    h->prof_runs.back().push_back(e);
    h->prof_ids.back().push_back(id_of_next);         // interval that STARTS at this event (-1 = end)
}
void nww_prof_begin(nww_handle* h) {
    if (!h->profiling) return;
    h->prof_active = (h->prof_counter++ % h->prof_period) == 0;
    if (!h->prof_active) return;
    h->prof_runs.emplace_back();
    h->prof_ids.emplace_back();
}

static thread_local std::string g_create_err;   // nww_create errors: per thread, so concurrent creates do not race

std::string& nww_create_err() { return g_create_err; }

#undef fail
int nww_fail(nww_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}
using Shape = std::vector<int64_t>;
static size_t numel(const Shape& s) { size_t n = 1; for (auto v : s) n *= (size_t)v; return n; }
#define fail nww_fail
#define prof_mark nww_prof_mark
#define prof_begin nww_prof_begin
#define ensure_ws nww_ensure_ws
#define run_head nww_run_head
#define check_run nww_check_run
#define frontend_dev nww_frontend_on_dev
#define forward_pcm_dev nww_forward_pcm_on_dev
#define h2d_small nww_h2d_small
#define copy_out nww_copy_out

// ------------------------------------------------------------------------------------------ create / load
// The default is the two-term binary16 form (round 4): float32-grade against float64 in every head (tests/test_gpu_parity.py::
// test_arithmetic_modes_against_float64: no worse than 2x the float32 MFMA path + 2e-6) at half the matrix instructions of bf16x6.
// It clamps the head input to +-NWW_F16_FEATURE_BOUND (include/nww.h); bf16x6 / bf16x9 / f32 do not.
#define NWW_DEFAULT_CONV_ARITH NWW_ARITH_F16X3
extern "C" void nww_default_config(nww_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->sample_rate = 16000; c->n_fft = 400; c->win_length = 400; c->hop_length = 160; c->n_mels = 64; c->center = 1;
    c->f_min = 0.f; c->f_max = 8000.f; c->amin = 1e-10f; c->db_multiplier = 10.f;
    c->head_type = NWW_HEAD_DNN; c->in_rows = 16; c->in_cols = 96; c->layer_dim = 128; c->n_blocks = 1;
    c->embedding_dim = 64; c->activation = NWW_ACT_RELU;
    c->n_crnn_channels = 3; c->crnn_channels[0] = 16; c->crnn_channels[1] = 32; c->crnn_channels[2] = 32;
    c->conformer_d_model = 144; c->conformer_n_head = 4; c->mel_major_features = 0;
}

extern "C" const char* nww_version(void) { return "nwwhip 0.1.0 (gfx950)"; }

extern "C" const char* nww_last_error(const nww_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int nww_create(const nww_config* cfg, nww_handle** out) {
    if (!cfg || !out) return fail(nullptr, NWW_ERR_INVALID, "nww_create: null argument");
    *out = nullptr;
    const nww_config& c = *cfg;
    if (c.head_type < 0 || c.head_type > NWW_HEAD_E2E_DNN) return fail(nullptr, NWW_ERR_INVALID, "Unsupported model_type code %d", c.head_type);
    if (c.activation < 0 || c.activation > 2) return fail(nullptr, NWW_ERR_INVALID, "bad activation code %d", c.activation);
    if (c.conv_arith != NWW_ARITH_DEFAULT && c.conv_arith != NWW_ARITH_F32 && c.conv_arith != NWW_ARITH_BF16X6 && c.conv_arith != NWW_ARITH_BF16X9 &&
        c.conv_arith != NWW_ARITH_F16X3)
        return fail(nullptr, NWW_ERR_INVALID, "bad conv_arith code %d", c.conv_arith);
    if (c.in_rows <= 0 || c.in_cols <= 0 || c.embedding_dim < 2 || c.layer_dim <= 0 || c.n_blocks < 0)
        return fail(nullptr, NWW_ERR_INVALID, "bad head dimensions");
    if (c.n_fft != 400) return fail(nullptr, NWW_ERR_UNSUPPORTED, "only n_fft=400 is implemented (got %d)", c.n_fft);
    if (c.win_length <= 0 || c.win_length > c.n_fft) return fail(nullptr, NWW_ERR_INVALID, "win_length must be in 1..n_fft");
    if (c.hop_length <= 0 || (c.hop_length & 1)) return fail(nullptr, NWW_ERR_UNSUPPORTED, "hop_length must be positive and even");
    if (c.n_mels <= 0 || c.n_mels > FE_MAX_MELS) return fail(nullptr, NWW_ERR_INVALID, "n_mels must be in 1..%d", FE_MAX_MELS);
    if (c.head_type == NWW_HEAD_CRNN && (c.n_crnn_channels < 1 || c.n_crnn_channels > 4))
        return fail(nullptr, NWW_ERR_INVALID, "crnn_cnn_channels must have 1..4 stages");
    if ((c.head_type == NWW_HEAD_CRNN || c.head_type == NWW_HEAD_GRU) && c.layer_dim > 512)
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "recurrent hidden size (layer_dim = %d) must be <= 512", c.layer_dim);
    if (c.head_type == NWW_HEAD_CONFORMER && (c.conformer_n_head <= 0 || c.conformer_d_model % c.conformer_n_head))
        return fail(nullptr, NWW_ERR_INVALID, "conformer_d_model must be divisible by conformer_n_head");
    if (c.act_dtype != NWW_ACT_DTYPE_F32 && c.act_dtype != NWW_ACT_DTYPE_BF16 && c.act_dtype != NWW_ACT_DTYPE_F16)
        return fail(nullptr, NWW_ERR_INVALID, "act_dtype must be NWW_ACT_DTYPE_F32, NWW_ACT_DTYPE_BF16 or NWW_ACT_DTYPE_F16");
    if (c.act_dtype != NWW_ACT_DTYPE_F32 && c.head_type != NWW_HEAD_BCRESNET)
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "act_dtype = bf16 / f16 is implemented for the BcResNet head only (BASELINE config 3)");
    if (c.head_type == NWW_HEAD_CONFORMER && !mha_head_dim_supported(c.conformer_d_model / c.conformer_n_head))
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "attention head_dim %d is wider than the widest compiled kernel (128)",
                    c.conformer_d_model / c.conformer_n_head);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NWW_ERR_HIP, "no HIP device available (%s): libnwwhip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (c.device < 0 || c.device >= ndev) return fail(nullptr, NWW_ERR_INVALID, "device %d out of range (0..%d)", c.device, ndev - 1);
    nww_handle* h = new nww_handle();
    h->cfg = c;
    {   // conv_arith: explicit config > library default
        int mode = c.conv_arith;
        if (mode == NWW_ARITH_DEFAULT) mode = NWW_DEFAULT_CONV_ARITH;
        h->conv_products = (mode == NWW_ARITH_BF16X6 || mode == NWW_ARITH_F16X3) ? 6 : mode == NWW_ARITH_BF16X9 ? 9 : 0;
        h->f16 = mode == NWW_ARITH_F16X3;
    }
    h->fe.sample_rate = c.sample_rate; h->fe.n_fft = c.n_fft; h->fe.win_length = c.win_length; h->fe.hop = c.hop_length;
    h->fe.n_mels = c.n_mels; h->fe.center = c.center; h->fe.f_min = c.f_min; h->fe.f_max = c.f_max;
    h->fe.amin = c.amin; h->fe.db_mult = c.db_multiplier;
    nww_build_spec(h);
    if (hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_err = "hipSetDevice/hipStreamCreate failed";
        delete h;
        return NWW_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) == hipSuccess) h->cu_count = prop.multiProcessorCount;
    *out = h;
    return NWW_OK;
}

static void free_ws(nww_handle* h) {
    for (void* p : {(void*)h->d_ws, (void*)h->d_pcm, (void*)h->d_logmel, (void*)h->d_feats, (void*)h->d_emb,
                    (void*)h->d_hid, (void*)h->d_logits, (void*)h->d_probs, (void*)h->d_splitk})
        if (p) (void)hipFree(p);
    h->d_ws = nullptr; h->d_pcm = nullptr; h->d_logmel = nullptr; h->d_feats = nullptr; h->d_emb = nullptr;
    h->d_hid = nullptr; h->d_logits = nullptr; h->d_probs = nullptr; h->d_splitk = nullptr; h->cap_B = 0; h->cap_N = 0;
}

extern "C" int nww_stream_close(nww_handle* h);

extern "C" int nww_emb_close(nww_handle* h);
extern "C" int nww_comm_destroy(nww_handle* h);
extern "C" int nww_destroy(nww_handle* h) {
    if (!h) return NWW_OK;
    (void)hipSetDevice(h->cfg.device);
    free_ws(h);
    nww_stream_close(h);
    nww_emb_close(h);
    nww_comm_destroy(h);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->d_weights) (void)hipFree(h->d_weights);
    for (auto& kv : h->x3_weights) (void)hipFree(kv.second);
    for (void* d : h->packed_weights) (void)hipFree(d);
    if (h->d_tables) (void)hipFree(h->d_tables);
    if (h->d_melplan) (void)hipFree(h->d_melplan);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    for (auto& run : h->prof_runs) for (auto e : run) (void)hipEventDestroy(e);
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    delete h;
    return NWW_OK;
}

extern "C" int nww_num_tensors(const nww_handle* h) { return h ? (int)h->keys.size() : 0; }

extern "C" int nww_tensor_info(const nww_handle* h, int32_t i, const char** key, int64_t* shape4, int32_t* ndim) {
    if (!h || i < 0 || i >= (int)h->keys.size()) return NWW_ERR_INVALID;
    const std::string& k = h->keys[i];
    const HostTensor& t = h->tensors.at(k);
    if (key) *key = k.c_str();
    if (ndim) *ndim = (int)t.shape.size();
    if (shape4) for (size_t d = 0; d < 4; ++d) shape4[d] = d < t.shape.size() ? t.shape[d] : 1;
    return NWW_OK;
}

extern "C" int nww_load_tensor(nww_handle* h, const char* key, const void* host, const int64_t* shape, int32_t ndim,
                               int32_t dtype) {
    if (!h || !key || !host || !shape || ndim < 0) return fail(h, NWW_ERR_INVALID, "nww_load_tensor: null argument");
    if (h->finalized) return fail(h, NWW_ERR_STATE, "nww_load_tensor('%s') after nww_finalize", key);
    if (dtype != NWW_DTYPE_F32) return fail(h, NWW_ERR_UNSUPPORTED, "only float32 tensors are accepted");
    std::string k(key);
    const size_t nbt = std::strlen("num_batches_tracked");
    if (k.size() >= nbt && k.compare(k.size() - nbt, nbt, "num_batches_tracked") == 0) return NWW_OK;
    Shape s(shape, shape + ndim);
    if (k == "frontend.window") {
        if (ndim != 1 || s[0] != h->cfg.win_length) return fail(h, NWW_ERR_SHAPE, "frontend.window must be [%d]", h->cfg.win_length);
    } else if (k == "frontend.mel_fb") {
        if (ndim != 2 || s[0] != h->cfg.n_fft / 2 + 1 || s[1] != h->cfg.n_mels)
            return fail(h, NWW_ERR_SHAPE, "frontend.mel_fb must be [%d,%d]", h->cfg.n_fft / 2 + 1, h->cfg.n_mels);
        h->tensors[k].shape = s;
    } else {
        auto it = h->tensors.find(k);
        if (it == h->tensors.end()) return fail(h, NWW_ERR_INVALID, "unexpected state_dict key '%s' for this head", key);
        if (it->second.shape != s) {
            std::string want, got;
            for (auto v : it->second.shape) want += std::to_string(v) + ",";
            for (auto v : s) got += std::to_string(v) + ",";
            return fail(h, NWW_ERR_SHAPE, "size mismatch for %s: expected [%s] got [%s]", key, want.c_str(), got.c_str());
        }
    }
    HostTensor& t = h->tensors[k];
    t.shape = s;
    t.data.assign(static_cast<const float*>(host), static_cast<const float*>(host) + numel(s));
    t.loaded = true;
    return NWW_OK;
}

extern "C" int32_t nww_num_frames(const nww_handle* h, int32_t n) { return h ? fe_num_frames(h->fe, n) : -1; }

extern "C" int nww_set_profiling(nww_handle* h, int32_t enable) {
    if (!h) return NWW_ERR_INVALID;
    for (auto& run : h->prof_runs) for (auto e : run) h->event_pool.push_back(e);
    h->prof_runs.clear(); h->prof_ids.clear();
    h->prof_ms.assign(h->plan.size() + 2, 0.0);
    h->prof_cnt.assign(h->plan.size() + 2, 0);
    h->profiling = enable != 0;
    h->prof_period = enable > 1 ? enable : 1;     // enable = n > 1: sample every n-th forward (an event per launch boundary costs ~10 us)
    h->prof_counter = 0; h->prof_active = false;
    return NWW_OK;
}

extern "C" int nww_get_profile(nww_handle* h, float* ms_total, int32_t* launches, int32_t* n_inout) {
    if (!h || !n_inout) return NWW_ERR_INVALID;
    const int n = (int)h->plan.size() + 2;
    if (*n_inout < n || !ms_total || !launches) { *n_inout = n; return fail(h, NWW_ERR_INVALID, "profile buffers too small (need %d)", n); }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if ((int)h->prof_ms.size() != n) { h->prof_ms.assign(n, 0.0); h->prof_cnt.assign(n, 0); }
    for (size_t r = 0; r < h->prof_runs.size(); ++r) {
        auto& ev = h->prof_runs[r];
        auto& ids = h->prof_ids[r];
        if (!ev.empty()) HIP_TRY(h, hipEventSynchronize(ev.back()));
        for (size_t i = 0; i + 1 < ev.size(); ++i) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            if (ids[i] >= 0 && ids[i] < n) { h->prof_ms[ids[i]] += ms; h->prof_cnt[ids[i]] += 1; }
        }
        for (auto e : ev) h->event_pool.push_back(e);
    }
    h->prof_runs.clear(); h->prof_ids.clear();
    for (int i = 0; i < n; ++i) { ms_total[i] = (float)h->prof_ms[i]; launches[i] = h->prof_cnt[i]; }
    *n_inout = n;
    return NWW_OK;
}

extern "C" float nww_feature_clamp(const nww_handle* h) {
    if (!h || !h->finalized) return 0.0f;
    // set at plan time by the steps that clamp the head input themselves (fused trunk, DNN layer1, fused recurrent input projection,
    // BcResNet front), or 16-bit activation storage; per-row-scaled consumers (lin_x3, ffn_x3, attn_x3) clamp nothing
    return (h->clamps_features || h->cfg.act_dtype != 0) ? NWW_F16_FEATURE_BOUND : 0.0f;
}

extern "C" int nww_describe_plan(const nww_handle* h, char* buf, int32_t buflen) {
    if (!h || !buf || buflen <= 0) return NWW_ERR_INVALID;
    std::string s = "frontend:fe_stft_mel_db_kernel\n";
    for (const auto& st : h->plan) s += st.name + "\n";
    s += "unary:sigmoid\n";
    std::snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return NWW_OK;
}

// ------------------------------------------------------------------------------------------ workspace / run
int nww_ensure_ws(nww_handle* h, int B, int N) {
    if (B <= h->cap_B && N <= h->cap_N) return NWW_OK;
    const int nB = B > h->cap_B ? B : h->cap_B, nN = N > h->cap_N ? N : h->cap_N;
    HIP_TRY(h, hipDeviceSynchronize());
    free_ws(h);
    const nww_config& c = h->cfg;
    size_t per = 0;
    for (int i = 0; i < 6; ++i) per += (h->buf_per_clip[i] + 3) & ~(size_t)3;
    HIP_TRY(h, hipMalloc(&h->d_ws, (per * (size_t)((nB + 127) / 128 * 128) + 4) * sizeof(float)));
    const int T = nN > 0 ? fe_num_frames(h->fe, nN) : 0;
    if (nN > 0) {
        HIP_TRY(h, hipMalloc(&h->d_pcm, (size_t)nB * nN * sizeof(int16_t) + 16));
        if (T > 0) HIP_TRY(h, hipMalloc(&h->d_logmel, (size_t)nB * T * c.n_mels * sizeof(float) + 16));
    }
    HIP_TRY(h, hipMalloc(&h->d_feats, (size_t)nB * c.in_rows * c.in_cols * sizeof(float) + 16));
    HIP_TRY(h, hipMalloc(&h->d_emb, (size_t)nB * c.embedding_dim * sizeof(float) + 16));
    HIP_TRY(h, hipMalloc(&h->d_hid, (size_t)nB * (c.embedding_dim / 2) * sizeof(float) + 16));
    HIP_TRY(h, hipMalloc(&h->d_logits, (size_t)nB * sizeof(float) + 16));
    HIP_TRY(h, hipMalloc(&h->d_probs, (size_t)nB * sizeof(float) + 16));
    if (h->splitk_per_clip) HIP_TRY(h, hipMalloc(&h->d_splitk, h->splitk_per_clip * (size_t)nB * sizeof(float) + 16));
    h->cap_B = nB; h->cap_N = nN; h->cap_rows = (nB + 127) / 128 * 128;
    return NWW_OK;
}

extern "C" int nww_reserve(nww_handle* h, int32_t B, int32_t N) {
    if (!h || B <= 0 || N < 0) return fail(h, NWW_ERR_INVALID, "nww_reserve: bad arguments");
    if (!h->finalized) return fail(h, NWW_ERR_STATE, "nww_reserve before nww_finalize");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return ensure_ws(h, B, N);
}

int nww_run_head(nww_handle* h, const float* d_x, int B, float* d_logits, float* d_probs, hipStream_t s, unsigned int* done_flag,
                 unsigned int done_seq, bool* done_armed, bool x_frames_major) {
    Run r;
    r.x_frames_major = x_frames_major;
    r.done_flag = done_flag; r.done_seq = done_seq;
    r.B = B; r.stream = s; r.x = d_x; r.emb = h->d_emb; r.hid = h->d_hid; r.logits = d_logits ? d_logits : h->d_logits;
    r.probs = d_probs;
    r.splitk_ws = h->d_splitk; r.splitk_floats = h->splitk_per_clip * (size_t)h->cap_B; r.cu_count = h->cu_count;
    if (h->sr.on) {                                    // a streaming hop (nww_stream.hip) configured this forward
        h->sr.on = false;
        r.x_stride = h->sr.x_stride;
        r.stream_mode = h->sr.mode;
        if (r.stream_mode) {
            const int W2 = h->stream_W / 4;
            r.a2_ring = h->d_a2_ring; r.a2_rows = h->a2_rows; r.a2_row0 = h->a2_pos;
            r.a2_ch_stride = (size_t)h->a2_rows * W2; r.a2_clip_stride = 32 * r.a2_ch_stride;
            if (h->d_seq[0]) {
                r.seq_new = h->d_seq[h->seq_cur];
                if (r.stream_mode == 2) { r.seq_prev = h->d_seq[h->seq_cur ^ 1]; r.a3_lo = h->a3_lo; r.a3_hi = h->a3_hi; r.a3_shift = h->a3_shift; }
            }
            if (r.stream_mode == 2) {
                r.a2_nsub = h->a2_nsub;
                for (int q = 0; q < h->a2_nsub; ++q) { r.a2_sub_a[q] = h->a2_sub_a[q]; r.a2_sub_b[q] = h->a2_sub_b[q]; }
            }
        }
    }
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
        r.buf[i] = h->d_ws + off;
        off += ((h->buf_per_clip[i] + 3) & ~(size_t)3) * (size_t)h->cap_rows;
    }
    int id = 1;
    for (auto& st : h->plan) {
        prof_mark(h, s, id++);
        hipError_t e = st.fn(r);
        if (e != hipSuccess) return fail(h, NWW_ERR_HIP, "launch '%s' failed: %s", st.name.c_str(), hipGetErrorString(e));
    }
    if (d_probs && r.need_sigmoid) {
        prof_mark(h, s, id);
        hipError_t e = launch_unary(r.logits, d_probs, (size_t)B, ACT_SIGMOID, s);
        if (e != hipSuccess) return fail(h, NWW_ERR_HIP, "launch 'sigmoid' failed: %s", hipGetErrorString(e));
    }
    prof_mark(h, s, -1);
    if (done_armed) *done_armed = r.done_armed;
    return NWW_OK;
}

int nww_check_run(nww_handle* h, int B) {
    if (!h) return NWW_ERR_INVALID;
    if (!h->finalized) return fail(h, NWW_ERR_STATE, "model not finalized: call nww_finalize after loading the state_dict");
    if (B <= 0) return fail(h, NWW_ERR_INVALID, "batch must be positive (got %d)", B);
    return NWW_OK;
}

int nww_frontend_on_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_db, float* d_mel, int frames_major,
                        hipStream_t s, int* frames_out, size_t row_stride, const Fe2Sub* sub) {
    const int T = fe_num_frames(h->fe, N);
    if (T <= 0) return fail(h, NWW_ERR_INVALID, "clip of %d samples is too short for n_fft=%d (center=%d)", N, h->fe.n_fft, h->fe.center);
    if (frames_out) *frames_out = T;
    static const int mel_env = 2;    // 2: register filters, 1: MFMA tiles, 0: sparse LDS loop
    hipError_t e = fe2_launch(d_pcm, row_stride ? row_stride : (size_t)N, B, N, T, h->fe, h->d_tables, h->d_melplan, d_db, d_mel, frames_major, mel_env, h->mel_max_taps, 256, h->cu_count * 3, s, sub);
    if (e != hipSuccess) return fail(h, NWW_ERR_HIP, "frontend launch failed: %s", hipGetErrorString(e));
    return NWW_OK;
}

extern "C" int nww_frontend_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N, float* d_logmel,
                                int32_t frames_major, void* stream) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!d_pcm || !d_logmel) return fail(h, NWW_ERR_INVALID, "null device pointer");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return frontend_dev(h, d_pcm, B, N, d_logmel, nullptr, frames_major, stream ? (hipStream_t)stream : h->own_stream, nullptr);
}

int nww_forward_pcm_on_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_logits, float* d_probs, hipStream_t s,
                           size_t row_stride, unsigned int* done_flag, unsigned int done_seq, bool* done_armed) {
    const nww_config& c = h->cfg;
    const int T = fe_num_frames(h->fe, N);
    if (T <= 0) return fail(h, NWW_ERR_INVALID, "clip of %d samples is too short for n_fft=%d (center=%d)", N, h->fe.n_fft, h->fe.center);
    const int rows = c.mel_major_features ? c.n_mels : T, cols = c.mel_major_features ? T : c.n_mels;
    if (rows != c.in_rows || cols != c.in_cols)
        return fail(h, NWW_ERR_SHAPE, "frontend yields (%d,%d) features for %d samples but the head was built for input_shape=(%d,%d)",
                    rows, cols, N, c.in_rows, c.in_cols);
    int rc = ensure_ws(h, B, N);
    if (rc) return rc;
    prof_begin(h);
    prof_mark(h, s, 0);
    const bool fm = !c.mel_major_features || h->e2e_transposed;
    rc = frontend_dev(h, d_pcm, B, N, h->d_logmel, nullptr, fm ? 1 : 0, s, nullptr, row_stride);
    if (rc) return rc;
    return run_head(h, h->d_logmel, B, d_logits, d_probs, s, done_flag, done_seq, done_armed, fm && c.mel_major_features);
}

extern "C" int nww_forward_pcm_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N, float* d_logits,
                                   float* d_probs, void* stream) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!d_pcm) return fail(h, NWW_ERR_INVALID, "null device pointer");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return forward_pcm_dev(h, d_pcm, B, N, d_logits, d_probs, stream ? (hipStream_t)stream : h->own_stream);
}

extern "C" int nww_forward_features_dev(nww_handle* h, const float* d_feats, int32_t B, float* d_logits, float* d_probs,
                                        void* stream) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!d_feats) return fail(h, NWW_ERR_INVALID, "null device pointer");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    rc = ensure_ws(h, B, 0);
    if (rc) return rc;
    prof_begin(h);
    return run_head(h, d_feats, B, d_logits, d_probs, stream ? (hipStream_t)stream : h->own_stream);
}

// ---- host-pointer entry points
extern "C" int nww_frontend_ex(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N, float* logmel_out,
                               float* melpower_out, int32_t* frames_out) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!pcm) return fail(h, NWW_ERR_INVALID, "Input audio must be a non-null int16 array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int T = fe_num_frames(h->fe, N);
    if (T <= 0) return fail(h, NWW_ERR_INVALID, "clip of %d samples is too short for n_fft=%d (center=%d)", N, h->fe.n_fft, h->fe.center);
    rc = ensure_ws(h, B, N);
    if (rc) return rc;
    hipStream_t s = h->own_stream;
    const size_t n_out = (size_t)B * T * h->cfg.n_mels;
    struct DevScratch {            // frees the optional mel-power scratch on every exit path
        float* p = nullptr;
        ~DevScratch() { if (p) (void)hipFree(p); }
    } mel;
    if (melpower_out) HIP_TRY(h, hipMalloc(&mel.p, n_out * sizeof(float)));
    HIP_TRY(h, hipMemcpyAsync(h->d_pcm, pcm, (size_t)B * N * sizeof(int16_t), hipMemcpyHostToDevice, s));
    rc = frontend_dev(h, h->d_pcm, B, N, h->d_logmel, mel.p, 0, s, frames_out);
    if (rc) return rc;
    if (logmel_out) HIP_TRY(h, hipMemcpyAsync(logmel_out, h->d_logmel, n_out * sizeof(float), hipMemcpyDeviceToHost, s));
    if (melpower_out) HIP_TRY(h, hipMemcpyAsync(melpower_out, mel.p, n_out * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return NWW_OK;
}

extern "C" int nww_frontend(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N, float* logmel_out, int32_t* frames_out) {
    return nww_frontend_ex(h, pcm, B, N, logmel_out, nullptr, frames_out);
}

// Small host-pointer calls (the interpreter's B = 1 .. 16 predict() path) go through pinned staging buffers: a
// hipMemcpyAsync from pageable memory costs ~40-60 us of runtime staging and synchronisation per call, a memcpy into a
// pinned buffer plus a true async copy ~10.
constexpr size_t PIN_BYTES = 1 << 20;
int nww_h2d_small(nww_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes <= PIN_BYTES) {
        if (!h->pin_in) HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), PIN_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
        // host-pointer entry points synchronise before returning, except on their error paths: wait for a copy a failed call left behind
        if (h->pin_in_busy) { HIP_TRY(h, hipStreamSynchronize(s)); h->pin_in_busy = false; }
        std::memcpy(h->pin_in, src, bytes);
        h->pin_in_busy = true;
        HIP_TRY(h, hipMemcpyAsync(dst, h->pin_in, bytes, hipMemcpyHostToDevice, s));
        return NWW_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return NWW_OK;
}

// Zero-copy staging for the interpreter's small calls: the kernels read the input straight out of the pinned staging buffer
// (device-visible host memory) and the classifier tail writes logits / probabilities straight into the pinned output buffer,
// so a call is memcpy -> kernels -> stream sync -> memcpy: no hipMemcpyAsync command in either direction (each costs ~8 us
// of a B = 1 call).  Returns false when the buffers do not fit the staging area.
static bool zero_copy_ptrs(nww_handle* h, size_t in_bytes, int B, void** d_in, float** d_logits, float** d_probs) {
    if (in_bytes > PIN_BYTES || (size_t)2 * B * sizeof(float) + 16 > PIN_BYTES) return false;
    // fine-grained coherent host memory whatever HIP_HOST_COHERENT says: the host polls a word the running kernel writes
    if (!h->pin_in && hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), PIN_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
    if (!h->pin_out) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), PIN_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
        std::memset(h->pin_out + PIN_BYTES - 16, 0, 16);          // the completion word starts at 0; sequence numbers start at 1
    }
    void *di = nullptr, *dout = nullptr;
    if (hipHostGetDevicePointer(&di, h->pin_in, 0) != hipSuccess || hipHostGetDevicePointer(&dout, h->pin_out, 0) != hipSuccess) return false;
    *d_in = di;
    *d_logits = static_cast<float*>(dout);
    *d_probs = static_cast<float*>(dout) + B;
    return true;
}

// Wait for a zero-copy call: poll the completion word the classifier tail writes into the pinned output buffer (a few us
// sooner than the runtime's stream synchronisation returns), falling back to the stream sync when no kernel was armed to
// write it or it does not show up within ~2 ms.
static int zero_copy_wait(nww_handle* h, hipStream_t s, bool armed, volatile unsigned int* flag, unsigned int seq) {
    if (armed) {
        for (int spin = 0; spin < 2000000; ++spin) {
            if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); h->pin_in_busy = false; return NWW_OK; }
        }
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    h->pin_in_busy = false;
    return NWW_OK;
}

int nww_copy_out(nww_handle* h, int B, float* logits, float* probs, float* emb, hipStream_t s) {
    const size_t nb = (size_t)B * sizeof(float), ne = (size_t)B * h->cfg.embedding_dim * sizeof(float);
    const size_t need = (logits ? nb : 0) + (probs ? nb : 0) + (emb ? ne : 0);
    if (need <= PIN_BYTES) {
        if (!h->pin_out) {
            HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), PIN_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(h->pin_out + PIN_BYTES - 16, 0, 16);
        }
        size_t off = 0;
        unsigned char *pl = nullptr, *pp = nullptr, *pe = nullptr;
        if (logits) { pl = h->pin_out + off; off += nb; HIP_TRY(h, hipMemcpyAsync(pl, h->d_logits, nb, hipMemcpyDeviceToHost, s)); }
        if (probs) { pp = h->pin_out + off; off += nb; HIP_TRY(h, hipMemcpyAsync(pp, h->d_probs, nb, hipMemcpyDeviceToHost, s)); }
        if (emb) { pe = h->pin_out + off; off += ne; HIP_TRY(h, hipMemcpyAsync(pe, h->d_emb, ne, hipMemcpyDeviceToHost, s)); }
        HIP_TRY(h, hipStreamSynchronize(s));
        h->pin_in_busy = false;
        if (logits) std::memcpy(logits, pl, nb);
        if (probs) std::memcpy(probs, pp, nb);
        if (emb) std::memcpy(emb, pe, ne);
        return NWW_OK;
    }
    if (logits) HIP_TRY(h, hipMemcpyAsync(logits, h->d_logits, nb, hipMemcpyDeviceToHost, s));
    if (probs) HIP_TRY(h, hipMemcpyAsync(probs, h->d_probs, nb, hipMemcpyDeviceToHost, s));
    if (emb) HIP_TRY(h, hipMemcpyAsync(emb, h->d_emb, ne, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    h->pin_in_busy = false;
    return NWW_OK;
}

extern "C" int nww_forward_pcm(nww_handle* h, const int16_t* pcm, int32_t B, int32_t N, float* logits, float* probs) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!pcm) return fail(h, NWW_ERR_INVALID, "Input audio must be a non-null int16 array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    rc = ensure_ws(h, B, N);
    if (rc) return rc;
    hipStream_t s = h->own_stream;
    {
        void* d_in = nullptr; float *zl = nullptr, *zp = nullptr;
        const size_t bytes = (size_t)B * N * sizeof(int16_t);
        if (zero_copy_ptrs(h, bytes, B, &d_in, &zl, &zp)) {
            if (h->pin_in_busy) { HIP_TRY(h, hipStreamSynchronize(s)); h->pin_in_busy = false; }
            std::memcpy(h->pin_in, pcm, bytes);
            h->pin_in_busy = true;
            unsigned int* flag = reinterpret_cast<unsigned int*>(zl) + (PIN_BYTES / sizeof(float) - 4);     // last words of the pinned output buffer
            if (++h->done_seq == 0) ++h->done_seq;                   // 0 is the word's initial value, never a sequence number
            const unsigned int seq = h->done_seq;
            bool armed = false;
            rc = forward_pcm_dev(h, static_cast<const int16_t*>(d_in), B, N, zl, zp, s, 0, flag, seq, &armed);
            if (rc) return rc;
            rc = zero_copy_wait(h, s, armed, reinterpret_cast<volatile unsigned int*>(h->pin_out + PIN_BYTES - 16), seq);
            if (rc) return rc;
            if (logits) std::memcpy(logits, h->pin_out, (size_t)B * sizeof(float));
            if (probs) std::memcpy(probs, h->pin_out + (size_t)B * sizeof(float), (size_t)B * sizeof(float));
            return NWW_OK;
        }
    }
    rc = h2d_small(h, h->d_pcm, pcm, (size_t)B * N * sizeof(int16_t), s);
    if (rc) return rc;
    rc = forward_pcm_dev(h, h->d_pcm, B, N, h->d_logits, h->d_probs, s);
    if (rc) return rc;
    return copy_out(h, B, logits, probs, nullptr, s);
}

extern "C" int nww_forward_features_ex(nww_handle* h, const float* feats, int32_t B, float* logits, float* probs, float* emb) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!feats) return fail(h, NWW_ERR_INVALID, "null feature pointer");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    rc = ensure_ws(h, B, 0);
    if (rc) return rc;
    hipStream_t s = h->own_stream;
    if (!emb) {
        void* d_in = nullptr; float *zl = nullptr, *zp = nullptr;
        const size_t bytes = (size_t)B * h->cfg.in_rows * h->cfg.in_cols * sizeof(float);
        if (zero_copy_ptrs(h, bytes, B, &d_in, &zl, &zp)) {
            if (h->pin_in_busy) { HIP_TRY(h, hipStreamSynchronize(s)); h->pin_in_busy = false; }
            std::memcpy(h->pin_in, feats, bytes);
            h->pin_in_busy = true;
            prof_begin(h);
            unsigned int* flag = reinterpret_cast<unsigned int*>(zl) + (PIN_BYTES / sizeof(float) - 4);
            if (++h->done_seq == 0) ++h->done_seq;
            const unsigned int seq = h->done_seq;
            bool armed = false;
            rc = run_head(h, static_cast<const float*>(d_in), B, zl, zp, s, flag, seq, &armed);
            if (rc) return rc;
            rc = zero_copy_wait(h, s, armed, reinterpret_cast<volatile unsigned int*>(h->pin_out + PIN_BYTES - 16), seq);
            if (rc) return rc;
            if (logits) std::memcpy(logits, h->pin_out, (size_t)B * sizeof(float));
            if (probs) std::memcpy(probs, h->pin_out + (size_t)B * sizeof(float), (size_t)B * sizeof(float));
            return NWW_OK;
        }
    }
    rc = h2d_small(h, h->d_feats, feats, (size_t)B * h->cfg.in_rows * h->cfg.in_cols * sizeof(float), s);
    if (rc) return rc;
    prof_begin(h);
    rc = run_head(h, h->d_feats, B, h->d_logits, h->d_probs, s);
    if (rc) return rc;
    return copy_out(h, B, logits, probs, emb, s);
}

extern "C" int nww_forward_features(nww_handle* h, const float* feats, int32_t B, float* logits, float* probs) {
    return nww_forward_features_ex(h, feats, B, logits, probs, nullptr);
}

