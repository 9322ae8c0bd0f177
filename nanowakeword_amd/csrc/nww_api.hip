// nww_api.hip - the C-ABI of include/nww.h: handle, weights, plans (sequence of kernel launches per head).
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
#include "dual_x3.h"
#include "emb_stream.h"
#include <dlfcn.h>

namespace {

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

}  // namespace

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
    EmbState* emb = nullptr;       // embedding-mode preprocessor state (nww_emb_*)
    void* comm = nullptr;          // ncclComm_t of this rank (nww_comm_init)
    unsigned char* pin_in = nullptr; unsigned char* pin_out = nullptr;   // pinned staging for small host-pointer calls
    bool pin_in_busy = false;                                            // an async copy out of pin_in may still be in flight
    unsigned int done_seq = 0;                                           // completion-word sequence of the zero-copy small calls
    int comm_rank = 0, comm_world = 1;
    int ring_S = 0, ring_W = 0, ring_hop = 0, ring_pos = 0; long long ring_filled = 0;
    size_t splitk_per_clip = 0;    // floats per clip (max over the plan's split GEMMs)
    std::map<const float*, void*> x3_weights;      // GEMM weights pre-split into bf16 terms (gemm_x3.hip)
    std::vector<void*> packed_weights;             // other plan-time weight packings (ffn_x3.hip)
    int conv_products = 0;                         // fused trunk: 0 = float32 MFMA, 6 | 9 = bf16 split products
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

static hipEvent_t prof_event(nww_handle* h) {
    hipEvent_t e = nullptr;
    if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static void prof_mark(nww_handle* h, hipStream_t s, int id_of_next) {
    if (!h->profiling || !h->prof_active) return;
    hipEvent_t e = prof_event(h);
    if (!e) return;
    (void)hipEventRecord(e, s);
    h->prof_runs.back().push_back(e);
    h->prof_ids.back().push_back(id_of_next);         // interval that STARTS at this event (-1 = end)
}
static void prof_begin(nww_handle* h) {
    if (!h->profiling) return;
    h->prof_active = (h->prof_counter++ % h->prof_period) == 0;
    if (!h->prof_active) return;
    h->prof_runs.emplace_back();
    h->prof_ids.emplace_back();
}

static thread_local std::string g_create_err;   // nww_create errors: per thread, so concurrent creates do not race

static int fail(nww_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(h, NWW_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------ spec
namespace {
using Shape = std::vector<int64_t>;
struct SpecBuilder {
    std::vector<std::string>& keys;
    std::map<std::string, HostTensor>& t;
    void add(const std::string& k, Shape s) { keys.push_back(k); t[k].shape = std::move(s); }
    void lin(const std::string& p, int out, int in) { add(p + ".weight", {out, in}); add(p + ".bias", {out}); }
    void ln(const std::string& p, int d) { add(p + ".weight", {d}); add(p + ".bias", {d}); }
    void bn(const std::string& p, int c) {
        add(p + ".weight", {c}); add(p + ".bias", {c}); add(p + ".running_mean", {c}); add(p + ".running_var", {c});
    }
    void gru(const std::string& p, int in, int H, int layers, int G = 3) {   // G = 3: nn.GRU, 4: nn.LSTM
        for (int l = 0; l < layers; ++l) {
            const int isz = l == 0 ? in : 2 * H;
            for (const char* sfx : {"", "_reverse"}) {
                const std::string s = "_l" + std::to_string(l) + sfx;
                add(p + ".weight_ih" + s, {G * H, isz}); add(p + ".weight_hh" + s, {G * H, H});
                add(p + ".bias_ih" + s, {G * H}); add(p + ".bias_hh" + s, {G * H});
            }
        }
    }
};

void crnn_out(const nww_config& c, int* C, int* H, int* W) {
    int h = c.in_rows, w = c.in_cols;
    for (int i = 0; i < c.n_crnn_channels; ++i) { h /= 2; w /= 2; }
    *C = c.crnn_channels[c.n_crnn_channels - 1]; *H = h; *W = w;
}

// Mirrors nanowakeword_amd/config.py:param_spec == Model.state_dict() of the reference (model.py:67-296).
void build_spec(nww_handle* h) {
    const nww_config& c = h->cfg;
    SpecBuilder s{h->keys, h->tensors};
    const int T = c.in_rows, F = c.in_cols, L = c.layer_dim, E = c.embedding_dim, nb = c.n_blocks;
    switch (c.head_type) {
        case NWW_HEAD_DNN:
            s.lin("model.layer1", L, T * F); s.ln("model.layernorm1", L);
            for (int i = 0; i < nb; ++i) {
                const std::string p = "model.blocks." + std::to_string(i);
                s.lin(p + ".fcn_layer", L, L); s.ln(p + ".layer_norm", L);
            }
            s.lin("model.last_layer", E, L);
            break;
        case NWW_HEAD_CNN:
            s.add("model.conv1.weight", {16, 1, 3, 3}); s.add("model.conv1.bias", {16});
            s.add("model.conv2.weight", {32, 16, 3, 3}); s.add("model.conv2.bias", {32});
            s.lin("model.fc1", 128, 32 * (T / 4) * (F / 4)); s.lin("model.fc2", E, 128);
            break;
        case NWW_HEAD_CRNN: {
            int cin = 1;
            for (int i = 0; i < c.n_crnn_channels; ++i) {
                const int co = c.crnn_channels[i];
                const std::string p = "model.cnn." + std::to_string(4 * i);
                s.add(p + ".weight", {co, cin, 3, 3}); s.add(p + ".bias", {co});
                s.bn("model.cnn." + std::to_string(4 * i + 1), co);
                cin = co;
            }
            int C, H, W; crnn_out(c, &C, &H, &W);
            s.gru("model.rnn", C * H, L, nb, c.crnn_rnn_lstm ? 4 : 3); s.lin("model.fc", E, 2 * L);
            break;
        }
        case NWW_HEAD_GRU:
            s.gru("model.gru", F, L, nb); s.lin("model.fc", E, 2 * L);
            break;
        case NWW_HEAD_BCRESNET: {
            s.add("model.init_conv.0.weight", {32, 1, 3, 3}); s.bn("model.init_conv.1", 32);
            const int ch[4] = {32, 64, 128, 256};
            for (int i = 1; i <= 3; ++i) {
                const std::string p = "model.block" + std::to_string(i);
                s.add(p + ".depthwise.weight", {ch[i - 1], 1, 3, 3});
                s.add(p + ".pointwise.weight", {ch[i], ch[i - 1], 1, 1}); s.bn(p + ".bn1", ch[i]);
                s.add(p + ".shortcut.0.weight", {ch[i], ch[i - 1], 1, 1}); s.bn(p + ".shortcut.1", ch[i]);
            }
            s.lin("model.fc", E, 256);
            break;
        }
        case NWW_HEAD_CONFORMER: {
            const int D = c.conformer_d_model;
            s.lin("model.input_proj", D, F);
            for (int i = 0; i < nb; ++i) {
                const std::string p = "model.conformer_blocks." + std::to_string(i);
                for (const char* ff : {".ff1", ".ff2"}) {
                    s.ln(p + ff + ".layer_norm", D); s.lin(p + ff + ".linear1", 4 * D, D); s.lin(p + ff + ".linear2", D, 4 * D);
                }
                s.add(p + ".attention.in_proj_weight", {3 * D, D}); s.add(p + ".attention.in_proj_bias", {3 * D});
                s.lin(p + ".attention.out_proj", D, D);
                s.ln(p + ".conv_module.layer_norm", D);
                s.add(p + ".conv_module.conv1.weight", {2 * D, D, 1}); s.add(p + ".conv_module.conv1.bias", {2 * D});
                s.add(p + ".conv_module.depthwise_conv.weight", {D, 1, 31}); s.add(p + ".conv_module.depthwise_conv.bias", {D});
                s.bn(p + ".conv_module.batch_norm", D);
                s.add(p + ".conv_module.conv2.weight", {D, D, 1}); s.add(p + ".conv_module.conv2.bias", {D});
                s.ln(p + ".layer_norm", D);
            }
            s.lin("model.output_proj", E, D);
            break;
        }
        case NWW_HEAD_E2E_DNN: {
            int cin = 1;
            const int ch[3] = {16, 32, 64};
            for (int i = 0; i < 3; ++i) {
                const std::string p = "model.conv_block." + std::to_string(4 * i);
                s.add(p + ".weight", {ch[i], cin, 3, 3}); s.add(p + ".bias", {ch[i]});
                s.bn("model.conv_block." + std::to_string(4 * i + 1), ch[i]);
                cin = ch[i];
            }
            s.lin("model.fc1", 128, 256); s.bn("model.bn1", 128); s.lin("model.out", E, 128);
            break;
        }
    }
    s.lin("classifier.0", E / 2, E);
    s.lin("classifier.3", 1, E / 2);
}

size_t numel(const Shape& s) { size_t n = 1; for (auto v : s) n *= (size_t)v; return n; }

}  // namespace

// ------------------------------------------------------------------------------------------ create / load
// bf16x6 is float32-grade (tools/x3_accuracy.py: max |dlogit| vs float64 5.2e-6, the float32 MFMA path 5.0e-6) and 1.7x faster
#define NWW_DEFAULT_CONV_ARITH NWW_ARITH_BF16X6
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
    if (c.conv_arith != NWW_ARITH_DEFAULT && c.conv_arith != NWW_ARITH_F32 && c.conv_arith != NWW_ARITH_BF16X6 && c.conv_arith != NWW_ARITH_BF16X9)
        return fail(nullptr, NWW_ERR_INVALID, "bad conv_arith code %d", c.conv_arith);
    if (c.in_rows <= 0 || c.in_cols <= 0 || c.embedding_dim < 2 || c.layer_dim <= 0 || c.n_blocks < 0)
        return fail(nullptr, NWW_ERR_INVALID, "bad head dimensions");
    if (c.n_fft != 400) return fail(nullptr, NWW_ERR_UNSUPPORTED, "only n_fft=400 is implemented (got %d)", c.n_fft);
    if (c.win_length <= 0 || c.win_length > c.n_fft) return fail(nullptr, NWW_ERR_INVALID, "win_length must be in 1..n_fft");
    if (c.hop_length <= 0 || (c.hop_length & 1)) return fail(nullptr, NWW_ERR_UNSUPPORTED, "hop_length must be positive and even");
    if (c.n_mels <= 0 || c.n_mels > FE_MAX_MELS) return fail(nullptr, NWW_ERR_INVALID, "n_mels must be in 1..%d", FE_MAX_MELS);
    if (c.head_type == NWW_HEAD_CRNN && (c.n_crnn_channels < 1 || c.n_crnn_channels > 4))
        return fail(nullptr, NWW_ERR_INVALID, "crnn_cnn_channels must have 1..4 stages");
    if ((c.head_type == NWW_HEAD_CRNN || c.head_type == NWW_HEAD_GRU) && (c.layer_dim % 4 != 0 || c.layer_dim > 256))
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "recurrent hidden size (layer_dim = %d) must be a multiple of 4 and <= 256", c.layer_dim);
    if (c.head_type == NWW_HEAD_CONFORMER && (c.conformer_n_head <= 0 || c.conformer_d_model % c.conformer_n_head))
        return fail(nullptr, NWW_ERR_INVALID, "conformer_d_model must be divisible by conformer_n_head");
    if (c.act_dtype != NWW_ACT_DTYPE_F32 && c.act_dtype != NWW_ACT_DTYPE_BF16) return fail(nullptr, NWW_ERR_INVALID, "act_dtype must be NWW_ACT_DTYPE_F32 or NWW_ACT_DTYPE_BF16");
    if (c.act_dtype == NWW_ACT_DTYPE_BF16 && c.head_type != NWW_HEAD_BCRESNET)
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "act_dtype = bf16 is implemented for the BcResNet head only (BASELINE config 3)");
    if (c.head_type == NWW_HEAD_CONFORMER && !mha_head_dim_supported(c.conformer_d_model / c.conformer_n_head))
        return fail(nullptr, NWW_ERR_UNSUPPORTED, "attention head_dim %d has no compiled kernel (multiples of 4 up to 72, or 18)",
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
        h->conv_products = mode == NWW_ARITH_BF16X6 ? 6 : mode == NWW_ARITH_BF16X9 ? 9 : 0;
    }
    h->fe.sample_rate = c.sample_rate; h->fe.n_fft = c.n_fft; h->fe.win_length = c.win_length; h->fe.hop = c.hop_length;
    h->fe.n_mels = c.n_mels; h->fe.center = c.center; h->fe.f_min = c.f_min; h->fe.f_max = c.f_max;
    h->fe.amin = c.amin; h->fe.db_mult = c.db_multiplier;
    build_spec(h);
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

// ------------------------------------------------------------------------------------------ plan helpers
namespace {

struct PlanCtx {
    nww_handle* h;
    const float* W(const std::string& k) const {
        auto it = h->tensors.find(k);
        return it == h->tensors.end() || !it->second.loaded ? nullptr : h->d_weights + it->second.dev_off;
    }
    void need(int buf, size_t floats_per_clip) {
        if (h->buf_per_clip[buf] < floats_per_clip) h->buf_per_clip[buf] = floats_per_clip;
    }
    void add(const std::string& name, std::function<hipError_t(Run&)> fn) { h->plan.push_back({name, std::move(fn)}); }
    void pop_last() { if (!h->plan.empty()) h->plan.pop_back(); }        // a step just planned is re-planned in another form
    // the head's last Linear (-> embedding), deferred so that it can be fused with the classifier into one launch
    std::string tail_name; int tail_in = 99, tail_K = 0; const float *tail_W = nullptr, *tail_b = nullptr;
};

// source selector for a step input: -1 = head input x, -2 = emb, -3 = hid, >=0 workspace buffer
inline const float* src(Run& r, int id) { return id == -1 ? r.x : id == -2 ? r.emb : id == -3 ? r.hid : r.buf[id]; }
inline float* dst(Run& r, int id) { return id == -2 ? r.emb : id == -3 ? r.hid : id == -4 ? r.logits : r.buf[id]; }

// rows_per_clip: M = B*rows_per_clip
void add_gemm(PlanCtx& p, const std::string& name, int in_id, int out_id, int rows_per_clip, int N, int K,
              const float* W, const float* bias, int act, const float* alpha = nullptr, const float* beta = nullptr,
              int res_id = 99, float rscale = 1.f, bool* a_blocked_inout = nullptr, bool feeds_tail = false, bool feeds_ln = false) {
    if (out_id >= 0) p.need(out_id, (size_t)rows_per_clip * N);
    // Contractions run on the bf16 matrix cores by exact operand splitting (gemm_x3.hip) where that kernel wins -
    // measured per shape on the Conformer / GRU / CNN heads at full batch (ms, split-operand vs float32 MFMA):
    // (N,K) = (576,144) 0.32 / 0.46, (144,576) 0.41 / 0.51, (432,144) 0.26 / 0.33, (288,144) 0.18 / 0.24,
    // (144,64) 0.05 / 0.07, (384,64) 0.23 / 0.29, (128,12800) 0.09 / 0.14 - and stay on the float32 MFMA kernel for the
    // small square ones, (144,144) 0.23 / 0.18, whose single padded column tile wastes the wide kernel.  Long-K layers
    // get a fine split-K.  The choice depends on (N, K) only, so batch invariance is kept.
    // NWW_GEMM_X3 = 0: never, 2: every shape with N, K >= 32, 3: the round-1 rule (K >= 4096 only).
    static const int x3_mode = [] { const char* e = getenv("NWW_GEMM_X3"); return e ? atoi(e) : 1; }();
    const bool small_square = N <= 160 && K > 64 && K <= 160;
    const bool use_x3 = p.h->conv_products != 0 &&
                        (x3_mode == 2 ? (N >= 32 && K >= 32)
                         : x3_mode == 3 ? (K >= 4096 && N >= 64 && N <= 256)
                         : (x3_mode == 1 && N >= 64 && K >= 32 && !small_square));
    const void* wx3 = nullptr;
    if (use_x3) {
        auto it = p.h->x3_weights.find(W);
        if (it == p.h->x3_weights.end()) {
            void* d = nullptr;
            if (hipMalloc(&d, gemm_x3_weight_bytes(N, K)) == hipSuccess &&
                launch_split_weights_x3(W, d, N, K, p.h->own_stream) == hipSuccess)
                it = p.h->x3_weights.emplace(W, d).first;
            else if (d) (void)hipFree(d);
        }
        if (it != p.h->x3_weights.end()) wx3 = it->second;
    }
    // the producer may write A directly as this kernel's [128][32] tiles (the fused trunk feeding fc1)
    const int a_blocked = (a_blocked_inout && *a_blocked_inout && wx3 && K % 32 == 0 && rows_per_clip == 1) ? K / 32 : 0;
    if (a_blocked_inout) *a_blocked_inout = a_blocked != 0;
    if (K >= 2048 && (size_t)16 * rows_per_clip * N > p.h->splitk_per_clip) p.h->splitk_per_clip = (size_t)16 * rows_per_clip * N;
    p.add("gemm:" + name, [=](Run& r) {
        GemmArgs g;
        g.A = src(r, in_id); g.lda = K; g.W = W; g.C = dst(r, out_id); g.ldc = N;
        g.M = r.B * rows_per_clip; g.N = N; g.K = K; g.bias = bias; g.alpha = alpha; g.beta = beta; g.act = act;
        g.res = res_id == 99 ? nullptr : src(r, res_id); g.ldres = N; g.rscale = rscale;
        g.Wx3 = wx3;
        g.a_blocked = a_blocked;
        g.splitk = gemm_recommended_splitk(g.M, N, K, r.cu_count);
        // split-operand layers: chunks of ~16-25 k-tiles, so that a small batch's chunk is ONE round of gemm_x3_chain_kernel
        // (32 k-tiles in flight) on a few dozen CUs - K = 12 800: 16 chunks of 25, K = 6 464: 12 of 17, K = 3 920 (C1): 7 of 18
        if (wx3 && x3_mode != 2 && K >= 2048) { g.splitk = K >= 8192 ? K / 800 : K / 512; if (g.splitk > 16) g.splitk = 16; if (g.splitk < 1) g.splitk = 1; }
        g.splitk_ws = r.splitk_ws;
        if (g.splitk > 1 && (size_t)g.splitk * g.M * N > r.splitk_floats) g.splitk = 1;
        r.deferred.active = false;
        // a handful of clips (the interpreter's calls): the fused tail sums the partials itself, in the same order - one
        // dependent launch less (B = 1: 62 -> 58 us back-to-back).  Larger batches keep the reduce launch: the tail's few
        // workgroups read the 16 partials slower than the full-grid reduce does (B = 4096: 0.028 vs 0.019 + 0.007 ms).
        if (feeds_tail && g.M <= 8 && g.splitk > 1 && g.splitk_ws && !g.res) {
            g.defer_reduce = true;
            r.deferred.active = true; r.deferred.out_id = out_id; r.deferred.parts = g.splitk; r.deferred.stride = (size_t)g.M * N;
            r.deferred.bias = bias; r.deferred.alpha = alpha; r.deferred.beta = beta; r.deferred.act = act;
        }
        // a LayerNorm right behind a split-K Linear (DNN layer1) sums the partials itself, at every batch size: the reduce launch
        // and its round trip go (the caller's LayerNorm step checks r.deferred)
        if (feeds_ln && g.splitk > 1 && g.splitk_ws && !g.res && !alpha && act == ACT_NONE && N <= 256) {
            g.defer_reduce = true;
            r.deferred.active = true; r.deferred.out_id = out_id; r.deferred.parts = g.splitk; r.deferred.stride = (size_t)g.M * N;
            r.deferred.bias = bias; r.deferred.alpha = nullptr; r.deferred.beta = nullptr; r.deferred.act = ACT_NONE;
        }
        return launch_gemm(g, r.stream);
    });
}

// Short-K Linear on the input-stationary split-operand kernel (lin_x3.hip); false -> the caller plans the general GEMM.
// epi 0: out = y + b; 1: out = res + rscale (y + b); 2: LayerNorm(ln_w, ln_b) first when given, W = [2N][K], out = a * sigmoid(b)
bool add_lin_x3(PlanCtx& p, const std::string& name, int in_id, int out_id, int rows_per_clip, int N, int K, const float* W,
                const float* bias, int epi, int res_id = 99, float rscale = 1.f, const float* ln_w = nullptr,
                const float* ln_b = nullptr, int qkv_T = 0, int qkv_dh = 0) {
    static const int enabled = [] { const char* e = getenv("NWW_LIN_X3"); return e ? atoi(e) : 1; }();
    if (!enabled || p.h->conv_products != 6 || !lin_x3_supported(K, N)) return false;
    const int parts = epi == 2 ? 2 : 1;
    void* packed = nullptr;
    if (hipMalloc(&packed, lin_x3_packed_bytes(K, N, parts)) != hipSuccess) return false;
    if (launch_lin_x3_pack(W, bias, packed, K, N, parts, N, p.h->own_stream) != hipSuccess) { (void)hipFree(packed); return false; }
    p.h->packed_weights.push_back(packed);
    p.need(out_id, (size_t)rows_per_clip * N);
    p.add("lin_x3:" + name, [=](Run& r) {
        LinArgs a;
        a.x = src(r, in_id); a.ldx = K; a.out = dst(r, out_id); a.ldc = N;
        a.res = res_id == 99 ? nullptr : src(r, res_id); a.ldres = N; a.rscale = rscale;
        a.ln_w = ln_w; a.ln_b = ln_b; a.packed = static_cast<const unsigned char*>(packed);
        a.M = r.B * rows_per_clip; a.N = N; a.qkv_T = qkv_T; a.qkv_dh = qkv_dh;
        return launch_lin_x3(a, K, epi, ln_w != nullptr, r.stream);
    });
    return true;
}

void set_tail(PlanCtx& p, const std::string& name, int in_id, int K, const float* W, const float* b) {
    p.tail_name = name; p.tail_in = in_id; p.tail_K = K; p.tail_W = W; p.tail_b = b;
}

void add_conv(PlanCtx& p, const std::string& name, int in_id, int out_id, int Cin, int Cout, int H, int W,
              const float* w, const float* bias, const float* alpha, const float* beta, int act, int pool, int nhwc_out = 0) {
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    p.need(out_id, (size_t)Cout * Ho * Wo);
    p.add("conv3x3:" + name, [=](Run& r) {
        Conv3Args a{src(r, in_id), w, bias, alpha, beta, dst(r, out_id), r.B, Cin, Cout, H, W, act, pool};
        a.nhwc_out = nhwc_out;
        return launch_conv3x3(a, r.stream);
    });
}

static int trunk_fits(int C1, int H, int W) { int per_cu = 0; return trunk_pick_strips(C1, H, W, &per_cu); }
// fused conv1+pool+conv2+pool (trunk.hip) when the 1->16->32 pattern fits LDS; returns false if not applicable
bool add_trunk(PlanCtx& p, const std::string& name, int in_id, int out_id, int C1, int C2, int H, int W,
               const float* w1, const float* b1, const float* al1, const float* be1, const float* w2,
               const float* b2, const float* al2, const float* be2, int act, const bool* out_blocked = nullptr) {
    static const int enabled = [] { const char* e = getenv("NWW_TRUNK"); return e ? atoi(e) : 1; }();
    if (!enabled || C1 != 16 || C2 != 32 || H < 4 || W < 4 || trunk_fits(C1, H, W) == 0) return false;
    p.need(out_id, (size_t)C2 * (H / 4) * (W / 4));
    const int max_grid = p.h->cu_count;
    // both convolutions on the bf16 matrix cores by exact operand splitting (trunk_b.hip) or on the float32 MFMA (nww_config.conv_arith)
    const int x3 = p.h->conv_products;
    if ((x3 == 6 || x3 == 9) && trunk_b_pick_strips(H, W) > 0) {
        // both convolutions' weights as the MFMA register images, split into bf16 terms once (trunk_b.hip)
        void* packed = nullptr;
        if (hipMalloc(&packed, trunk_b_packed_bytes()) != hipSuccess) return false;
        if (launch_trunk_b_pack(w1, w2, static_cast<unsigned char*>(packed), p.h->own_stream) != hipSuccess) { (void)hipFree(packed); return false; }
        p.h->packed_weights.push_back(packed);
        p.add("trunk_x3:" + name, [=](Run& r) {
            TrunkArgs a{src(r, in_id), w1, b1, al1, be1, w2, b2, al2, be2, dst(r, out_id), r.B, H, W, act};
            if (out_blocked && *out_blocked) a.out_blocked = C2 * (H / 4) * (W / 4) / 32;     // decided by the consumer (add_gemm) at plan time
            a.wpack = static_cast<const unsigned char*>(packed);
            return launch_cnn_trunk_b(a, x3, max_grid, r.stream);
        });
        return true;
    }
    p.add("trunk:" + name, [=](Run& r) {
        TrunkArgs a{src(r, in_id), w1, b1, al1, be1, w2, b2, al2, be2, dst(r, out_id), r.B, H, W, act};
        return launch_cnn_trunk(a, C1, C2, max_grid, r.stream);
    });
    return true;
}

// 3x3 conv stage with 32 input channels on MFMA (trunk.hip) when it fits; false -> caller uses the VALU kernel
bool add_conv_mfma(PlanCtx& p, const std::string& name, int in_id, int out_id, int Cin, int Cout, int H, int W,
                   const float* w, const float* bias, const float* alpha, const float* beta, int act, int pool,
                   int avg_kw = 0, int avg_sw = 0, int avg_ow = 0, bool* seq_inout = nullptr, int avg_y = 0) {
    static const int enabled = [] { const char* e = getenv("NWW_CONV_MFMA"); return e ? atoi(e) : 1; }();
    if (!enabled || Cin != 32 || Cout % 32 != 0 || (8 % (Cout / 32)) != 0 || H < 2 || W < 2 ||
        conv_mfma_lds_bytes(Cin, H, W) > 160 * 1024)
        return false;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    p.need(out_id, avg_ow > 0 ? (size_t)Cout * avg_ow : (size_t)Cout * Ho * Wo);
    const int max_grid = p.h->cu_count;
    // split-operand bf16 instance (conv3_x3.hip) under the same arithmetic switch as the fused trunk; the 9-product
    // mode keeps the float32-MFMA kernel (the conv3 instance implements the 6-product form only)
    static const int x3_enabled = [] { const char* e = getenv("NWW_CONV3_X3"); return e ? atoi(e) : 1; }();
    if (x3_enabled && p.h->conv_products == 6 && conv3_x3_fits(H, W, Cout, avg_ow, pool)) {
        const size_t lds = conv3_x3_lds_bytes(H, W, avg_ow);
        const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
        const int seq_out = (seq_inout && *seq_inout && pool && avg_ow == 0) ? 1 : 0;      // the caller wants the sequence layout
        p.add(std::string(avg_ow > 0 ? "conv3_x3+avgpool:" : seq_out ? "conv3_x3+seq:" : "conv3_x3:") + name, [=](Run& r) {
            ConvMfmaArgs a{src(r, in_id), w, bias, alpha, beta, dst(r, out_id), r.B, H, W, Cout, act, pool};
            a.avg_kw = avg_kw; a.avg_sw = avg_sw; a.avg_ow = avg_ow; a.seq_out = seq_out; a.avg_y = avg_y;
            return launch_conv3_x3(a, max_grid * per_cu, r.stream);
        });
        return true;
    }
    if (avg_y) return false;                                   // only conv3_x3 pools along y (the caller checked e2e_transposed_ok)
    if (seq_inout) *seq_inout = false;                         // the float32-MFMA instance writes planes
    p.add(std::string(avg_ow > 0 ? "conv3x3_mfma+avgpool:" : "conv3x3_mfma:") + name, [=](Run& r) {
        ConvMfmaArgs a{src(r, in_id), w, bias, alpha, beta, dst(r, out_id), r.B, H, W, Cout, act, pool};
        a.avg_kw = avg_kw; a.avg_sw = avg_sw; a.avg_ow = avg_ow;
        return launch_conv3x3_mfma(a, Cin, max_grid, r.stream);
    });
    return true;
}

// The E2E head can run on the TRANSPOSED plane (frames, n_mels) = (101, 64) instead of (64, 101): the fused trunk's 32-pixel
// conv1 groups and 16-column conv2 tiles waste 28 % on a 101-wide plane and nothing on a 64-wide one, conv3's 2 x 16 tiles 22 %
// against 4 %, and the frontend's frames-major output is its fast path.  Needs the split-operand kernels (default arithmetic).
bool e2e_transposed_ok(PlanCtx& p, int n_mels, int frames) {
    static const int on = [] { const char* e = getenv("NWW_E2E_TRANSPOSED"); return e ? atoi(e) : 1; }();
    static const int trunk_on = [] { const char* e = getenv("NWW_TRUNK"); return e ? atoi(e) : 1; }();
    static const int mfma_on = [] { const char* e = getenv("NWW_CONV_MFMA"); return e ? atoi(e) : 1; }();
    static const int c3_on = [] { const char* e = getenv("NWW_CONV3_X3"); return e ? atoi(e) : 1; }();
    const int H = frames, W = n_mels;
    return on && trunk_on && mfma_on && c3_on && p.h->conv_products == 6 && H >= 16 && W >= 4 && trunk_b_pick_strips(H, W) > 0 &&
           conv_mfma_lds_bytes(32, H / 4, W / 4) <= 160 * 1024 && conv3_x3_fits(H / 4, W / 4, 64, 4, 0);
}

// nn.GRU / nn.LSTM (bidirectional; G = 3 / 4 gates) -> rnn_out[:, -1, :] into buffer `last_id` [B][2H]; uses buffers
// xg_id, seqA, seqB.
void add_bigru_last(PlanCtx& p, const std::string& prefix, int in_id, int T, int I, int H, int layers, int xg_id,
                    int seqA, int seqB, int last_id, int G = 3) {
    p.need(xg_id, (size_t)(T + 1) * G * H);                  // + one row per clip: the reverse direction's last-frame projection
    p.need(last_id, (size_t)2 * H);
    const int products = p.h->conv_products;                 // the recurrent product follows the handle's arithmetic switch
    GruArgs probe; probe.H = H; probe.products = products; probe.w_hh = nullptr;
    const bool x3 = rnn_x3_enabled(probe);                    // (weights come from hipMalloc: 16-byte aligned)
    int cur_in = in_id, cur_I = I;
    for (int l = 0; l < layers; ++l) {
        const bool last = l == layers - 1;
        const int seq_out = (l % 2 == 0) ? seqA : seqB;
        if (!last) p.need(seq_out, (size_t)T * 2 * H);
        // rnn_out[:, -1] needs ONE step of the last layer's reverse direction, hence the input projection of frame T-1 only (a
        // strided GEMM over M = B rows instead of B*T, into the row behind the forward direction's xg) - and with h = 0 that
        // step has no recurrent product: rnn_x3 computes it in the forward direction's launch.
        const bool fold = last && x3;
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = "_l" + std::to_string(l) + (dir ? "_reverse" : "");
            const float* wih = p.W(prefix + ".weight_ih" + sfx);
            const float* whh = p.W(prefix + ".weight_hh" + sfx);
            const float* bih = p.W(prefix + ".bias_ih" + sfx);
            const float* bhh = p.W(prefix + ".bias_hh" + sfx);
            if (last && dir) {
                const int Iin = cur_I, in_buf = cur_in;
                p.add("gemm:" + prefix + ".ih" + sfx + "(last frame)", [=](Run& r) {
                    GemmArgs g;
                    g.A = src(r, in_buf) + (size_t)(T - 1) * Iin; g.lda = T * Iin; g.W = wih;
                    if (fold) { g.C = r.buf[xg_id] + (size_t)r.B * T * G * H; g.ldc = G * H; }      // behind the forward direction's rows
                    else { g.C = r.buf[xg_id] + (size_t)(T - 1) * G * H; g.ldc = T * G * H; }       // in place (the forward recurrence is done)
                    g.M = r.B; g.N = G * H; g.K = Iin; g.bias = bih; g.alpha = nullptr; g.beta = nullptr; g.act = ACT_NONE;
                    g.res = nullptr; g.ldres = 0; g.rscale = 1.f;
                    return launch_gemm(g, r.stream);
                });
            } else {
                // short-K input projections (the GRU head's 64 mel bins) on the input-stationary kernel; the rest on the general GEMM
                if (!add_lin_x3(p, prefix + ".ih" + sfx, cur_in, xg_id, T, G * H, cur_I, wih, bih, 0))
                    add_gemm(p, prefix + ".ih" + sfx, cur_in, xg_id, T, G * H, cur_I, wih, bih, ACT_NONE);
            }
            if (fold && dir == 0) continue;                  // the forward recurrence is launched after the reverse projection
            const int in_T = T;
            const float* whh_f = fold ? p.W(prefix + ".weight_hh_l" + std::to_string(l)) : whh;
            const float* bhh_f = fold ? p.W(prefix + ".bias_hh_l" + std::to_string(l)) : bhh;
            const std::string nm = fold ? (G == 4 ? "lstm:" : "gru:") + prefix + "_l" + std::to_string(l) + " + first reverse step"
                                        : (G == 4 ? "lstm:" : "gru:") + prefix + sfx;
            p.add(nm, [=](Run& r) {
                GruArgs a;
                a.products = products;
                a.xg = r.buf[xg_id]; a.w_hh = whh_f; a.b_hh = bhh_f;
                a.seq_out = last ? nullptr : r.buf[seq_out]; a.ld_seq = 2 * H;
                a.last_out = last ? r.buf[last_id] : nullptr; a.ld_last = 2 * H;
                a.B = r.B; a.T = in_T; a.H = H;
                if (fold) {
                    a.col_off = 0; a.reverse = 0; a.steps = in_T;
                    a.xg2 = r.buf[xg_id] + (size_t)r.B * in_T * G * H; a.xg2_bstride = (size_t)G * H; a.b_hh2 = bhh; a.col_off2 = H;
                } else {
                    a.col_off = dir ? H : 0; a.reverse = dir;
                    a.steps = (last && dir) ? 1 : in_T;      // reverse half of rnn_out[:, -1] is its first step
                }
                return G == 4 ? launch_lstm(a, r.stream) : launch_gru(a, r.stream);
            });
        }
        cur_in = seq_out; cur_I = 2 * H;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------ finalize
extern "C" int nww_finalize(nww_handle* h) {
    if (!h) return NWW_ERR_INVALID;
    if (h->finalized) return NWW_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const nww_config& c = h->cfg;
    for (const auto& k : h->keys)
        if (!h->tensors[k].loaded) return fail(h, NWW_ERR_MISSING, "Missing key(s) in state_dict: '%s'", k.c_str());
    // ---- fold every BatchNorm (eval): alpha = w/sqrt(var+eps), beta = b - mean*alpha (PyTorch CPU kernel form)
    std::vector<std::string> bn_prefixes;
    for (const auto& k : h->keys) {
        const std::string sfx = ".running_var";
        if (k.size() > sfx.size() && k.compare(k.size() - sfx.size(), sfx.size(), sfx) == 0)
            bn_prefixes.push_back(k.substr(0, k.size() - sfx.size()));
    }
    for (const auto& p : bn_prefixes) {
        const HostTensor &w = h->tensors[p + ".weight"], &b = h->tensors[p + ".bias"], &m = h->tensors[p + ".running_mean"],
                         &v = h->tensors[p + ".running_var"];
        HostTensor al, be;
        al.shape = be.shape = w.shape;
        al.data.resize(w.data.size()); be.data.resize(w.data.size());
        for (size_t i = 0; i < w.data.size(); ++i) {
            const float invstd = 1.0f / std::sqrt(v.data[i] + 1e-5f);
            al.data[i] = w.data[i] * invstd;
            be.data[i] = b.data[i] - m.data[i] * al.data[i];
        }
        al.loaded = be.loaded = true;
        h->tensors[p + ".alpha"] = al;
        h->tensors[p + ".beta"] = be;
    }
    // ---- depthwise 3x3 weights tap-major [9][C] for the channels-last kernels (BcResNet)
    if (c.head_type == NWW_HEAD_BCRESNET)
        for (int i = 1; i <= 3; ++i) {
            const std::string k = "model.block" + std::to_string(i) + ".depthwise.weight";
            const HostTensor& w = h->tensors[k];
            const int C = (int)w.shape[0];
            HostTensor wt;
            wt.shape = {9, C};
            wt.data.resize((size_t)9 * C);
            for (int ch = 0; ch < C; ++ch)
                for (int tap = 0; tap < 9; ++tap) wt.data[(size_t)tap * C + ch] = w.data[(size_t)ch * 9 + tap];
            wt.loaded = true;
            h->tensors[k + "_t"] = wt;
        }
    // ---- weight arena (each tensor 16-byte aligned)
    size_t total = 0;
    for (auto& kv : h->tensors) {
        if (!kv.second.loaded || kv.first.rfind("frontend.", 0) == 0) continue;
        kv.second.dev_off = total;
        total += (kv.second.data.size() + 3) & ~(size_t)3;
    }
    HIP_TRY(h, hipMalloc(&h->d_weights, (total + 4) * sizeof(float)));
    for (auto& kv : h->tensors) {
        if (!kv.second.loaded || kv.first.rfind("frontend.", 0) == 0) continue;
        HIP_TRY(h, hipMemcpy(h->d_weights + kv.second.dev_off, kv.second.data.data(), kv.second.data.size() * sizeof(float),
                             hipMemcpyHostToDevice));
    }
    // ---- frontend tables
    {
        std::vector<float> win, fb;
        auto wi = h->tensors.find("frontend.window");
        if (wi != h->tensors.end() && wi->second.loaded) win = wi->second.data; else fe_default_window(h->fe.win_length, win);
        auto fi = h->tensors.find("frontend.mel_fb");
        if (fi != h->tensors.end() && fi->second.loaded) fb = fi->second.data; else fe_default_melfb(h->fe, fb);
        FeTables tb;
        const std::string e = fe_build_tables(h->fe, win.data(), fb.data(), &tb);
        if (!e.empty()) return fail(h, NWW_ERR_INVALID, "frontend tables: %s", e.c_str());
        const size_t tbytes = (sizeof(FeTables) + 15) & ~(size_t)15;
        HIP_TRY(h, hipMalloc(&h->d_tables, tbytes));
        HIP_TRY(h, hipMemset(h->d_tables, 0, tbytes));
        HIP_TRY(h, hipMemcpy(h->d_tables, &tb, sizeof(FeTables), hipMemcpyHostToDevice));
        h->mel_max_taps = 0;
        for (int j = 0; j < h->fe.n_mels; ++j) h->mel_max_taps = tb.mel_cnt[j] > h->mel_max_taps ? tb.mel_cnt[j] : h->mel_max_taps;
        std::vector<Fe2MelPlan> plan(1);
        const std::string e2 = fe2_build_mel_plan(h->fe, fb.data(), plan.data());
        if (!e2.empty()) return fail(h, NWW_ERR_INVALID, "frontend mel plan: %s", e2.c_str());
        HIP_TRY(h, hipMalloc(&h->d_melplan, sizeof(Fe2MelPlan)));
        HIP_TRY(h, hipMemcpy(h->d_melplan, plan.data(), sizeof(Fe2MelPlan), hipMemcpyHostToDevice));
    }
    // ---- plan
    PlanCtx p{h};
    const int T = c.in_rows, F = c.in_cols, L = c.layer_dim, E = c.embedding_dim, nb = c.n_blocks, act = c.activation;
    switch (c.head_type) {
        case NWW_HEAD_DNN: {                      // Net: architectures.py:110-126
            add_gemm(p, "layer1", -1, 0, 1, L, T * F, p.W("model.layer1.weight"), p.W("model.layer1.bias"), ACT_NONE, nullptr, nullptr, 99, 1.f, nullptr, false, true);
            {
                const float *lw1 = p.W("model.layernorm1.weight"), *lb1 = p.W("model.layernorm1.bias");
                p.add("layernorm:layernorm1", [=](Run& r) {
                    if (r.deferred.active && r.deferred.out_id == 0) {           // layer1 left its split-K partials: sum them here
                        r.deferred.active = false;
                        return launch_layernorm_parts(r.splitk_ws, r.deferred.parts, r.deferred.stride, r.deferred.bias, r.buf[0], lw1, lb1, r.B, L, act, r.stream);
                    }
                    return launch_layernorm(r.buf[0], r.buf[0], lw1, lb1, r.B, L, act, r.stream);
                });
            }
            int cur = 0;
            for (int i = 0; i < nb; ++i) {
                const std::string q = "model.blocks." + std::to_string(i);
                const int nxt = cur ^ 1;
                add_gemm(p, q + ".fcn_layer", cur, nxt, 1, L, L, p.W(q + ".fcn_layer.weight"), p.W(q + ".fcn_layer.bias"), ACT_NONE);
                const float *lw = p.W(q + ".layer_norm.weight"), *lb = p.W(q + ".layer_norm.bias");
                p.add("layernorm:" + q, [=](Run& r) { return launch_layernorm(r.buf[nxt], r.buf[nxt], lw, lb, r.B, L, act, r.stream); });
                cur = nxt;
            }
            set_tail(p, "last_layer", cur, L, p.W("model.last_layer.weight"), p.W("model.last_layer.bias"));
            break;
        }
        case NWW_HEAD_CNN: {                      // CNNModel: architectures.py:51-80
            // trunk -> fc1 hand-over as the GEMM's own A tiles when both run on the split-operand path and the geometry allows
            // 16-byte stores inside a 32-feature tile row
            const int H2 = T / 4, W2 = F / 4;
            h->trunk_blocked = (h->conv_products == 6 || h->conv_products == 9) && trunk_b_pick_strips(T, F) > 0 &&
                               (W2 % 4) == 0 && ((H2 * W2) % 4) == 0 && ((32 * H2 * W2) % 32) == 0;
            const bool fused = add_trunk(p, "conv1+pool+conv2+pool", -1, 1, 16, 32, T, F, p.W("model.conv1.weight"), p.W("model.conv1.bias"), nullptr, nullptr,
                                         p.W("model.conv2.weight"), p.W("model.conv2.bias"), nullptr, nullptr, act, &h->trunk_blocked);
            if (!fused) {
                h->trunk_blocked = false;
                add_conv(p, "conv1", -1, 0, 1, 16, T, F, p.W("model.conv1.weight"), p.W("model.conv1.bias"), nullptr, nullptr, act, 1);
                add_conv(p, "conv2", 0, 1, 16, 32, T / 2, F / 2, p.W("model.conv2.weight"), p.W("model.conv2.bias"), nullptr, nullptr, act, 1);
            }
            // fc1's split-K partials are reduced by the classifier tail itself when that is the fused kernel
            static const int tail_on = [] { const char* e = getenv("NWW_TAIL"); return e ? atoi(e) : 1; }();
            add_gemm(p, "fc1", 1, 0, 1, 128, 32 * H2 * W2, p.W("model.fc1.weight"), p.W("model.fc1.bias"), act, nullptr, nullptr, 99, 1.f,
                     &h->trunk_blocked, tail_on && tail_supported(128, E));
            set_tail(p, "fc2", 0, 128, p.W("model.fc2.weight"), p.W("model.fc2.bias"));
            break;
        }
        case NWW_HEAD_E2E_DNN: {                  // E2E_MelSpectrogram_CNN body: architectures.py:840-865,877-889
            const int Hh = T, Ww = F;             // (n_mels, frames)
            const int ch[3] = {16, 32, 64};
            if (e2e_transposed_ok(p, Hh, Ww)) {
                const int Ht = Ww, Wt = Hh;       // the plane the kernels see: (frames, n_mels)
                float* wt = nullptr;
                const int nf[3] = {16, 32 * 16, 64 * 32};
                if (hipMalloc(&wt, (size_t)(nf[0] + nf[1] + nf[2]) * 9 * sizeof(float)) != hipSuccess) return fail(h, NWW_ERR_HIP, "hipMalloc failed");
                p.h->packed_weights.push_back(wt);
                float* wts[3] = {wt, wt + (size_t)nf[0] * 9, wt + (size_t)(nf[0] + nf[1]) * 9};
                for (int i = 0; i < 3; ++i)
                    if (launch_transpose3x3(p.W("model.conv_block." + std::to_string(4 * i) + ".weight"), wts[i], nf[i], p.h->own_stream) != hipSuccess)
                        return fail(h, NWW_ERR_HIP, "weight transpose failed");
                h->e2e_transposed = true;
                p.need(2, (size_t)Hh * Ww);
                p.add("transpose:mel-major features -> frames-major (skipped after the frontend)", [=](Run& r) {
                    if (r.x_frames_major) return hipSuccess;
                    const hipError_t e = launch_transpose_planes(r.x, r.buf[2], r.B, Hh, Ww, r.stream);
                    r.x = r.buf[2];
                    return e;
                });
                if (!add_trunk(p, "conv_block.0-7 (transposed plane)", -1, 1, 16, 32, Ht, Wt, wts[0], p.W("model.conv_block.0.bias"),
                               p.W("model.conv_block.1.alpha"), p.W("model.conv_block.1.beta"), wts[1], p.W("model.conv_block.4.bias"),
                               p.W("model.conv_block.5.alpha"), p.W("model.conv_block.5.beta"), act))
                    return fail(h, NWW_ERR_UNSUPPORTED, "e2e_dnn: the transposed trunk does not fit");
                const int h3 = Ht / 4, w3 = Wt / 4;           // (25, 16): AdaptiveAvgPool2d((1,4))'s windows run along the FRAMES, here y
                const int sw4 = h3 / 4, kw4 = h3 - 3 * sw4;
                if (h3 < 4 || !add_conv_mfma(p, "model.conv_block.8 (transposed plane)", 1, 0, 32, 64, h3, w3, wts[2], p.W("model.conv_block.8.bias"),
                                   p.W("model.conv_block.9.alpha"), p.W("model.conv_block.9.beta"), act, 0, kw4, sw4, 4, nullptr, 1))
                    return fail(h, NWW_ERR_UNSUPPORTED, "e2e_dnn: the transposed third conv does not fit");
                add_gemm(p, "fc1+bn1", 0, 1, 1, 128, 256, p.W("model.fc1.weight"), p.W("model.fc1.bias"), act, p.W("model.bn1.alpha"), p.W("model.bn1.beta"));
                set_tail(p, "out", 1, 128, p.W("model.out.weight"), p.W("model.out.bias"));
                break;
            }
            int cin = 1, hh = Hh, ww = Ww, cur = -1;
            int first = 0;
            bool fused_pool = false;
            if (add_trunk(p, "conv_block.0-7", -1, 1, 16, 32, Hh, Ww, p.W("model.conv_block.0.weight"), p.W("model.conv_block.0.bias"),
                          p.W("model.conv_block.1.alpha"), p.W("model.conv_block.1.beta"), p.W("model.conv_block.4.weight"),
                          p.W("model.conv_block.4.bias"), p.W("model.conv_block.5.alpha"), p.W("model.conv_block.5.beta"), act)) {
                first = 2; cin = 32; hh = Hh / 4; ww = Ww / 4; cur = 1;
            }
            for (int i = first; i < 3; ++i) {
                const std::string cw = "model.conv_block." + std::to_string(4 * i), bnp = "model.conv_block." + std::to_string(4 * i + 1);
                const int out = (i % 2 == 0) ? 0 : 1;
                if (i == 2 && ww >= 4) {
                    // conv3 + AdaptiveAvgPool2d((1,4)) in its exported AvgPool2d form, fused when the MFMA kernel applies
                    const int sw4 = ww / 4, kw4 = ww - 3 * sw4;
                    if (add_conv_mfma(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 0, kw4, sw4, 4)) {
                        fused_pool = true; cin = ch[i]; cur = out;
                        continue;
                    }
                }
                if (!add_conv_mfma(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, i < 2))
                    add_conv(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, i < 2);
                if (i < 2) { hh /= 2; ww /= 2; }
                cin = ch[i]; cur = out;
            }
            if (hh < 1 || ww < 4) return fail(h, NWW_ERR_INVALID, "e2e_dnn input too small for AdaptiveAvgPool2d((1,4))");
            // AdaptiveAvgPool2d((1,4)) in its exported AvgPool2d form (_export/onnx.py:146-152)
            const int sh = hh / 1, kh = hh, sw = ww / 4, kw = ww - 3 * sw;
            int fc_in = cur;                                  // buffer holding [B][256] after the pool
            if (!fused_pool) {
                const int pin = cur, pout = cur ^ 1;
                p.need(pout, 256);
                p.add("avgpool:export(1,4)", [=](Run& r) { return launch_avgpool(r.buf[pin], r.buf[pout], r.B * 64, hh, ww, kh, kw, sh, sw, 1, 4, r.stream); });
                fc_in = pout;
            }
            const int fc_out = fc_in ^ 1;
            add_gemm(p, "fc1+bn1", fc_in, fc_out, 1, 128, 256, p.W("model.fc1.weight"), p.W("model.fc1.bias"), act, p.W("model.bn1.alpha"), p.W("model.bn1.beta"));
            set_tail(p, "out", fc_out, 128, p.W("model.out.weight"), p.W("model.out.bias"));
            break;
        }
        case NWW_HEAD_CRNN: {                     // CRNNModel: architectures.py:209-287
            int cin = 1, hh = T, ww = F, cur = -1;
            int first = 0;
            bool seq_written = false;
            if (c.n_crnn_channels >= 2 && c.crnn_channels[0] == 16 && c.crnn_channels[1] == 32 &&
                add_trunk(p, "cnn.0-7", -1, 1, 16, 32, T, F, p.W("model.cnn.0.weight"), p.W("model.cnn.0.bias"), p.W("model.cnn.1.alpha"),
                          p.W("model.cnn.1.beta"), p.W("model.cnn.4.weight"), p.W("model.cnn.4.bias"), p.W("model.cnn.5.alpha"),
                          p.W("model.cnn.5.beta"), act)) {
                first = 2; cin = 32; hh = T / 4; ww = F / 4; cur = 1;
            }
            for (int i = first; i < c.n_crnn_channels; ++i) {
                const std::string cw = "model.cnn." + std::to_string(4 * i), bnp = "model.cnn." + std::to_string(4 * i + 1);
                const int out = (i % 2 == 0) ? 0 : 1;
                // the last conv stage may write the recurrent layers' [W][C * H] sequence layout itself (conv3_x3.hip)
                bool seq = i == c.n_crnn_channels - 1;
                if (!add_conv_mfma(p, cw, cur, out, cin, c.crnn_channels[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 1, 0, 0, 0, &seq)) {
                    seq = false;
                    add_conv(p, cw, cur, out, cin, c.crnn_channels[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 1);
                }
                seq_written = seq;
                hh /= 2; ww /= 2; cin = c.crnn_channels[i]; cur = out;
            }
            if (hh < 1 || ww < 1) return fail(h, NWW_ERR_INVALID, "crnn input too small for the conv stack");
            const int seq = seq_written ? cur : cur ^ 1, C = cin, Hc = hh, Wc = ww;
            if (!seq_written) {
                p.need(seq, (size_t)C * Hc * Wc);
                p.add("crnn_seq", [=](Run& r) { return launch_crnn_seq(r.buf[cur], r.buf[seq], r.B, C, Hc, Wc, r.stream); });
            }
            add_bigru_last(p, "model.rnn", seq, Wc, C * Hc, L, nb, 2, seq ^ 1, 3, 4, c.crnn_rnn_lstm ? 4 : 3);      // seq ^ 1: the free one of buffers 0 / 1
            set_tail(p, "fc", 4, 2 * L, p.W("model.fc.weight"), p.W("model.fc.bias"));
            break;
        }
        case NWW_HEAD_GRU: {                      // GRUModel: architectures.py:129-145
            add_bigru_last(p, "model.gru", -1, T, F, L, nb, 2, 0, 1, 4);
            set_tail(p, "fc", 4, 2 * L, p.W("model.fc.weight"), p.W("model.fc.bias"));
            break;
        }
        case NWW_HEAD_BCRESNET: {                 // BcResNetModel: architectures.py:620-687, channels-last on the GPU
            // init conv (+BN+act+pool) writes [B][H1][W1][32]; each block: one depthwise kernel emits d = dw3x3(x) and
            // xs = x at the strided centres, then two MFMA GEMMs over M = B*Ho*Wo pixels:
            //   R = BN_s(xs . Wsc^T) ;  out = act(BN_1(d . Wpw^T)) + R      (activation BEFORE the residual add, :646-647)
            static const int ic_mfma = [] { const char* e = getenv("NWW_CONV_MFMA"); return e ? atoi(e) : 1; }();
            // init conv fused with block1's depthwise (trunk.hip: the 32-channel planes never reach HBM)
            static const int bc_front = [] { const char* e = getenv("NWW_BC_FRONT"); return e ? atoi(e) : 1; }();
            const bool front_fused = ic_mfma && bc_front && conv1_pool_nhwc_mfma_fits(T, F) && conv1_pool_dw_rows(T, F, 2) > 0;
            // nww_config.act_dtype = NWW_ACT_DTYPE_BF16: every activation tensor between the kernels of this head is stored as bf16
            // (arithmetic and accumulation stay float32); implemented on the fused front + split-operand block path only
            const bool act_bf16 = c.act_dtype == NWW_ACT_DTYPE_BF16;
            if (act_bf16 && !(front_fused && p.h->conv_products == 6))
                return fail(h, NWW_ERR_UNSUPPORTED, "act_dtype = bf16 needs the fused BcResNet front kernel and conv_arith bf16x6 for this input shape");
            if (front_fused) {
                const float *w0 = p.W("model.init_conv.0.weight"), *a0 = p.W("model.init_conv.1.alpha"), *b0 = p.W("model.init_conv.1.beta");
                const float* dwt1 = p.W("model.block1.depthwise.weight_t");
                const int ho1 = (T / 2 - 1) / 2 + 1, wo1 = (F / 2 - 1) / 2 + 1;
                p.need(2, (size_t)32 * ho1 * wo1); p.need(3, (size_t)32 * ho1 * wo1);
                const int max_grid = p.h->cu_count;
                // the convolution from split operands on the bf16 matrix cores (trunk_b.hip) under the handle's arithmetic switch;
                // NWW_BC_FRONT = 2 keeps the float32-MFMA kernel
                void* fpack = nullptr;
                const int fprod = p.h->conv_products;
                if ((fprod == 6 || fprod == 9) && bc_front != 2 && bc_front_b_rows(T, F, 2) > 0 &&
                    hipMalloc(&fpack, bc_front_b_packed_bytes()) == hipSuccess) {
                    if (launch_bc_front_b_pack(w0, static_cast<unsigned char*>(fpack), p.h->own_stream) == hipSuccess) p.h->packed_weights.push_back(fpack);
                    else { (void)hipFree(fpack); fpack = nullptr; }
                }
                p.add(std::string(fpack ? "conv1_dw_x3" : "conv1_dw_mfma") + ":init_conv + block1.depthwise (nhwc" + (act_bf16 ? ", bf16 out)" : ")"), [=](Run& r) {
                    Conv1DwArgs a{src(r, -1), w0, nullptr, a0, b0, dwt1, r.buf[2], r.buf[3], r.B, T, F, act, 2, 2};
                    a.bf16_out = act_bf16 ? 1 : 0;
                    if (fpack) {
                        a.wpack = static_cast<const unsigned char*>(fpack);
                        return launch_bc_front_b(a, fprod, max_grid, r.stream);
                    }
                    return launch_conv1_pool_dw_nhwc(a, max_grid, r.stream);
                });
            } else if (ic_mfma && conv1_pool_nhwc_mfma_fits(T, F)) {
                const float *w0 = p.W("model.init_conv.0.weight"), *a0 = p.W("model.init_conv.1.alpha"), *b0 = p.W("model.init_conv.1.beta");
                p.need(0, (size_t)32 * (T / 2) * (F / 2));
                const int max_grid = p.h->cu_count;
                p.add("conv1_mfma:init_conv(nhwc)", [=](Run& r) {
                    Conv1NhwcArgs a{src(r, -1), w0, nullptr, a0, b0, r.buf[0], r.B, T, F, act};
                    return launch_conv1_pool_nhwc_mfma(a, max_grid, r.stream);
                });
            } else {
                add_conv(p, "init_conv(nhwc)", -1, 0, 1, 32, T, F, p.W("model.init_conv.0.weight"), nullptr, p.W("model.init_conv.1.alpha"), p.W("model.init_conv.1.beta"), act, 1, 1);
            }
            int hh = T / 2, ww = F / 2, cur = 0;
            const int ch[4] = {32, 64, 128, 256};
            const int st[3][2] = {{2, 2}, {2, 2}, {2, 1}};
            for (int i = 1; i <= 3; ++i) {
                const std::string q = "model.block" + std::to_string(i);
                const int ci = ch[i - 1], co = ch[i], sh = st[i - 1][0], sw = st[i - 1][1];
                const int ho = (hh - 1) / sh + 1, wo = (ww - 1) / sw + 1;
                const int dwb = 2, xsb = 3, resb = 4, outb = cur ^ 1;
                p.need(dwb, (size_t)ci * ho * wo); p.need(xsb, (size_t)ci * ho * wo);
                const float* dwt = p.W(q + ".depthwise.weight_t");
                const int hin = hh, win = ww;
                if (!(front_fused && i == 1))
                    p.add("dwconv3x3_nhwc:" + q, [=](Run& r) { return launch_dwconv3x3_nhwc(r.buf[cur], dwt, r.buf[dwb], r.buf[xsb], r.B, ci, hin, win, sh, sw, r.stream); });
                // one dual GEMM per block: shortcut and pointwise products in the same workgroup, no residual round trip
                {
                    const float *wpw = p.W(q + ".pointwise.weight"), *a1 = p.W(q + ".bn1.alpha"), *b1 = p.W(q + ".bn1.beta");
                    const float *wsc = p.W(q + ".shortcut.0.weight"), *as = p.W(q + ".shortcut.1.alpha"), *bs = p.W(q + ".shortcut.1.beta");
                    const int rows = ho * wo;
                    p.need(outb, (size_t)rows * co);
                    // both products from split operands on the bf16 matrix cores (dual_x3.hip) under the same arithmetic switch
                    static const int dual_x3 = [] { const char* e = getenv("NWW_BC_DUAL_X3"); return e ? atoi(e) : 1; }();
                    void* packed = nullptr;
                    if (dual_x3 && p.h->conv_products == 6 && dual_x3_supported(ci, co) &&
                        hipMalloc(&packed, dual_x3_packed_bytes(ci, co)) == hipSuccess) {
                        if (launch_dual_x3_pack(wpw, wsc, a1, b1, as, bs, packed, ci, co, p.h->own_stream) == hipSuccess) {
                            p.h->packed_weights.push_back(packed);
                            // when the block input is in HBM (every block but the one whose depthwise ran inside the fused front kernel) the
                            // shortcut rows are gathered from it and the depthwise kernel planned just above writes no copy of them
                            const bool gather = !(front_fused && i == 1);
                            if (gather) {
                                p.pop_last();
                                p.add("dwconv3x3_nhwc:" + q, [=](Run& r) { return launch_dwconv3x3_nhwc(r.buf[cur], dwt, r.buf[dwb], nullptr, r.B, ci, hin, win, sh, sw, r.stream, act_bf16); });
                            }
                            p.add(std::string(gather ? "dual_x3(xs gathered):" : "dual_x3:") + q + ".pointwise+bn+act + shortcut+bn" + (act_bf16 ? " (bf16 activations)" : ""), [=](Run& r) {
                                DualArgs a{r.buf[dwb], r.buf[xsb], r.buf[outb], static_cast<const unsigned char*>(packed), r.B * rows, co};
                                if (gather) { a.x = r.buf[cur]; a.H = hin; a.W = win; a.Ho = ho; a.Wo = wo; a.sh = sh; a.sw = sw; }
                                a.bf16 = act_bf16 ? 1 : 0;
                                return launch_dual_x3(a, ci, act, r.stream);
                            });
                            hh = ho; ww = wo; cur = outb;
                            continue;
                        }
                        (void)hipFree(packed);
                    }
                    if (act_bf16) return fail(h, NWW_ERR_UNSUPPORTED, "act_dtype = bf16: block %d has no split-operand kernel (channels %d -> %d)", i, ci, co);
                    p.add("gemm2:" + q + ".pointwise+bn+act + shortcut+bn", [=](Run& r) {
                        GemmArgs g;
                        g.A = r.buf[dwb]; g.lda = ci; g.W = wpw; g.K = ci; g.alpha = a1; g.beta = b1; g.bias = nullptr; g.act = act;
                        g.A2 = r.buf[xsb]; g.lda2 = ci; g.W2 = wsc; g.K2 = ci; g.alpha2 = as; g.beta2 = bs;
                        g.C = r.buf[outb]; g.ldc = co; g.M = r.B * rows; g.N = co;
                        g.res = nullptr; g.ldres = 0; g.rscale = 1.0f;
                        return launch_gemm(g, r.stream);
                    });
                    (void)resb;
                }
                hh = ho; ww = wo; cur = outb;
            }
            const int hw = hh * ww;
            p.need(2, 256);
            p.add("mean:global_avg_pool", [=](Run& r) { return launch_mean_mid(r.buf[cur], r.buf[2], r.B, hw, 256, r.stream, act_bf16); });
            set_tail(p, "fc", 2, 256, p.W("model.fc.weight"), p.W("model.fc.bias"));
            break;
        }
        case NWW_HEAD_CONFORMER: {                // ConformerModel: architectures.py:441-543
            const int D = c.conformer_d_model, NH = c.conformer_n_head;
            const int hb = 0, t1 = 1, t3 = 2, big = 3;      // h, LN/glu/attn scratch, dwconv scratch, wide scratch
            bool last_fused = false;
            p.need(t1, (size_t)T * D); p.need(t3, (size_t)T * D);
            if (!add_lin_x3(p, "input_proj", -1, hb, T, D, F, p.W("model.input_proj.weight"), p.W("model.input_proj.bias"), 0))
                add_gemm(p, "input_proj", -1, hb, T, D, F, p.W("model.input_proj.weight"), p.W("model.input_proj.bias"), ACT_NONE);
            for (int i = 0; i < nb; ++i) {
                const std::string q = "model.conformer_blocks." + std::to_string(i);
                auto ffn = [&](const std::string& ff) {
                    const float *lw = p.W(q + ff + ".layer_norm.weight"), *lb = p.W(q + ff + ".layer_norm.bias");
                    // LayerNorm + linear1 + swish + linear2 + half-step residual in one kernel (ffn_x3.hip); same arithmetic
                    // switch as the split-operand GEMMs it replaces
                    static const int fused = [] { const char* e = getenv("NWW_FFN_FUSED"); return e ? atoi(e) : 1; }();
                    if (fused && p.h->conv_products == 6 && ffn_x3_supported(D)) {
                        void* packed = nullptr;
                        if (hipMalloc(&packed, ffn_x3_packed_bytes(D)) == hipSuccess &&
                            launch_ffn_x3_pack(p.W(q + ff + ".linear1.weight"), p.W(q + ff + ".linear1.bias"),
                                               p.W(q + ff + ".linear2.weight"), packed, D, p.h->own_stream) == hipSuccess) {
                            p.h->packed_weights.push_back(packed);
                            const float* b2 = p.W(q + ff + ".linear2.bias");
                            p.add("ffn_x3:" + q + ff + " (ln+linear1+swish+linear2+0.5res)", [=](Run& r) {
                                FfnArgs a{r.buf[hb], lw, lb, static_cast<const unsigned char*>(packed), b2, r.B * T, 0.5f};
                                return launch_ffn_x3(a, D, r.stream);
                            });
                            return;
                        }
                        if (packed) (void)hipFree(packed);
                    }
                    p.add("layernorm:" + q + ff, [=](Run& r) { return launch_layernorm(r.buf[hb], r.buf[t1], lw, lb, r.B * T, D, ACT_NONE, r.stream); });
                    add_gemm(p, q + ff + ".linear1+swish", t1, big, T, 4 * D, D, p.W(q + ff + ".linear1.weight"), p.W(q + ff + ".linear1.bias"), ACT_SILU);
                    add_gemm(p, q + ff + ".linear2+0.5res", big, hb, T, D, 4 * D, p.W(q + ff + ".linear2.weight"), p.W(q + ff + ".linear2.bias"), ACT_NONE, nullptr, nullptr, hb, 0.5f);
                };
                ffn(".ff1");
                // in_proj writes q, k, v head-major when the matrix-core attention consumes them: every (clip, head) block is then
                // one contiguous run for its LDS-DMA (NWW_QKV_HEAD_MAJOR=0: nn.Linear's [B][T][3 D] rows)
                static const int mha_mfma0 = [] { const char* e = getenv("NWW_MHA_MFMA"); return e ? atoi(e) : 1; }();
                const bool want_hm = mha_mfma0 && mha_mfma_supported(T, D, NH) && 3 * D <= 1024;
                bool head_major = false;
                if (add_lin_x3(p, q + (want_hm ? ".attention.in_proj(head-major)" : ".attention.in_proj"), hb, big, T, 3 * D, D, p.W(q + ".attention.in_proj_weight"), p.W(q + ".attention.in_proj_bias"), 0,
                               99, 1.f, nullptr, nullptr, want_hm ? T : 0, want_hm ? D / NH : 0))
                    head_major = want_hm;
                else
                    add_gemm(p, q + ".attention.in_proj", hb, big, T, 3 * D, D, p.W(q + ".attention.in_proj_weight"), p.W(q + ".attention.in_proj_bias"), ACT_NONE);
                static const int mha_mfma = [] { const char* e = getenv("NWW_MHA_MFMA"); return e ? atoi(e) : 1; }();
                if (mha_mfma && mha_mfma_supported(T, D, NH))
                    p.add("mha_mfma:" + q, [=](Run& r) { return launch_mha_mfma(r.buf[big], r.buf[t1], r.B, T, D, NH, r.stream, head_major ? 1 : 0); });
                else
                    p.add("mha_core:" + q, [=](Run& r) { return launch_mha_core(r.buf[big], r.buf[t1], r.B, T, D, NH, r.stream); });
                if (!add_lin_x3(p, q + ".attention.out_proj+res", t1, hb, T, D, D, p.W(q + ".attention.out_proj.weight"), p.W(q + ".attention.out_proj.bias"), 1, hb, 1.0f))
                    add_gemm(p, q + ".attention.out_proj+res", t1, hb, T, D, D, p.W(q + ".attention.out_proj.weight"), p.W(q + ".attention.out_proj.bias"), ACT_NONE, nullptr, nullptr, hb, 1.0f);
                {
                    const std::string m = q + ".conv_module";
                    const float *lw = p.W(m + ".layer_norm.weight"), *lb = p.W(m + ".layer_norm.bias");
                    // LayerNorm + pointwise conv1 + GLU in one launch (lin_x3.hip), else the three separate ones
                    if (!add_lin_x3(p, m + ".layer_norm+conv1(pw)+glu", hb, t1, T, D, D, p.W(m + ".conv1.weight"), p.W(m + ".conv1.bias"), 2, 99, 1.f, lw, lb)) {
                        p.add("layernorm:" + m, [=](Run& r) { return launch_layernorm(r.buf[hb], r.buf[t1], lw, lb, r.B * T, D, ACT_NONE, r.stream); });
                        add_gemm(p, m + ".conv1(pw)", t1, big, T, 2 * D, D, p.W(m + ".conv1.weight"), p.W(m + ".conv1.bias"), ACT_NONE);
                        p.add("glu:" + m, [=](Run& r) { return launch_glu(r.buf[big], r.buf[t1], r.B * T, D, r.stream); });
                    }
                    const float *dw = p.W(m + ".depthwise_conv.weight"), *db = p.W(m + ".depthwise_conv.bias");
                    const float *ba = p.W(m + ".batch_norm.alpha"), *bb = p.W(m + ".batch_norm.beta");
                    p.add("dwconv1d+bn+swish:" + m, [=](Run& r) { return launch_dwconv1d_bn_swish(r.buf[t1], dw, db, ba, bb, r.buf[t3], r.B, T, D, 31, r.stream); });
                    if (!add_lin_x3(p, m + ".conv2(pw)+res", t3, hb, T, D, D, p.W(m + ".conv2.weight"), p.W(m + ".conv2.bias"), 1, hb, 1.0f))
                        add_gemm(p, m + ".conv2(pw)+res", t3, hb, T, D, D, p.W(m + ".conv2.weight"), p.W(m + ".conv2.bias"), ACT_NONE, nullptr, nullptr, hb, 1.0f);
                }
                ffn(".ff2");
                const float *lw = p.W(q + ".layer_norm.weight"), *lb = p.W(q + ".layer_norm.bias");
                // the last block's LayerNorm feeds only the mean over time: one pass for both (NWW_LN_MEAN=0: two launches)
                if (i == nb - 1 && D <= 256) {
                    p.add("layernorm+mean:" + q + " + time", [=](Run& r) { return launch_ln_mean(r.buf[hb], r.buf[t1], lw, lb, r.B, T, D, r.stream); });
                    last_fused = true;
                } else {
                    p.add("layernorm:" + q, [=](Run& r) { return launch_layernorm(r.buf[hb], r.buf[hb], lw, lb, r.B * T, D, ACT_NONE, r.stream); });
                }
            }
            if (!last_fused) p.add("mean:time", [=](Run& r) { return launch_mean_mid(r.buf[hb], r.buf[t1], r.B, T, D, r.stream); });
            set_tail(p, "output_proj", t1, D, p.W("model.output_proj.weight"), p.W("model.output_proj.bias"));
            break;
        }
    }
    // embedding Linear + Model.classifier (model.py:291-296) (+ sigmoid) -> emb [B][E], logits [B] (, probs [B])
    static const int tail_fused = [] { const char* e = getenv("NWW_TAIL"); return e ? atoi(e) : 1; }();
    if (tail_fused && tail_supported(p.tail_K, E)) {
        const float *We = p.tail_W, *be = p.tail_b, *W0 = p.W("classifier.0.weight"), *b0 = p.W("classifier.0.bias"),
                    *w3 = p.W("classifier.3.weight"), *b3 = p.W("classifier.3.bias");
        const int tin = p.tail_in, tK = p.tail_K;
        p.add("tail:" + p.tail_name + "+classifier", [=](Run& r) {
            TailArgs t{src(r, tin), tK, We, be, E, W0, b0, w3, b3, r.emb, r.logits, r.probs, r.B, act};
            if (r.deferred.active && r.deferred.out_id == tin) {
                t.parts = r.splitk_ws; t.nparts = r.deferred.parts; t.part_stride = r.deferred.stride;
                t.in_bias = r.deferred.bias; t.in_alpha = r.deferred.alpha; t.in_beta = r.deferred.beta; t.in_act = r.deferred.act;
            }
            r.deferred.active = false;
            r.need_sigmoid = false;
            if (r.done_flag && r.B <= 16) { t.done_flag = r.done_flag; t.done_seq = r.done_seq; r.done_armed = true; }
            return launch_classifier_tail(t, r.stream);
        });
    } else {
        add_gemm(p, p.tail_name, p.tail_in, -2, 1, E, p.tail_K, p.tail_W, p.tail_b, ACT_NONE);
        add_gemm(p, "classifier.0", -2, -3, 1, E / 2, E, p.W("classifier.0.weight"), p.W("classifier.0.bias"), act);
        add_gemm(p, "classifier.3", -3, -4, 1, 1, E / 2, p.W("classifier.3.weight"), p.W("classifier.3.bias"), ACT_NONE);
    }
    // the plan-time weight packings above were enqueued on own_stream; a forward may arrive on any caller stream
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    h->finalized = true;
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

extern "C" int nww_describe_plan(const nww_handle* h, char* buf, int32_t buflen) {
    if (!h || !buf || buflen <= 0) return NWW_ERR_INVALID;
    std::string s = "frontend:fe_stft_mel_db_kernel\n";
    for (const auto& st : h->plan) s += st.name + "\n";
    s += "unary:sigmoid\n";
    std::snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return NWW_OK;
}

// ------------------------------------------------------------------------------------------ workspace / run
static int ensure_ws(nww_handle* h, int B, int N) {
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

static int run_head(nww_handle* h, const float* d_x, int B, float* d_logits, float* d_probs, hipStream_t s, unsigned int* done_flag = nullptr,
                    unsigned int done_seq = 0, bool* done_armed = nullptr, bool x_frames_major = false) {
    Run r;
    r.x_frames_major = x_frames_major;
    r.done_flag = done_flag; r.done_seq = done_seq;
    r.B = B; r.stream = s; r.x = d_x; r.emb = h->d_emb; r.hid = h->d_hid; r.logits = d_logits ? d_logits : h->d_logits;
    r.probs = d_probs;
    r.splitk_ws = h->d_splitk; r.splitk_floats = h->splitk_per_clip * (size_t)h->cap_B; r.cu_count = h->cu_count;
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

static int check_run(nww_handle* h, int B) {
    if (!h) return NWW_ERR_INVALID;
    if (!h->finalized) return fail(h, NWW_ERR_STATE, "model not finalized: call nww_finalize after loading the state_dict");
    if (B <= 0) return fail(h, NWW_ERR_INVALID, "batch must be positive (got %d)", B);
    return NWW_OK;
}

static int frontend_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_db, float* d_mel, int frames_major,
                        hipStream_t s, int* frames_out, size_t row_stride = 0) {
    const int T = fe_num_frames(h->fe, N);
    if (T <= 0) return fail(h, NWW_ERR_INVALID, "clip of %d samples is too short for n_fft=%d (center=%d)", N, h->fe.n_fft, h->fe.center);
    if (frames_out) *frames_out = T;
    static const int mel_env = [] { const char* e = getenv("NWW_FE_MEL"); return e ? atoi(e) : 2; }();    // 2: register filters, 1: MFMA tiles, 0: sparse LDS loop
    // three 4-wave workgroups per CU (frontend2.hip)
    hipError_t e = fe2_launch(d_pcm, row_stride ? row_stride : (size_t)N, B, N, T, h->fe, h->d_tables, h->d_melplan, d_db, d_mel, frames_major, mel_env, h->mel_max_taps, 256, h->cu_count * 3, s);
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

static int forward_pcm_dev(nww_handle* h, const int16_t* d_pcm, int B, int N, float* d_logits, float* d_probs, hipStream_t s,
                           size_t row_stride = 0, unsigned int* done_flag = nullptr, unsigned int done_seq = 0, bool* done_armed = nullptr) {
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
static int h2d_small(nww_handle* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes <= PIN_BYTES) {
        if (!h->pin_in) HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), PIN_BYTES, hipHostMallocDefault));
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
    if (!h->pin_in && hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), PIN_BYTES, hipHostMallocDefault) != hipSuccess) return false;
    if (!h->pin_out && hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), PIN_BYTES, hipHostMallocDefault) != hipSuccess) return false;
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

static int copy_out(nww_handle* h, int B, float* logits, float* probs, float* emb, hipStream_t s) {
    const size_t nb = (size_t)B * sizeof(float), ne = (size_t)B * h->cfg.embedding_dim * sizeof(float);
    const size_t need = (logits ? nb : 0) + (probs ? nb : 0) + (emb ? ne : 0);
    if (need <= PIN_BYTES) {
        if (!h->pin_out) HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), PIN_BYTES, hipHostMallocDefault));
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

// ------------------------------------------------------------------------------------------ RCCL (multi-GPU gather)
// The path's only exchange: an all-gather of the per-clip float32 logits (4 B per clip) over RCCL / xGMI, enqueued on
// the SAME stream as the kernels so a step never touches the host.  RCCL is bound at run time (dlopen): a process
// that already carries one (PyTorch's bundled librccl.so) is reused, otherwise the system librccl.so.1 is loaded; a
// single-GPU user never loads it at all.
struct NcclId { char internal[128]; };       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
}  // namespace
static RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        // 1. the RCCL that sits next to the HIP runtime this process actually runs on (a PyTorch process carries its own
        //    libamdhip64 + librccl pair; mixing one stack's RCCL with the other's HSA runtime fails at communicator
        //    creation), 2. one that is already loaded, 3. the system library
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.find_last_of('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* name : {"librccl.so", "librccl.so.1"}) {
                    a.lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (a.lib) break;
                }
            }
        }
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        }
        if (!a.lib) a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) a.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) { a.err = std::string("cannot load RCCL: ") + dlerror(); return a; }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) a.err = "RCCL library lacks the expected symbols";
        return a;
    }();
    return api;
}
static const char* rccl_str(int rc) { return rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error"; }

extern "C" int nww_comm_unique_id(void* id128) {
    if (!id128) return NWW_ERR_INVALID;
    RcclApi& a = rccl();
    if (!a.err.empty()) { g_create_err = a.err; return NWW_ERR_UNSUPPORTED; }
    const int rc = a.GetUniqueId(id128);
    if (rc != 0) { g_create_err = std::string("ncclGetUniqueId: ") + rccl_str(rc); return NWW_ERR_HIP; }
    return NWW_OK;
}

extern "C" int nww_comm_destroy(nww_handle* h) {
    if (!h) return NWW_ERR_INVALID;
    if (h->comm) {
        (void)hipSetDevice(h->cfg.device);
        (void)rccl().CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0; h->comm_world = 1;
    return NWW_OK;
}

extern "C" int nww_comm_init(nww_handle* h, int32_t rank, int32_t world, const void* id128) {
    if (!h) return NWW_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world || !id128) return fail(h, NWW_ERR_INVALID, "nww_comm_init: bad rank/world/id");
    RcclApi& a = rccl();
    if (!a.err.empty()) return fail(h, NWW_ERR_UNSUPPORTED, "%s", a.err.c_str());
    nww_comm_destroy(h);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    NcclId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    const int rc = a.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(h, NWW_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_str(rc));
    h->comm = comm; h->comm_rank = rank; h->comm_world = world;
    return NWW_OK;
}

static int all_gather_dev(nww_handle* h, const float* d_send, float* d_recv, int count, hipStream_t s) {
    if (!h->comm) return fail(h, NWW_ERR_STATE, "no communicator (nww_comm_init)");
    const int rc = rccl().AllGather(d_send, d_recv, (size_t)count, /* ncclFloat32 */ 7, h->comm, s);
    if (rc != 0) return fail(h, NWW_ERR_HIP, "ncclAllGather: %s", rccl_str(rc));
    return NWW_OK;
}

// d_send [count] of this rank -> d_recv [world][count] on every rank, enqueued on `stream` (no synchronisation)
extern "C" int nww_all_gather_logits(nww_handle* h, const float* d_send, float* d_recv, int32_t count, void* stream) {
    if (!h) return NWW_ERR_INVALID;
    if (!d_send || !d_recv || count <= 0) return fail(h, NWW_ERR_INVALID, "nww_all_gather_logits: bad arguments");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return all_gather_dev(h, d_send, d_recv, count, stream ? (hipStream_t)stream : h->own_stream);
}

// One sharded step without a host hop: this rank's B clips -> its B logits (written at d_all_logits + rank * B), then
// the all-gather into d_all_logits [world][B], both on `stream`.
extern "C" int nww_forward_pcm_gather_dev(nww_handle* h, const int16_t* d_pcm, int32_t B, int32_t N, float* d_all_logits, void* stream) {
    int rc = check_run(h, B);
    if (rc) return rc;
    if (!d_pcm || !d_all_logits) return fail(h, NWW_ERR_INVALID, "null device pointer");
    if (!h->comm) return fail(h, NWW_ERR_STATE, "no communicator (nww_comm_init)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
    float* mine = d_all_logits + (size_t)h->comm_rank * B;
    rc = forward_pcm_dev(h, d_pcm, B, N, mine, nullptr, s);
    if (rc) return rc;
    return all_gather_dev(h, mine, d_all_logits, B, s);       // in place: send buffer = this rank's slot of the receive buffer
}

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
            const unsigned int seq = ++h->done_seq;
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
            const unsigned int seq = ++h->done_seq;
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
