// nww_plan.hip - Model.state_dict() spec per head and the launch plans nww_finalize builds from the loaded weights.
#include "nww_internal.h"
#include "bc_chain.h"
#define prof_mark nww_prof_mark
#define prof_begin nww_prof_begin
#define ensure_ws nww_ensure_ws
#define run_head nww_run_head
#define check_run nww_check_run
#define frontend_dev nww_frontend_on_dev
#define forward_pcm_dev nww_forward_pcm_on_dev
#define h2d_small nww_h2d_small
#define copy_out nww_copy_out

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
}  // namespace

void nww_build_spec(nww_handle* h) {
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
    // DNN: LayerNorm1 + blocks run inside the tail's launch (TailArgs::ln0_w)
    bool dnn_body = false; const float *dnn_ln0_w = nullptr, *dnn_ln0_b = nullptr; int dnn_n_mid = 0; const float* dnn_mid[4][4] = {};
};

// ---- two-term binary16 arithmetic (NWW_ARITH_F16X3): plan-time bounds and power-of-two scales
// device array -> host (plan time only; pack kernels of own_stream may still be writing derived arrays)
std::vector<float> f16_fetch(nww_handle* h, const float* d, size_t n) {
    std::vector<float> v(n, 0.0f);
    if (d && n) {
        (void)hipStreamSynchronize(h->own_stream);
        if (hipMemcpy(v.data(), d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) v.assign(n, INFINITY);
    }
    return v;
}
double f16_pow2_floor(double x) { int e; (void)std::frexp(x, &e); return std::ldexp(1.0, e - 1); }      // largest power of two <= x
// scale that keeps |v| <= bound inside the binary16 range (65504), with 2 % to spare for the host / device difference of the bound
float f16_scale(double bound) {
    if (!(bound > 1e-30)) bound = 1e-30;
    if (!(bound < 1e30)) return 0.0f;                             // no usable bound
    double s = f16_pow2_floor(65504.0 / (bound * 1.02));
    if (s > 1099511627776.0) s = 1099511627776.0;                // 2^40
    return (float)s;
}
float f16_wscale(const std::vector<float>& w) {                   // the largest weight lands in [2^14, 2^15)
    double m = 0;
    for (float x : w) m = std::fmax(m, std::fabs((double)x));
    return f16_scale(m * 2.0);
}
// |act((sum_k w[c][k] x[k] + b[c]) al[c] + be[c])| <= max_c (sum_k |w[c][k]| bound + |b[c]|) |al[c]| + |be[c]| for ReLU / GELU / SiLU
double f16_layer_bound(const std::vector<float>& w, int Cout, int K, const std::vector<float>& b, bool has_b,
                       const std::vector<float>& al, const std::vector<float>& be, bool has_bn, double in_bound) {
    double worst = 0;
    for (int c = 0; c < Cout; ++c) {
        double t = 0;
        for (int k = 0; k < K; ++k) t += std::fabs((double)w[(size_t)c * K + k]);
        t = t * in_bound + (has_b ? std::fabs((double)b[c]) : 0.0);
        if (has_bn) t = t * std::fabs((double)al[c]) + std::fabs((double)be[c]);
        worst = std::fmax(worst, t);
    }
    return worst;
}

// A tensor's plan-time range: a BOUND on its magnitude (what the scale is derived from) and a rough TYPICAL magnitude.  Two binary16
// terms hold 22-23 bits of a value down to 2^-17 of the scaled bound; a bound that is more pessimistic than that against the values
// that matter would cost precision silently - so a layer takes the two-term form only when bound <= 2^16 typ (typ-sized values then
// sit at >= 2^-1 after scaling, one bit above where `lo` starts to lose bits), and otherwise stays on
// the three-term bf16 form (which has float32's range and needs no bound).  typ: the head's features ~ 32 (dB values), a layer's
// output ~ sqrt(sum_k w^2) typ_in (uncorrelated terms; |folded-BN alpha| applied) - order-of-magnitude, which is all the guard needs.
// how far above the typical magnitude a worst-case bound may lie: 2^16 leaves typical values 22+ significant bits in two binary16 terms.
// NWW_F16_RANGE_LOG2 (A/B and test knob): a larger exponent lets deeper stages take the two-term kernels at reduced precision of small values
inline double f16_range_factor() {
    static const double f = [] { const char* e = getenv("NWW_F16_RANGE_LOG2"); return std::ldexp(1.0, e ? atoi(e) : 16); }();
    return f;
}
struct F16Range {
    double bound = 0.0, typ = 0.0;
    bool ok() const { return bound > 0.0 && typ > 0.0 && bound <= typ * f16_range_factor(); }
};
const F16Range F16_FEATURES{NWW_F16_FEATURE_BOUND, 32.0};
double f16_layer_typ(const std::vector<float>& w, int Cout, int K, const std::vector<float>& al, bool has_bn, double typ_in) {
    double acc = 0;
    for (int c = 0; c < Cout; ++c) {
        double q = 0;
        for (int k = 0; k < K; ++k) q += (double)w[(size_t)c * K + k] * (double)w[(size_t)c * K + k];
        acc += std::sqrt(q) * (has_bn ? std::fabs((double)al[c]) : 1.0);
    }
    return acc / (Cout > 0 ? Cout : 1) * typ_in;
}

// source selector for a step input: -1 = head input x, -2 = emb, -3 = hid, >=0 workspace buffer
inline const float* src(Run& r, int id) { return id == -1 ? r.x : id == -2 ? r.emb : id == -3 ? r.hid : r.buf[id]; }
inline float* dst(Run& r, int id) { return id == -2 ? r.emb : id == -3 ? r.hid : id == -4 ? r.logits : r.buf[id]; }

// rows_per_clip: M = B*rows_per_clip
void add_gemm(PlanCtx& p, const std::string& name, int in_id, int out_id, int rows_per_clip, int N, int K,
              const float* W, const float* bias, int act, const float* alpha = nullptr, const float* beta = nullptr,
              int res_id = 99, float rscale = 1.f, bool* a_blocked_inout = nullptr, bool feeds_tail = false, bool feeds_ln = false,
              F16Range a_range = F16Range{}, bool a_bound_assumed = false) {
    const double a_bound = a_range.ok() ? a_range.bound : 0.0;
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
    // two binary16 terms (NWW_ARITH_F16X3) when the caller knows a bound on |A|: A times a_scale stays inside the binary16 range
    float h2_as = 0.0f, h2_ws = 0.0f;
    if (use_x3 && p.h->f16 && a_bound > 0.0) {
        h2_as = f16_scale(a_bound);
        if (h2_as > 0.0f) h2_ws = f16_wscale(f16_fetch(p.h, W, (size_t)N * K));
    }
    const bool h2_req = h2_as > 0.0f && h2_ws > 0.0f;
    if (use_x3) {
        // the packed image comes in two formats (scaled binary16 terms / bf16 terms): the cache is keyed on the weight AND the format
        // (with its scale), so a second layer on the same W with a different decision gets its own image
        const X3Key key{W, h2_req ? 1 : 0, h2_req ? h2_ws : 0.0f};
        auto it = p.h->x3_weights.find(key);
        if (it == p.h->x3_weights.end()) {
            void* d = nullptr;
            if (hipMalloc(&d, gemm_x3_weight_bytes(N, K)) == hipSuccess &&
                (h2_req ? launch_split_weights_h2(W, d, N, K, h2_ws, p.h->own_stream) : launch_split_weights_x3(W, d, N, K, p.h->own_stream)) == hipSuccess)
                it = p.h->x3_weights.emplace(key, d).first;
            else if (d) (void)hipFree(d);
        }
        if (it != p.h->x3_weights.end()) wx3 = it->second;
    }
    const bool h2 = h2_req && wx3 != nullptr;                  // the two-term epilogue only with its image
    if (h2 && a_bound_assumed && in_id == -1) p.h->clamps_features = true;
    // the producer may write A directly as this kernel's [128][32] tiles (the fused trunk feeding fc1)
    const int a_blocked = (a_blocked_inout && *a_blocked_inout && wx3 && K % 32 == 0 && rows_per_clip == 1) ? K / 32 : 0;
    if (a_blocked_inout) *a_blocked_inout = a_blocked != 0;
    if (K >= 2048 && (size_t)16 * rows_per_clip * N > p.h->splitk_per_clip) p.h->splitk_per_clip = (size_t)16 * rows_per_clip * N;
    p.add("gemm:" + name + (h2 && wx3 ? " [f16x3]" : ""), [=](Run& r) {
        GemmArgs g;
        g.A = src(r, in_id); g.lda = K; g.W = W; g.C = dst(r, out_id); g.ldc = N;
        g.M = r.B * rows_per_clip; g.N = N; g.K = K; g.bias = bias; g.alpha = alpha; g.beta = beta; g.act = act;
        g.res = res_id == 99 ? nullptr : src(r, res_id); g.ldres = N; g.rscale = rscale;
        g.Wx3 = wx3;
        if (h2) { g.h2 = 1; g.a_scale = h2_as; g.c_scale = 1.0f / (h2_as * h2_ws); g.a_clamp = a_bound_assumed ? (float)a_bound : 0.0f; }
        g.a_blocked = a_blocked;
        g.splitk = gemm_recommended_splitk(g.M, N, K, r.cu_count);
        // split-operand layers: chunks of ~16-25 k-tiles, so that a small batch's chunk is ONE round of gemm_x3_chain_kernel
        // (32 k-tiles in flight) on a few dozen CUs - K = 12 800: 16 chunks of 25, K = 6 464: 12 of 17, K = 3 920 (C1): 7 of 18
        if (wx3 && x3_mode != 2 && K >= 2048) { g.splitk = K >= 8192 ? K / 800 : K / 512; if (g.splitk > 16) g.splitk = 16; if (g.splitk < 1) g.splitk = 1; }
        g.splitk_ws = r.splitk_ws;
        if (g.splitk > 1 && (size_t)g.splitk * g.M * N > r.splitk_floats) g.splitk = 1;
        r.deferred.active = false;
        // only the MFMA kernels leave split-K partials; the VALU fallback (K % 4 != 0 or an A pointer that is not 16-byte aligned)
        // writes C itself, so nothing may be deferred to the consumer then
        const bool partials = g.splitk > 1 && g.splitk_ws && gemm_writes_partials(g);
        // a handful of clips (the interpreter's calls): the fused tail sums the partials itself, in the same order - one
        // dependent launch less (B = 1: 62 -> 58 us back-to-back).  Larger batches keep the reduce launch: the tail's few
        // workgroups read the 16 partials slower than the full-grid reduce does (B = 4096: 0.028 vs 0.019 + 0.007 ms).
        if (feeds_tail && g.M <= 8 && partials && !g.res) {
            g.defer_reduce = true;
            r.deferred.active = true; r.deferred.out_id = out_id; r.deferred.parts = g.splitk; r.deferred.stride = (size_t)g.M * N;
            r.deferred.bias = bias; r.deferred.alpha = alpha; r.deferred.beta = beta; r.deferred.act = act;
        }
        // a LayerNorm right behind a split-K Linear (DNN layer1) sums the partials itself, at every batch size: the reduce launch
        // and its round trip go (the caller's LayerNorm step checks r.deferred)
        if (feeds_ln && partials && !g.res && !alpha && act == ACT_NONE && N <= 256) {
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
    if (!enabled || p.h->conv_products != 6 || !lin_x3_supported(K, N, true)) return false;
    const int parts = epi == 2 ? 2 : 1;
    // under NWW_ARITH_F16X3: two binary16 terms per operand, the input rows scaled per row in the kernel (LinArgs::h2: no bound on the
    // tensor needed); conv_arith = bf16x6 keeps the three-term bf16 form
    float ws = 0.0f;
    if (p.h->f16) ws = f16_wscale(f16_fetch(p.h, W, (size_t)parts * N * K));
    const bool h2 = ws > 0.0f;
    if (!lin_x3_supported(K, N, h2)) return false;                 // K = 192 / 256 exist in the two-term form only
    const int terms = h2 ? 2 : 3;
    void* packed = nullptr;
    if (hipMalloc(&packed, lin_x3_packed_bytes(K, N, parts, terms)) != hipSuccess) return false;
    if (launch_lin_x3_pack(W, bias, packed, K, N, parts, N, p.h->own_stream, terms, h2 ? ws : 1.0f) != hipSuccess) { (void)hipFree(packed); return false; }
    p.h->packed_weights.push_back(packed);
    p.need(out_id, (size_t)rows_per_clip * N);
    const float w_un = h2 ? 1.0f / ws : 1.0f;
    p.add("lin_x3:" + name + (h2 ? " [f16x3]" : ""), [=](Run& r) {
        LinArgs a;
        a.x = src(r, in_id); a.ldx = K; a.out = dst(r, out_id); a.ldc = N;
        a.res = res_id == 99 ? nullptr : src(r, res_id); a.ldres = N; a.rscale = rscale;
        a.ln_w = ln_w; a.ln_b = ln_b; a.packed = static_cast<const unsigned char*>(packed);
        a.M = r.B * rows_per_clip; a.N = N; a.qkv_T = qkv_T; a.qkv_dh = qkv_dh;
        a.h2 = h2 ? 1 : 0; a.w_un = w_un;
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
               const float* b2, const float* al2, const float* be2, int act, const bool* out_blocked = nullptr,
               F16Range in_range = F16Range{}, F16Range* out_range = nullptr) {
    if (out_range) *out_range = F16Range{};
    const double in_bound = in_range.ok() ? in_range.bound : 0.0;
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
        // NWW_ARITH_F16X3: two binary16 terms per operand when the input is bounded; the scales from bounds on conv1's / conv2's outputs
        float f_in = 0.0f, f_s1 = 0.0f, f_w1 = 0.0f, f_w2 = 0.0f;
        double bound2 = 0.0, typ2 = 0.0;
        if (p.h->f16 && x3 == 6 && in_bound > 0.0 && (act == ACT_RELU || act == ACT_GELU || act == ACT_SILU)) {
            const auto hw1 = f16_fetch(p.h, w1, (size_t)C1 * 9), hw2 = f16_fetch(p.h, w2, (size_t)C2 * C1 * 9);
            const auto hb1 = f16_fetch(p.h, b1, C1), hb2 = f16_fetch(p.h, b2, C2);
            const auto ha1 = f16_fetch(p.h, al1, C1), he1 = f16_fetch(p.h, be1, C1), ha2 = f16_fetch(p.h, al2, C2), he2 = f16_fetch(p.h, be2, C2);
            const double bound1 = f16_layer_bound(hw1, C1, 9, hb1, b1 != nullptr, ha1, he1, al1 != nullptr, in_bound);
            bound2 = f16_layer_bound(hw2, C2, C1 * 9, hb2, b2 != nullptr, ha2, he2, al2 != nullptr, bound1);
            const F16Range r1{bound1, f16_layer_typ(hw1, C1, 9, ha1, al1 != nullptr, in_range.typ)};
            typ2 = f16_layer_typ(hw2, C2, C1 * 9, ha2, al2 != nullptr, r1.typ);
            if (r1.ok()) { f_in = f16_scale(in_bound); f_s1 = f16_scale(bound1); f_w1 = f16_wscale(hw1); f_w2 = f16_wscale(hw2); }
        }
        const bool f16 = f_in > 0.0f && f_s1 > 0.0f && f_w1 > 0.0f && f_w2 > 0.0f;
        if ((f16 ? launch_trunk_b_pack_f16(w1, w2, static_cast<unsigned char*>(packed), f_w1, f_w2, p.h->own_stream)
                 : launch_trunk_b_pack(w1, w2, static_cast<unsigned char*>(packed), p.h->own_stream)) != hipSuccess) { (void)hipFree(packed); return false; }
        p.h->packed_weights.push_back(packed);
        if (f16 && out_range) *out_range = F16Range{bound2, typ2};
        const int products = f16 ? 3 : x3;
        // BN + ReLU stems (CRNN, E2E): all folded-BN factors of both layers non-negative (gamma > 0, the usual case) -> the pooled value is
        // the window's maximum pushed through BN + ReLU (trunk_b.hip: POS instance)
        int bn_pos = 0;
        if (f16 && al1 && al2 && act == ACT_RELU) {
            bn_pos = 1;
            for (float v : f16_fetch(p.h, al1, C1)) if (!(v >= 0.0f)) bn_pos = 0;
            for (float v : f16_fetch(p.h, al2, C2)) if (!(v >= 0.0f)) bn_pos = 0;
        }
        if (in_id == -1 && !p.h->e2e_transposed) p.h->x_stride_ok = true;                   // this step reads the head input with any clip stride
        if (f16 && in_id == -1) p.h->clamps_features = true;
        p.add("trunk_x3:" + name + (f16 ? " [f16x3]" : ""), [=](Run& r) {
            TrunkArgs a{src(r, in_id), w1, b1, al1, be1, w2, b2, al2, be2, dst(r, out_id), r.B, H, W, act};
            if (out_blocked && *out_blocked) a.out_blocked = C2 * (H / 4) * (W / 4) / 32;     // decided by the consumer (add_gemm) at plan time
            a.wpack = static_cast<const unsigned char*>(packed);
            if (f16) { a.f16_in = f_in; a.f16_k1 = f_in * f_w1; a.f16_s1 = f_s1; a.f16_k2 = f_s1 * f_w2; a.f16_so = 1.0f; a.f16_clamp = (float)in_bound; }
            a.bn_pos = bn_pos;
            if (in_id == -1) a.in_clip_stride = r.x_stride;
            if (r.stream_mode) {                        // streaming hop: pooled rows into the per-stream rings, all of them or the invalidated ones
                a.out = r.a2_ring; a.out_ring_rows = r.a2_rows; a.out_row0 = r.a2_row0;
                a.out_ch_stride = r.a2_ch_stride; a.out_clip_stride = r.a2_clip_stride;
                a.n_sub = r.stream_mode == 2 ? r.a2_nsub : 0;
                for (int q = 0; q < a.n_sub; ++q) { a.sub_a[q] = r.a2_sub_a[q]; a.sub_b[q] = r.a2_sub_b[q]; }
            }
            return launch_cnn_trunk_b(a, products, max_grid, r.stream);
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
                   int avg_kw = 0, int avg_sw = 0, int avg_ow = 0, bool* seq_inout = nullptr, int avg_y = 0, bool* ring_in = nullptr,
                   F16Range in_range = F16Range{}, F16Range* out_range = nullptr, int scratch_id = -1) {
    if (out_range) *out_range = F16Range{};
    const double in_bound = in_range.ok() ? in_range.bound : 0.0;
    static const int enabled = [] { const char* e = getenv("NWW_CONV_MFMA"); return e ? atoi(e) : 1; }();
    static const int x3_enabled = [] { const char* e = getenv("NWW_CONV3_X3"); return e ? atoi(e) : 1; }();
    // 64 input channels (a fourth CRNN stage): conv3_x3's wide instance - two-term arithmetic on a bounded input only
    bool wide = Cin == 64 && enabled && x3_enabled && pool && avg_ow == 0 && !avg_y && p.h->conv_products == 6 && p.h->f16 && in_bound > 0.0 &&
                Cout % 32 == 0 && conv3_x3_wide_fits(Cin, H, W, Cout);
    if (wide) {
        const auto hw = f16_fetch(p.h, w, (size_t)Cout * Cin * 9);
        if (f16_scale(in_bound) <= 0.0f || f16_wscale(hw) <= 0.0f) wide = false;
    }
    // more than 32 input channels without a usable bound (the worst-case bound of a fourth stage is usually too loose for two terms): the
    // 32-channel three-term instance once per 32 input channels - every pass but the last leaves raw, un-pooled sums in a scratch buffer, the
    // next one starts its accumulators from them (k-split; the planes of such stages are a few hundred pixels)
    static const int ksplit_on = 1;
    if (!wide && Cin > 32) {
        if (!ksplit_on || !enabled || !x3_enabled || Cin % 32 != 0 || Cin > 256 || !pool || avg_ow > 0 || avg_y || p.h->conv_products != 6 || scratch_id < 0 ||
            Cout % 32 != 0) return false;
        const bool whole = conv3_x3_fits(H, W, Cout, 0, 1) && conv3_x3_fits(H, W, Cout, 0, 0);
        const int sh = whole ? 0 : conv3_x3_strip_rows(H, W, Cout);          // larger planes: in strips of rows
        if (!whole && sh == 0) return false;
        if (ring_in) *ring_in = false;
        p.need(out_id, (size_t)Cout * (H / 2) * (W / 2));
        p.need(scratch_id, (size_t)Cout * H * W);
        const int np = Cin / 32, grid = p.h->cu_count * ((sh ? conv3_x3_strip_lds_bytes(sh, W) : conv3_x3_lds_bytes(H, W, 0)) * 2 <= 160 * 1024 ? 2 : 1);
        const int seq_out = (seq_inout && *seq_inout) ? 1 : 0;
        for (int pass = 0; pass < np; ++pass) {
            const bool fin = pass == np - 1;
            p.add(std::string(fin && seq_out ? "conv3_x3+seq:" : "conv3_x3:") + name + " (input channels " + std::to_string(32 * pass) + "-" + std::to_string(32 * pass + 31) +
                      (fin ? ", epilogue)" : ", raw sums)"), [=](Run& r) {
                ConvMfmaArgs a{src(r, in_id), w, fin ? bias : nullptr, fin ? alpha : nullptr, fin ? beta : nullptr, fin ? dst(r, out_id) : r.buf[scratch_id],
                               r.B, H, W, Cout, fin ? act : (int)ACT_NONE, fin ? 1 : 0};
                a.w_cin = Cin; a.w_cin_off = 32 * pass; a.acc_in = pass ? r.buf[scratch_id] : nullptr; a.seq_out = fin ? seq_out : 0; a.strip_h = sh;
                return launch_conv3_x3(a, grid, r.stream);
            });
        }
        return true;
    }
    if (wide) {
        if (ring_in) *ring_in = false;
    } else if (!enabled || Cin != 32 || Cout % 32 != 0 || H < 2 || W < 2)
        return false;
    // (the float32-MFMA instance shares its 8 waves among the 32-channel groups; conv3_x3 takes one group per workgroup)
    const bool f32_fits = !wide && (8 % (Cout / 32)) == 0 && conv_mfma_lds_bytes(Cin, H, W) <= 160 * 1024;
    // pooled planes of more than 512 pixels (clips longer than ~1.3 s, a second stage behind an unfused first one): conv3_x3 in strips of rows
    const int strip_h = (!wide && pool && avg_ow == 0 && !avg_y && !conv3_x3_fits(H, W, Cout, avg_ow, pool)) ? conv3_x3_strip_rows(H, W, Cout) : 0;
    const bool x3_shape = strip_h > 0 || conv3_x3_fits(H, W, Cout, avg_ow, pool);
    if (!wide && !f32_fits && !(x3_enabled && p.h->conv_products == 6 && x3_shape)) return false;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    p.need(out_id, avg_ow > 0 ? (size_t)Cout * avg_ow : (size_t)Cout * Ho * Wo);
    const int max_grid = p.h->cu_count;
    // split-operand bf16 instance (conv3_x3.hip) under the same arithmetic switch as the fused trunk; the 9-product
    // mode keeps the float32-MFMA kernel (the conv3 instance implements the 6-product form only)
    if (wide || (x3_enabled && p.h->conv_products == 6 && x3_shape)) {
        const size_t lds = wide ? conv3_x3_wide_lds_bytes(Cin, H, W) : strip_h ? conv3_x3_strip_lds_bytes(strip_h, W) : conv3_x3_lds_bytes(H, W, avg_ow);
        if (strip_h && ring_in) *ring_in = false;                                             // strips read dense planes
        const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
        const int seq_out = (seq_inout && *seq_inout && pool && avg_ow == 0) ? 1 : 0;      // the caller wants the sequence layout
        const bool ring_ok = ring_in && *ring_in;                                             // ... and may hand the input over in rings
        // NWW_ARITH_F16X3: two binary16 terms when the input is bounded (the fused trunk's plan-time bound)
        float h2_in = 0.0f, h2_w = 0.0f;
        if (p.h->f16 && in_bound > 0.0) {
            const auto hw = f16_fetch(p.h, w, (size_t)Cout * Cin * 9);
            h2_in = f16_scale(in_bound); h2_w = f16_wscale(hw);
            if (out_range && h2_in > 0.0f && h2_w > 0.0f) {
                const auto hal = f16_fetch(p.h, alpha, Cout);
                *out_range = F16Range{f16_layer_bound(hw, Cout, Cin * 9, f16_fetch(p.h, bias, Cout), bias != nullptr, hal, f16_fetch(p.h, beta, Cout), alpha != nullptr, in_bound),
                                      f16_layer_typ(hw, Cout, Cin * 9, hal, alpha != nullptr, in_range.typ)};
            }
        }
        const bool h2 = h2_in > 0.0f && h2_w > 0.0f;
        p.add(std::string(avg_ow > 0 ? "conv3_x3+avgpool:" : seq_out ? "conv3_x3+seq:" : "conv3_x3:") + name + (strip_h ? " (strips of " + std::to_string(strip_h) + " rows)" : "") + (h2 ? " [f16x3]" : ""), [=](Run& r) {
            ConvMfmaArgs a{src(r, in_id), w, bias, alpha, beta, dst(r, out_id), r.B, H, W, Cout, act, pool};
            a.Cin = Cin; a.w_cin = Cin; a.strip_h = strip_h;
            a.avg_kw = avg_kw; a.avg_sw = avg_sw; a.avg_ow = avg_ow; a.seq_out = seq_out; a.avg_y = avg_y;
            if (h2) { a.h2_in = h2_in; a.h2_w = h2_w; }
            if (ring_ok && r.stream_mode) {             // streaming hop: the fused trunk's pooled rows are read from their rings
                a.in = r.a2_ring; a.in_ring_rows = r.a2_rows; a.in_row0 = r.a2_row0;
                a.in_ch_stride = r.a2_ch_stride; a.in_clip_stride = r.a2_clip_stride;
                if (seq_out && r.seq_new) {             // ... and the sequence rows a hop does not invalidate are carried over from the previous hop's buffer
                    r.buf[out_id] = r.seq_new;          // (the recurrent layers planned behind this step read r.buf[out_id])
                    a.out = r.seq_new;
                    a.seq_prev = r.seq_prev; a.keep_lo = r.a3_lo; a.keep_hi = r.a3_hi; a.keep_shift = r.a3_shift;
                }
            }
            return launch_conv3_x3(a, max_grid * per_cu, r.stream);
        });
        return true;
    }
    if (avg_y) return false;                                   // only conv3_x3 pools along y (the caller checked e2e_transposed_ok)
    if (seq_inout) *seq_inout = false;                         // the float32-MFMA instance writes planes
    if (ring_in) *ring_in = false;                             // ... and reads dense planes
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
    static const int on = 1;
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
    // the recurrent product follows the handle's arithmetic switch; NWW_ARITH_F16X3: two binary16 terms (|h| <= 1 bounds the one
    // operand, W_hh's scale comes from the weights)
    static const int rnn_h2 = 1;
    const int products = (p.h->f16 && rnn_h2) ? 3 : p.h->conv_products;
    GruArgs probe; probe.H = H; probe.products = products;
    probe.w_hh = p.W(prefix + ".weight_hh_l" + std::to_string(layers - 1));       // the pointer the fused launch will really get (alignment test)
    static const int gru16_on = 1;
    const bool streamed = gru16_on && rnn_stream_usable(probe);      // 128 < H <= 256: W_hh streamed from L2 (rnn_stream.hip)
    bool x3 = probe.w_hh != nullptr && (rnn_x3_enabled(probe) || streamed);
    if (x3 && H != 32 && H != 64 && H != 128) {              // zero-padded / streamed instances: two-term form only, so the last layer's W_hh must scale
        const float* wl = probe.w_hh;
        if (!(f16_wscale(f16_fetch(p.h, wl, (size_t)G * H * H)) > 0.0f)) x3 = false;
    }
    int cur_in = in_id, cur_I = I;
    for (int l = 0; l < layers; ++l) {
        const bool last = l == layers - 1;
        const int seq_out = (l % 2 == 0) ? seqA : seqB;
        if (!last) p.need(seq_out, (size_t)T * 2 * H);
        // rnn_out[:, -1] needs ONE step of the last layer's reverse direction, hence the input projection of frame T-1 only (a
        // strided GEMM over M = B rows instead of B*T, into the row behind the forward direction's xg) - and with h = 0 that
        // step has no recurrent product: rnn_x3 computes it in the forward direction's launch.
        const bool fold = last && x3;
        // GRU head, first layer fed by the head's own input features (<= 64 per frame), two-term arithmetic: the forward direction's input
        // projection is computed inside the recurrence kernel - x_t W_ih^T on the matrix pipe beside h W_hh^T - instead of a GEMM that
        // writes T x 3H gate pre-activations per clip to HBM for the recurrence to read back (635 MB each way at B = 4096, T = 101, H = 128).
        // NWW_RNN_FUSE_IH = 0 keeps the separate projection.
        static const int fuse_env = 1;
        const float* wih_f = p.W(prefix + ".weight_ih_l" + std::to_string(l));
        const float wi_scale = (fuse_env && fold && l == 0 && in_id == -1 && G == 3 && products == 3 && (cur_I == 32 || cur_I == 64) && (H == 32 || H == 64 || H == 128) &&
                                wih_f && (reinterpret_cast<uintptr_t>(wih_f) & 15) == 0)
                                   ? f16_wscale(f16_fetch(p.h, wih_f, (size_t)G * H * cur_I)) : 0.0f;
        const float* whh_fwd = p.W(prefix + ".weight_hh_l" + std::to_string(l));
        const bool fuse_ih = wi_scale > 0.0f && f16_scale(F16_FEATURES.bound) > 0.0f && (H % 4) == 0 && H <= 256 &&
                             whh_fwd && f16_wscale(f16_fetch(p.h, whh_fwd, (size_t)G * H * H)) > 0.0f;      // ... and the recurrence itself takes the two-term form
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
            } else if (fuse_ih && dir == 0) {
                // (the forward direction's input projection runs inside its recurrence: GruArgs::fin)
            } else {
                // short-K input projections (the GRU head's 64 mel bins) on the input-stationary kernel; the rest on the general GEMM
                if (!add_lin_x3(p, prefix + ".ih" + sfx, cur_in, xg_id, T, G * H, cur_I, wih, bih, 0))
                    add_gemm(p, prefix + ".ih" + sfx, cur_in, xg_id, T, G * H, cur_I, wih, bih, ACT_NONE);
            }
            if (fold && dir == 0) continue;                  // the forward recurrence is launched after the reverse projection
            const int in_T = T;
            const float* whh_f = fold ? p.W(prefix + ".weight_hh_l" + std::to_string(l)) : whh;
            // widths the register-resident / 32-units-per-wave kernels do not take: zero-padded weight rows for the any-width kernel
            int ldw = 0;
            if (H % 4 != 0 || H > 256) {
                float* padded = nullptr;
                if (hipMalloc(&padded, rnn_wide_weight_bytes(G, H)) == hipSuccess && launch_rnn_pad_weights(whh_f, padded, G, H, p.h->own_stream) == hipSuccess) {
                    p.h->packed_weights.push_back(padded);
                    whh_f = padded; ldw = (H + 7) & ~7;
                } else if (padded) (void)hipFree(padded);
            }
            const float* bhh_f = fold ? p.W(prefix + ".bias_hh_l" + std::to_string(l)) : bhh;
            const std::string nm = fold ? (G == 4 ? "lstm:" : "gru:") + prefix + "_l" + std::to_string(l) + " + first reverse step"
                                        : (G == 4 ? "lstm:" : "gru:") + prefix + sfx;
            const float w_scale = products == 3 ? f16_wscale(f16_fetch(p.h, whh_f, (size_t)G * H * H)) : 1.0f;
            const int products_l = (products == 3 && !(w_scale > 0.0f)) ? p.h->conv_products : products;
            const void* w_packed = nullptr;
            if (streamed && products_l == 3 && ldw == 0) {
                void* pk = nullptr;
                if (hipMalloc(&pk, rnn_stream_packed_bytes(G, H)) == hipSuccess && launch_rnn_stream_pack(whh_f, pk, G, H, w_scale, p.h->own_stream) == hipSuccess) {
                    p.h->packed_weights.push_back(pk);
                    w_packed = pk;
                } else {
                    // the folded form (reverse step inside the forward launch) exists only on the streamed kernel for these widths: without its
                    // packed W_hh every forward would fail at run time - fail the finalize instead
                    if (pk) (void)hipFree(pk);
                    if (fold) p.h->plan_error = "out of device memory for the streamed W_hh image of " + prefix;
                }
            }
            const bool fused_here = fuse_ih && fold && products_l == 3 && ldw == 0;
            const float* bih_f = p.W(prefix + ".bias_ih_l" + std::to_string(l));
            const int fin = cur_I;
            if (fused_here) p.h->clamps_features = true;       // the fused input projection clamps the features it splits
            p.add(nm + (fused_here ? " + input projection" : "") + (products_l == 3 && x3 && ldw == 0 ? (w_packed ? " [f16x3, W_hh streamed]" : " [f16x3]") : ""), [=](Run& r) {
                GruArgs a;
                a.products = products_l; a.w_scale = w_scale; a.w_packed = w_packed;
                if (fused_here) {
                    a.x_in = r.x; a.w_ih = wih_f; a.b_ih = bih_f; a.fin = fin;
                    a.x_scale = f16_scale(F16_FEATURES.bound); a.x_clamp = (float)F16_FEATURES.bound; a.wi_scale = wi_scale;
                }
                a.xg = r.buf[xg_id]; a.w_hh = whh_f; a.b_hh = bhh_f;
                a.seq_out = last ? nullptr : r.buf[seq_out]; a.ld_seq = 2 * H;
                a.last_out = last ? r.buf[last_id] : nullptr; a.ld_last = 2 * H;
                a.B = r.B; a.T = in_T; a.H = H; a.ldw = ldw;
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
        // (the DFT on the matrix pipe - frontend3 - was built in round 5, parity-green and slower: tools/ubench/fe3/)
    }
    // ---- plan
    PlanCtx p{h};
    const int T = c.in_rows, F = c.in_cols, L = c.layer_dim, E = c.embedding_dim, nb = c.n_blocks, act = c.activation;
    switch (c.head_type) {
        case NWW_HEAD_DNN: {                      // Net: architectures.py:110-126
            // Everything behind layer1 runs inside the tail's launch (layers.hip: TailArgs::ln0_w) when the widths allow: LayerNorm1 on
            // layer1's split-K partials, the blocks' Linear + LayerNorm, last_layer, classifier - two launches per forward instead of six
            static const int tail_on = [] { const char* e = getenv("NWW_TAIL"); return e ? atoi(e) : 1; }();
            static const int body_on = 1;
            if (tail_on && body_on && L <= 256 && nb <= 4 && tail_supported(L, E)) {
                add_gemm(p, "layer1", -1, 0, 1, L, T * F, p.W("model.layer1.weight"), p.W("model.layer1.bias"), ACT_NONE, nullptr, nullptr, 99, 1.f, nullptr, false, true, F16_FEATURES, true);
                p.dnn_body = true;
                p.dnn_ln0_w = p.W("model.layernorm1.weight"); p.dnn_ln0_b = p.W("model.layernorm1.bias");
                p.dnn_n_mid = nb;
                for (int i = 0; i < nb; ++i) {
                    const std::string q = "model.blocks." + std::to_string(i);
                    p.dnn_mid[i][0] = p.W(q + ".fcn_layer.weight"); p.dnn_mid[i][1] = p.W(q + ".fcn_layer.bias");
                    p.dnn_mid[i][2] = p.W(q + ".layer_norm.weight"); p.dnn_mid[i][3] = p.W(q + ".layer_norm.bias");
                }
                set_tail(p, "layernorm1+blocks+last_layer", 0, L, p.W("model.last_layer.weight"), p.W("model.last_layer.bias"));
                break;
            }
            add_gemm(p, "layer1", -1, 0, 1, L, T * F, p.W("model.layer1.weight"), p.W("model.layer1.bias"), ACT_NONE, nullptr, nullptr, 99, 1.f, nullptr, false, true, F16_FEATURES, true);
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
            F16Range a2_bound;                    // NWW_ARITH_F16X3: the range of the trunk's output, fc1's operand
            const bool fused = add_trunk(p, "conv1+pool+conv2+pool", -1, 1, 16, 32, T, F, p.W("model.conv1.weight"), p.W("model.conv1.bias"), nullptr, nullptr,
                                         p.W("model.conv2.weight"), p.W("model.conv2.bias"), nullptr, nullptr, act, &h->trunk_blocked,
                                         F16_FEATURES, &a2_bound);
            if (!fused) {
                h->trunk_blocked = false;
                add_conv(p, "conv1", -1, 0, 1, 16, T, F, p.W("model.conv1.weight"), p.W("model.conv1.bias"), nullptr, nullptr, act, 1);
                add_conv(p, "conv2", 0, 1, 16, 32, T / 2, F / 2, p.W("model.conv2.weight"), p.W("model.conv2.bias"), nullptr, nullptr, act, 1);
            }
            // fc1's split-K partials are reduced by the classifier tail itself when that is the fused kernel
            static const int tail_on = [] { const char* e = getenv("NWW_TAIL"); return e ? atoi(e) : 1; }();
            add_gemm(p, "fc1", 1, 0, 1, 128, 32 * H2 * W2, p.W("model.fc1.weight"), p.W("model.fc1.bias"), act, nullptr, nullptr, 99, 1.f,
                     &h->trunk_blocked, tail_on && tail_supported(128, E), false, a2_bound);
            set_tail(p, "fc2", 0, 128, p.W("model.fc2.weight"), p.W("model.fc2.bias"));
            break;
        }
        case NWW_HEAD_E2E_DNN: {                  // E2E_MelSpectrogram_CNN body: architectures.py:840-865,877-889
            const int Hh = T, Ww = F;             // (n_mels, frames)
            const int ch[3] = {16, 32, 64};
            // the transposed plan; whatever goes wrong while it is built (a shape one of its kernels does not take after all, an
            // allocation) drops what was planned and falls through to the reference's orientation below
            const auto try_transposed = [&]() -> bool {
                if (!e2e_transposed_ok(p, Hh, Ww)) return false;
                const size_t steps_before = h->plan.size();
                const int Ht = Ww, Wt = Hh;       // the plane the kernels see: (frames, n_mels)
                float* wt = nullptr;
                const int nf[3] = {16, 32 * 16, 64 * 32};
                if (hipMalloc(&wt, (size_t)(nf[0] + nf[1] + nf[2]) * 9 * sizeof(float)) != hipSuccess) return false;
                p.h->packed_weights.push_back(wt);
                float* wts[3] = {wt, wt + (size_t)nf[0] * 9, wt + (size_t)(nf[0] + nf[1]) * 9};
                for (int i = 0; i < 3; ++i)
                    if (launch_transpose3x3(p.W("model.conv_block." + std::to_string(4 * i) + ".weight"), wts[i], nf[i], p.h->own_stream) != hipSuccess)
                        return false;
                h->e2e_transposed = true;
                p.need(2, (size_t)Hh * Ww);
                p.add("transpose:mel-major features -> frames-major (skipped after the frontend)", [=](Run& r) {
                    if (r.x_frames_major) return hipSuccess;
                    const hipError_t e = launch_transpose_planes(r.x, r.buf[2], r.B, Hh, Ww, r.stream);
                    r.x = r.buf[2];
                    return e;
                });
                const int h3 = Ht / 4, w3 = Wt / 4;           // (25, 16): AdaptiveAvgPool2d((1,4))'s windows run along the FRAMES, here y
                const int sw4 = h3 / 4, kw4 = h3 - 3 * sw4;
                F16Range tbound;                              // NWW_ARITH_F16X3: the range of the trunk's output, the third conv's operand
                const bool ok = add_trunk(p, "conv_block.0-7 (transposed plane)", -1, 1, 16, 32, Ht, Wt, wts[0], p.W("model.conv_block.0.bias"),
                                          p.W("model.conv_block.1.alpha"), p.W("model.conv_block.1.beta"), wts[1], p.W("model.conv_block.4.bias"),
                                          p.W("model.conv_block.5.alpha"), p.W("model.conv_block.5.beta"), act, nullptr, F16_FEATURES, &tbound) &&
                                h->plan.back().name.rfind("trunk_x3:", 0) == 0 && h3 >= 4 &&
                                add_conv_mfma(p, "model.conv_block.8 (transposed plane)", 1, 0, 32, 64, h3, w3, wts[2], p.W("model.conv_block.8.bias"),
                                              p.W("model.conv_block.9.alpha"), p.W("model.conv_block.9.beta"), act, 0, kw4, sw4, 4, nullptr, 1, nullptr, tbound);
                if (!ok) {
                    h->plan.resize(steps_before);
                    h->e2e_transposed = false;
                    return false;
                }
                add_gemm(p, "fc1+bn1", 0, 1, 1, 128, 256, p.W("model.fc1.weight"), p.W("model.fc1.bias"), act, p.W("model.bn1.alpha"), p.W("model.bn1.beta"));
                set_tail(p, "out", 1, 128, p.W("model.out.weight"), p.W("model.out.bias"));
                return true;
            };
            if (try_transposed()) break;
            int cin = 1, hh = Hh, ww = Ww, cur = -1;
            int first = 0;
            bool fused_pool = false;
            F16Range cbound;                                  // NWW_ARITH_F16X3: range of the current stage's input (empty: unknown)
            if (add_trunk(p, "conv_block.0-7", -1, 1, 16, 32, Hh, Ww, p.W("model.conv_block.0.weight"), p.W("model.conv_block.0.bias"),
                          p.W("model.conv_block.1.alpha"), p.W("model.conv_block.1.beta"), p.W("model.conv_block.4.weight"),
                          p.W("model.conv_block.4.bias"), p.W("model.conv_block.5.alpha"), p.W("model.conv_block.5.beta"), act, nullptr, F16_FEATURES, &cbound)) {
                first = 2; cin = 32; hh = Hh / 4; ww = Ww / 4; cur = 1;
            }
            for (int i = first; i < 3; ++i) {
                const std::string cw = "model.conv_block." + std::to_string(4 * i), bnp = "model.conv_block." + std::to_string(4 * i + 1);
                const int out = (i % 2 == 0) ? 0 : 1;
                if (i == 2 && ww >= 4) {
                    // conv3 + AdaptiveAvgPool2d((1,4)) in its exported AvgPool2d form, fused when the MFMA kernel applies
                    const int sw4 = ww / 4, kw4 = ww - 3 * sw4;
                    if (add_conv_mfma(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 0, kw4, sw4, 4,
                                      nullptr, 0, nullptr, cbound)) {
                        fused_pool = true; cin = ch[i]; cur = out;
                        continue;
                    }
                }
                {
                    F16Range nb;
                    if (!add_conv_mfma(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, i < 2,
                                       0, 0, 0, nullptr, 0, nullptr, cbound, &nb))
                        add_conv(p, cw, cur, out, cin, ch[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, i < 2);
                    cbound = nb;
                }
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
            F16Range cbound;                                  // NWW_ARITH_F16X3: range of the current stage's input (empty: unknown)
            const size_t steps_before = h->plan.size();
            if (c.n_crnn_channels >= 2 && c.crnn_channels[0] == 16 && c.crnn_channels[1] == 32 &&
                add_trunk(p, "cnn.0-7", -1, 1, 16, 32, T, F, p.W("model.cnn.0.weight"), p.W("model.cnn.0.bias"), p.W("model.cnn.1.alpha"),
                          p.W("model.cnn.1.beta"), p.W("model.cnn.4.weight"), p.W("model.cnn.4.bias"), p.W("model.cnn.5.alpha"),
                          p.W("model.cnn.5.beta"), act, nullptr, F16_FEATURES, &cbound)) {
                first = 2; cin = 32; hh = T / 4; ww = F / 4; cur = 1;
            }
            for (int i = first; i < c.n_crnn_channels; ++i) {
                const std::string cw = "model.cnn." + std::to_string(4 * i), bnp = "model.cnn." + std::to_string(4 * i + 1);
                const int out = (i % 2 == 0) ? 0 : 1;
                // the last conv stage may write the recurrent layers' [W][C * H] sequence layout itself (conv3_x3.hip)
                bool seq = i == c.n_crnn_channels - 1;
                // the stage right behind a fused split-operand trunk may take its input from the streaming rings (nww_stream.hip)
                bool ring = i == 2 && first == 2 && h->plan.size() == steps_before + 1 && h->plan.back().name.rfind("trunk_x3:", 0) == 0 && ((F / 4) % 4) == 0;
                F16Range nb;
                const bool mf = add_conv_mfma(p, cw, cur, out, cin, c.crnn_channels[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 1, 0, 0, 0, &seq, 0, &ring, cbound, &nb, 2);      // (scratch: the recurrent layers' xg buffer, idle until they run)
                cbound = nb;
                if (!mf) {
                    ring = false;
                    seq = false;
                    add_conv(p, cw, cur, out, cin, c.crnn_channels[i], hh, ww, p.W(cw + ".weight"), p.W(cw + ".bias"), p.W(bnp + ".alpha"), p.W(bnp + ".beta"), act, 1);
                }
                if (ring) {
                    h->stream_conv = true; h->stream_H = T; h->stream_W = F;
                    h->stream_seq = seq && i == c.n_crnn_channels - 1;
                    h->seq_floats = (size_t)c.crnn_channels[i] * (hh / 2) * (ww / 2);
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
            // nww_config.act_dtype = NWW_ACT_DTYPE_BF16 / _F16: every activation tensor between the kernels of this head is stored in 16
            // bits (arithmetic and accumulation stay float32); implemented on the fused front + split-operand block path only.
            // binary16 (11 significant bits against bf16's 8) stores value x a power of two fixed here from a bound on the tensor
            // (features within +-NWW_F16_FEATURE_BOUND as in the f16x3 arithmetic; the bound is not allowed to sit more than 2^16 above
            // the tensor's typical magnitude, and the stores saturate), and a block's weights are two binary16 terms of weight x scale.
            const int act16 = c.act_dtype;                       // NWW_ACT_DTYPE_* == ACT16_* (split_h2.h)
            const bool act_bf16 = act16 != NWW_ACT_DTYPE_F32;    // any 16-bit storage
            const bool act_f16 = act16 == NWW_ACT_DTYPE_F16;
            if (act_bf16 && !(front_fused && p.h->conv_products == 6))
                return fail(h, NWW_ERR_UNSUPPORTED, "act_dtype = bf16 / f16 needs the fused BcResNet front kernel and a split-operand conv_arith for this input shape");
            float s_h[4] = {1.f, 1.f, 1.f, 1.f}, s_d[4] = {1.f, 1.f, 1.f, 1.f};     // scales of h_i (block i's output, h_0 = init conv) and d_i
            DualPackScales dps[4];
            if (act_f16) {
                auto cap = [](F16Range r) { return f16_scale(std::fmin(r.bound, r.typ * 65536.0)); };
                auto dw_range = [&](const float* wt, int C, F16Range in) {      // wt [9][C] tap-major
                    const auto w = f16_fetch(p.h, wt, (size_t)9 * C);
                    double worst = 0, typ = 0;
                    for (int ch = 0; ch < C; ++ch) {
                        double l1 = 0, l2 = 0;
                        for (int k = 0; k < 9; ++k) { const double v = w[(size_t)k * C + ch]; l1 += std::fabs(v); l2 += v * v; }
                        worst = std::fmax(worst, l1); typ += std::sqrt(l2);
                    }
                    return F16Range{worst * in.bound, typ / C * in.typ};
                };
                const auto hw0 = f16_fetch(p.h, p.W("model.init_conv.0.weight"), 32 * 9);
                const float *pa0 = p.W("model.init_conv.1.alpha"), *pb0 = p.W("model.init_conv.1.beta");
                const auto ha0 = f16_fetch(p.h, pa0, 32), hb0 = f16_fetch(p.h, pb0, 32);
                F16Range rh{f16_layer_bound(hw0, 32, 9, hb0, false, ha0, hb0, pa0 != nullptr, F16_FEATURES.bound),
                            f16_layer_typ(hw0, 32, 9, ha0, pa0 != nullptr, F16_FEATURES.typ)};
                s_h[0] = cap(rh);
                const int chs[4] = {32, 64, 128, 256};
                for (int i = 1; i <= 3; ++i) {
                    const std::string q = "model.block" + std::to_string(i);
                    const int ci = chs[i - 1], co = chs[i];
                    const F16Range rd = dw_range(p.W(q + ".depthwise.weight_t"), ci, rh);
                    s_d[i] = cap(rd);
                    const float *pa1 = p.W(q + ".bn1.alpha"), *pas = p.W(q + ".shortcut.1.alpha");
                    const auto wpw = f16_fetch(p.h, p.W(q + ".pointwise.weight"), (size_t)co * ci), wsc = f16_fetch(p.h, p.W(q + ".shortcut.0.weight"), (size_t)co * ci);
                    const auto ha1 = f16_fetch(p.h, pa1, co), hb1 = f16_fetch(p.h, p.W(q + ".bn1.beta"), co);
                    const auto has = f16_fetch(p.h, pas, co), hbs = f16_fetch(p.h, p.W(q + ".shortcut.1.beta"), co);
                    dps[i].pw_ws = f16_wscale(wpw); dps[i].sc_ws = f16_wscale(wsc);
                    if (!(s_d[i] > 0.f && s_h[i - 1] > 0.f && dps[i].pw_ws > 0.f && dps[i].sc_ws > 0.f))
                        return fail(h, NWW_ERR_UNSUPPORTED, "act_dtype = f16: no finite bound on the tensors of block %d", i);
                    dps[i].pw_un = 1.0f / (dps[i].pw_ws * s_d[i]); dps[i].sc_un = 1.0f / (dps[i].sc_ws * s_h[i - 1]);
                    const F16Range rpw{f16_layer_bound(wpw, co, ci, hb1, false, ha1, hb1, pa1 != nullptr, rd.bound), f16_layer_typ(wpw, co, ci, ha1, pa1 != nullptr, rd.typ)};
                    const F16Range rsc{f16_layer_bound(wsc, co, ci, hbs, false, has, hbs, pas != nullptr, rh.bound), f16_layer_typ(wsc, co, ci, has, pas != nullptr, rh.typ)};
                    rh = F16Range{rpw.bound + rsc.bound, std::hypot(rpw.typ, rsc.typ)};
                    s_h[i] = cap(rh);
                }
            }
            if (front_fused) {
                const float *w0 = p.W("model.init_conv.0.weight"), *a0 = p.W("model.init_conv.1.alpha"), *b0 = p.W("model.init_conv.1.beta");
                const float* dwt1 = p.W("model.block1.depthwise.weight_t");
                const int ho1 = (T / 2 - 1) / 2 + 1, wo1 = (F / 2 - 1) / 2 + 1;
                p.need(2, (size_t)32 * ho1 * wo1); p.need(3, (size_t)32 * ho1 * wo1);
                const int max_grid = p.h->cu_count;
                // the convolution from split operands on the bf16 matrix cores (trunk_b.hip) under the handle's arithmetic switch;
                // NWW_BC_FRONT = 2 keeps the float32-MFMA kernel
                void* fpack = nullptr;
                int fprod = p.h->conv_products;
                // under NWW_ARITH_F16X3 (BN present): two binary16 terms per operand, features clamped to +-NWW_F16_FEATURE_BOUND as in the
                // CNN trunk; NWW_BC_FRONT_H2 = 0 keeps the three-term bf16 form
                static const int front_h2_on = 1;
                float fin = 0.0f, fws = 1.0f;
                if (front_h2_on && p.h->f16 && fprod == 6 && a0 && bc_front != 2) {
                    fin = f16_scale(F16_FEATURES.bound); fws = f16_wscale(f16_fetch(p.h, w0, 32 * 9));
                    if (fin > 0.0f && fws > 0.0f) fprod = 3;
                }
                if ((fprod == 6 || fprod == 9 || fprod == 3) && bc_front != 2 && bc_front_b_rows(T, F, 2) > 0 &&
                    hipMalloc(&fpack, bc_front_b_packed_bytes()) == hipSuccess) {
                    const hipError_t pe = fprod == 3 ? launch_bc_front_b_pack_f16(w0, static_cast<unsigned char*>(fpack), fws, p.h->own_stream)
                                                     : launch_bc_front_b_pack(w0, static_cast<unsigned char*>(fpack), p.h->own_stream);
                    if (pe == hipSuccess) p.h->packed_weights.push_back(fpack);
                    else { (void)hipFree(fpack); fpack = nullptr; }
                }
                const float f_un = 1.0f / (fin > 0.0f ? fin * fws : 1.0f);
                // all folded-BN factors non-negative (the usual case: gamma > 0): max-pool commutes with BN + ReLU through the maximum alone
                int bn_pos = 0;
                if (a0) {
                    bn_pos = 1;
                    for (float v : f16_fetch(p.h, a0, 32)) if (!(v >= 0.0f)) bn_pos = 0;
                }
                if (fpack) p.h->clamps_features = true;
                p.add(std::string(fpack ? "conv1_dw_x3" : "conv1_dw_mfma") + ":init_conv + block1.depthwise (nhwc" + (act_f16 ? ", f16 out)" : act_bf16 ? ", bf16 out)" : ")") + (fpack && fprod == 3 ? " [f16x3]" : ""), [=](Run& r) {
                    Conv1DwArgs a{src(r, -1), w0, nullptr, a0, b0, dwt1, r.buf[2], r.buf[3], r.B, T, F, act, 2, 2};
                    a.bf16_out = act16; a.d_scale = s_d[1]; a.xs_scale = s_h[0];
                    if (fpack) {
                        a.wpack = static_cast<const unsigned char*>(fpack);
                        a.f16_in = fin; a.f16_clamp = NWW_F16_FEATURE_BOUND; a.f16_unscale = f_un; a.bn_pos = bn_pos;
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
            bool mean_fused = false;
            const int ch[4] = {32, 64, 128, 256};
            const int st[3][2] = {{2, 2}, {2, 2}, {2, 1}};
            // d_i / xs_i (depthwise output and strided block input) of the coming block live in buffers dwb / xsb; have_dx: they are
            // already there - written by the fused front kernel or by the previous block's chained kernel (bc_chain.hip)
            int dwb = 2, xsb = 3;
            bool have_dx = front_fused;
            static const int chain_on = [] { const char* e = getenv("NWW_BC_CHAIN"); return e ? atoi(e) : 1; }();
            for (int i = 1; i <= 3; ++i) {
                const std::string q = "model.block" + std::to_string(i);
                const int ci = ch[i - 1], co = ch[i], sh = st[i - 1][0], sw = st[i - 1][1];
                const int ho = (hh - 1) / sh + 1, wo = (ww - 1) / sw + 1;
                int outb = 4;
                for (int cand : {0, 1, 4})
                    if (cand != cur && cand != dwb && cand != xsb) { outb = cand; break; }
                const int resb = 4;
                p.need(dwb, (size_t)ci * ho * wo); p.need(xsb, (size_t)ci * ho * wo);
                const float* dwt = p.W(q + ".depthwise.weight_t");
                const int hin = hh, win = ww;
                if (!have_dx)
                    p.add("dwconv3x3_nhwc:" + q, [=](Run& r) { return launch_dwconv3x3_nhwc(r.buf[cur], dwt, r.buf[dwb], r.buf[xsb], r.B, ci, hin, win, sh, sw, r.stream); });
                // one dual GEMM per block: shortcut and pointwise products in the same workgroup, no residual round trip
                {
                    const float *wpw = p.W(q + ".pointwise.weight"), *a1 = p.W(q + ".bn1.alpha"), *b1 = p.W(q + ".bn1.beta");
                    const float *wsc = p.W(q + ".shortcut.0.weight"), *as = p.W(q + ".shortcut.1.alpha"), *bs = p.W(q + ".shortcut.1.beta");
                    const int rows = ho * wo;
                    p.need(outb, (size_t)rows * co);
                    // both products from split operands on the bf16 matrix cores (dual_x3.hip) under the same arithmetic switch
                    static const int dual_x3 = 1;
                    void* packed = nullptr;
                    // float32 activations under NWW_ARITH_F16X3: two binary16 terms per operand, the activation rows scaled per pixel in
                    // the kernel (DualArgs::h2) - no tensor bound needed; NWW_BC_DUAL_H2 = 0 keeps the three-term bf16 form
                    static const int dual_h2_on = 1;
                    const bool dual_h2 = dual_h2_on && p.h->f16 && !act_bf16;
                    // blocks 1 and 2 chained with the next block's depthwise (bc_chain.hip; two-term weights in every storage mode)
                    const bool will_chain = chain_on && i < 3 && have_dx && bc_chain_supported(ci, ho, wo) &&
                                            (act_f16 || dual_h2 || (act16 == NWW_ACT_DTYPE_BF16 && dual_h2_on && p.h->f16));
                    if (dual_h2 || (will_chain && !act_f16)) {
                        dps[i].pw_ws = f16_wscale(f16_fetch(p.h, wpw, (size_t)co * ci)); dps[i].sc_ws = f16_wscale(f16_fetch(p.h, wsc, (size_t)co * ci));
                        if (!(dps[i].pw_ws > 0.f && dps[i].sc_ws > 0.f)) return fail(h, NWW_ERR_INVALID, "block %d: non-finite weights", i);
                        dps[i].pw_un = 1.0f / dps[i].pw_ws; dps[i].sc_un = 1.0f / dps[i].sc_ws;
                    }
                    const int terms = act_f16 || dual_h2 || will_chain ? 2 : 3;
                    if (dual_x3 && p.h->conv_products == 6 && dual_x3_supported(ci, co) &&
                        hipMalloc(&packed, dual_x3_packed_bytes(ci, co, terms)) == hipSuccess) {
                        if (launch_dual_x3_pack(wpw, wsc, a1, b1, as, bs, packed, ci, co, p.h->own_stream, terms, dps[i]) == hipSuccess) {
                            p.h->packed_weights.push_back(packed);
                            // when the block input is in HBM (every block but the one whose depthwise ran inside the fused front kernel) the
                            // shortcut rows are gathered from it and the depthwise kernel planned just above writes no copy of them
                            const bool gather = !have_dx;
                            if (gather) {
                                const float dw_mul = s_d[i] / s_h[i - 1];
                                p.pop_last();
                                p.add("dwconv3x3_nhwc:" + q, [=](Run& r) { return launch_dwconv3x3_nhwc(r.buf[cur], dwt, r.buf[dwb], nullptr, r.B, ci, hin, win, sh, sw, r.stream, act16, dw_mul); });
                            }
                            // blocks 1 and 2 chained with the next block's depthwise (bc_chain.hip): the block's output stays in LDS, the
                            // next block finds its d / xs rows in the other buffer pair
                            const char* suffix = act_f16 ? " (f16 activations)" : act_bf16 ? " (bf16 activations)" : dual_h2 ? " [f16x3]" : "";
                            if (will_chain) {
                                const std::string qn = "model.block" + std::to_string(i + 1);
                                const float* dwn = p.W(qn + ".depthwise.weight_t");
                                const int sh2 = st[i][0], sw2 = st[i][1], ho2 = (ho - 1) / sh2 + 1, wo2 = (wo - 1) / sw2 + 1;
                                const int odb = dwb == 2 ? 0 : 2, oxb = odb + 1, idb = dwb, ixb = xsb;
                                p.need(odb, (size_t)co * ho2 * wo2); p.need(oxb, (size_t)co * ho2 * wo2);
                                const float d_mul = s_d[i + 1], xs_mul = s_h[i];
                                const int max_grid = p.h->cu_count;
                                p.add("bc_chain:" + q + ".pointwise+bn+act + shortcut+bn -> " + qn + ".depthwise" + suffix, [=](Run& r) {
                                    ChainArgs a{r.buf[idb], r.buf[ixb], r.buf[odb], r.buf[oxb], static_cast<const unsigned char*>(packed), dwn,
                                                r.B, ho, wo, sh2, sw2, ho2, wo2};
                                    a.act16 = act16; a.d_mul = d_mul; a.xs_mul = xs_mul;
                                    return launch_bc_chain(a, ci, act, max_grid, r.stream);
                                });
                                hh = ho; ww = wo; dwb = odb; xsb = oxb; have_dx = true;
                                continue;
                            }
                            // the last block feeds only the global average pool: averaged in the same launch, its output never reaches HBM
                            static const int mean_fused_on = 1;
                            const bool fuse_mean = mean_fused_on && i == 3 && ci == 128 && dual_x3_mean_supported(rows);
                            if (fuse_mean) { mean_fused = true; p.need(5, 256); }
                            const float out_mul = fuse_mean ? 1.0f : s_h[i];
                            p.add(std::string(gather ? "dual_x3(xs gathered):" : "dual_x3:") + q + ".pointwise+bn+act + shortcut+bn" + (fuse_mean ? " + global_avg_pool" : "") + suffix, [=](Run& r) {
                                DualArgs a{r.buf[dwb], r.buf[xsb], r.buf[outb], static_cast<const unsigned char*>(packed), r.B * rows, co};
                                if (gather) { a.x = r.buf[cur]; a.H = hin; a.W = win; a.Ho = ho; a.Wo = wo; a.sh = sh; a.sw = sw; }
                                a.act16 = act16; a.out_mul = out_mul; a.h2 = dual_h2 ? 1 : 0;
                                if (fuse_mean) { a.mean_out = r.buf[5]; a.mean_P = rows; }
                                return launch_dual_x3(a, ci, act, r.stream);
                            });
                            hh = ho; ww = wo; cur = outb; have_dx = false; dwb = 2; xsb = 3;
                            continue;
                        }
                        (void)hipFree(packed);
                    }
                    if (act_bf16) return fail(h, NWW_ERR_UNSUPPORTED, "act_dtype = bf16 / f16: block %d has no split-operand kernel (channels %d -> %d)", i, ci, co);
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
                hh = ho; ww = wo; cur = outb; have_dx = false; dwb = 2; xsb = 3;
            }
            const int hw = hh * ww;
            if (mean_fused) {
                set_tail(p, "fc", 5, 256, p.W("model.fc.weight"), p.W("model.fc.bias"));
                break;
            }
            p.need(2, 256);
            const float mean_un = act_f16 ? 1.0f / s_h[3] : 1.0f;
            p.add("mean:global_avg_pool", [=](Run& r) { return launch_mean_mid(r.buf[cur], r.buf[2], r.B, hw, 256, r.stream, act16, mean_un); });
            set_tail(p, "fc", 2, 256, p.W("model.fc.weight"), p.W("model.fc.bias"));
            break;
        }
        case NWW_HEAD_CONFORMER: {                // ConformerModel: architectures.py:441-543
            const int D = c.conformer_d_model, NH = c.conformer_n_head;
            const int hb = 0, t1 = 1, t3 = 2, big = 3;      // h, LN/glu/attn scratch, dwconv scratch, wide scratch
            bool last_fused = false;
            p.need(t1, (size_t)T * D); p.need(t3, (size_t)T * D);
            p.need(hb, (size_t)T * D);
            // round 6: the row-local Linears next to a feed-forward module run INSIDE its launch (FfnArgs::px: input_proj in front of the first
            // block's ff1, conv2 + residual in front of every ff2), and the last block's LayerNorm + time average behind its ff2
            // (FfnArgs::msum); NWW_FFN_FUSED = 0 and the other arithmetics keep the separate launches
            for (int i = 0; i < nb; ++i) {
                const std::string q = "model.conformer_blocks." + std::to_string(i);
                // pro: 0 none, 1 input_proj (x = the head input), 2 conv_module.conv2 + residual (x = the depthwise output in t3); epi: the block's
                // final LayerNorm + mean over time.  With pro / epi the step is planned only if it can be fused that way (false: nothing planned).
                auto ffn = [&](const std::string& ff, int pro, bool epi) -> bool {
                    const float *lw = p.W(q + ff + ".layer_norm.weight"), *lb = p.W(q + ff + ".layer_norm.bias");
                    // LayerNorm + linear1 + swish + linear2 + half-step residual in one kernel (ffn_x3.hip); same arithmetic
                    // switch as the split-operand GEMMs it replaces
                    static const int fused = [] { const char* e = getenv("NWW_FFN_FUSED"); return e ? atoi(e) : 1; }();
                    const int pro_k = pro == 1 ? F : pro == 2 ? D : 0;
                    if ((pro || epi) && !(fused && p.h->f16 && p.h->conv_products == 6 && (!pro || ffn_x3_pro_supported(D, pro_k)) && D == 144 && (!epi || T >= 32))) return false;
                    if (fused && p.h->conv_products == 6 && ffn_x3_supported(D, p.h->f16)) {
                        void* packed = nullptr;
                        // NWW_ARITH_F16X3: both operands of both products are bounded whatever the residual stream holds -
                        // |LayerNorm(h)_i| <= sqrt(D) |w_i| + |b_i|, |swish(v)| <= |v| - so the scales need nothing but the weights
                        float fx = 0.0f, fw1 = 0.0f, fh = 0.0f, fw2 = 0.0f;
                        if (p.h->f16) {
                            const auto hlw = f16_fetch(p.h, lw, D), hlb = f16_fetch(p.h, lb, D);
                            double bx = 0.0;
                            for (int i = 0; i < D; ++i) bx = std::fmax(bx, std::sqrt((double)D) * std::fabs((double)hlw[i]) + std::fabs((double)hlb[i]));
                            const auto w1 = f16_fetch(p.h, p.W(q + ff + ".linear1.weight"), (size_t)4 * D * D), w2 = f16_fetch(p.h, p.W(q + ff + ".linear2.weight"), (size_t)4 * D * D);
                            const auto b1 = f16_fetch(p.h, p.W(q + ff + ".linear1.bias"), (size_t)4 * D);
                            const double bh = f16_layer_bound(w1, 4 * D, D, b1, true, b1, b1, false, bx);
                            fx = f16_scale(bx); fw1 = f16_wscale(w1); fh = f16_scale(bh); fw2 = f16_wscale(w2);
                        }
                        const bool h2 = fx > 0.0f && fw1 > 0.0f && fh > 0.0f && fw2 > 0.0f;
                        // the prologue Linear's weights (two binary16 terms; its input rows are scaled per row in the kernel) and the epilogue's scale
                        const float* pw = pro == 1 ? p.W("model.input_proj.weight") : pro == 2 ? p.W(q + ".conv_module.conv2.weight") : nullptr;
                        const float* pbias = pro == 1 ? p.W("model.input_proj.bias") : pro == 2 ? p.W(q + ".conv_module.conv2.bias") : nullptr;
                        const float *l2w = epi ? p.W(q + ".layer_norm.weight") : nullptr, *l2b = epi ? p.W(q + ".layer_norm.bias") : nullptr;
                        float pws = 0.0f, mscale = 0.0f;
                        void* ppk = nullptr;
                        bool extras_ok = h2 || !(pro || epi);
                        if (pro && extras_ok) {
                            pws = f16_wscale(f16_fetch(p.h, pw, (size_t)D * pro_k));
                            extras_ok = pws > 0.0f && pbias && hipMalloc(&ppk, ffn_x3_pro_tile_bytes(pro_k) * ((D + 31) / 32)) == hipSuccess &&
                                        launch_ffn_x3_pro_pack(pw, ppk, D, pro_k, pws, p.h->own_stream) == hipSuccess;
                        }
                        if (epi && extras_ok) {
                            const auto h2w = f16_fetch(p.h, l2w, D), h2b = f16_fetch(p.h, l2b, D);
                            double by = 0.0;
                            for (int k = 0; k < D; ++k) by = std::fmax(by, std::sqrt((double)D) * std::fabs((double)h2w[k]) + std::fabs((double)h2b[k]));
                            mscale = by < 1e30 ? (float)f16_pow2_floor(68719476736.0 / std::fmax(by, 1e-30)) : 0.0f;      // |LayerNorm| x scale <= 2^36
                            extras_ok = mscale > 0.0f && std::isfinite(mscale);
                        }
                        if (!extras_ok) {
                            if (ppk) (void)hipFree(ppk);
                            if (pro || epi) return false;
                        }
                        if (ffn_x3_supported(D, h2) && hipMalloc(&packed, ffn_x3_packed_bytes(D)) == hipSuccess &&
                            launch_ffn_x3_pack(p.W(q + ff + ".linear1.weight"), p.W(q + ff + ".linear1.bias"),
                                               p.W(q + ff + ".linear2.weight"), packed, D, p.h->own_stream, h2 ? fw1 : 0.0f, h2 ? fw2 : 0.0f, pro ? 1 : 0) == hipSuccess) {
                            p.h->packed_weights.push_back(packed);
                            if (ppk) p.h->packed_weights.push_back(ppk);
                            const float* b2 = p.W(q + ff + ".linear2.bias");
                            const float p_un = pro ? 1.0f / pws : 1.0f;
                            // the epilogue's exact partial sums: per 32-row tile two segments x two planes of D floats (the idle wide scratch buffer)
                            if (epi) p.need(big, (size_t)((T + 31) / 32 + 1) * 4 * D + (size_t)16 * D);
                            const std::string what = std::string(pro == 1 ? "input_proj+" : pro == 2 ? "conv2(pw)+res+" : "") + "ln+linear1+swish+linear2+0.5res" + (epi ? "+layernorm+time sums" : "");
                            p.add("ffn_x3:" + q + ff + " (" + what + ")" + (h2 ? " [f16x3]" : ""), [=](Run& r) {
                                FfnArgs a{r.buf[hb], lw, lb, static_cast<const unsigned char*>(packed), b2, r.B * T, 0.5f};
                                if (h2) { a.h2_x = fx; a.h2_w1 = fw1; a.h2_h = fh; a.h2_w2 = fw2; }
                                if (pro) {
                                    a.px = pro == 1 ? r.x : r.buf[t3]; a.ppacked = static_cast<const unsigned char*>(ppk); a.pb = pbias;
                                    a.pro_k = pro_k; a.pro_res = pro == 2 ? 1 : 0; a.p_un = p_un;
                                }
                                if (epi) { a.ln2_w = l2w; a.ln2_b = l2b; a.msum = r.buf[big]; a.T = T; a.m_scale = mscale; }
                                return launch_ffn_x3(a, D, r.stream);
                            });
                            if (epi)
                                p.add("mean_finish:" + q + " (time average of the exact tile sums)", [=](Run& r) {
                                    return launch_ffn_x3_mean_finish(r.buf[big], r.buf[t1], r.B, T, D, mscale, r.stream);
                                });
                            return true;
                        }
                        if (packed) (void)hipFree(packed);
                        if (ppk) (void)hipFree(ppk);
                        if (pro || epi) return false;
                    }
                    p.add("layernorm:" + q + ff, [=](Run& r) { return launch_layernorm(r.buf[hb], r.buf[t1], lw, lb, r.B * T, D, ACT_NONE, r.stream); });
                    add_gemm(p, q + ff + ".linear1+swish", t1, big, T, 4 * D, D, p.W(q + ff + ".linear1.weight"), p.W(q + ff + ".linear1.bias"), ACT_SILU);
                    add_gemm(p, q + ff + ".linear2+0.5res", big, hb, T, D, 4 * D, p.W(q + ff + ".linear2.weight"), p.W(q + ff + ".linear2.bias"), ACT_NONE, nullptr, nullptr, hb, 0.5f);
                    return true;
                };
                if (!(i == 0 && ffn(".ff1", 1, false))) {
                    if (i == 0 && !add_lin_x3(p, "input_proj", -1, hb, T, D, F, p.W("model.input_proj.weight"), p.W("model.input_proj.bias"), 0))
                        add_gemm(p, "input_proj", -1, hb, T, D, F, p.W("model.input_proj.weight"), p.W("model.input_proj.bias"), ACT_NONE);
                    ffn(".ff1", 0, false);
                }
                // the whole attention module (in_proj, per-head softmax(q k^T) v, out_proj, residual) in one launch per clip-resident
                // workgroup (attn_x3.hip) under the default arithmetic at the compiled shape; NWW_ATTN_FUSED=0: the three launches below
                static const int attn_fused = [] { const char* e = getenv("NWW_ATTN_FUSED"); return e ? atoi(e) : 1; }();
                bool attn_done = false;
                if (attn_fused && p.h->f16 && p.h->conv_products == 6 && attn_x3_supported(T, D, NH)) {
                    const float *iw = p.W(q + ".attention.in_proj_weight"), *ib = p.W(q + ".attention.in_proj_bias");
                    const float *ow = p.W(q + ".attention.out_proj.weight"), *ob = p.W(q + ".attention.out_proj.bias");
                    const auto hiw = f16_fetch(p.h, iw, (size_t)3 * D * D), how = f16_fetch(p.h, ow, (size_t)D * D);
                    const float ws_in = f16_wscale(hiw), ws_out = f16_wscale(how);
                    // |raw k / v accumulator| <= 2^15 (the clip-scaled rows) x the L1 norm of the scaled weight row: powers of two that keep them below 2^15
                    auto l1max = [&](int r0) {
                        double worst = 0;
                        for (int r = r0; r < r0 + D; ++r) {
                            double t = 0;
                            for (int k = 0; k < D; ++k) t += std::fabs((double)hiw[(size_t)r * D + k]);
                            worst = std::fmax(worst, t);
                        }
                        return worst;
                    };
                    const double lk = l1max(D) * ws_in * 1.02, lv = l1max(2 * D) * ws_in * 1.02;
                    void *packed = nullptr, *bc = nullptr;
                    if (ws_in > 0.0f && ws_out > 0.0f && lk < 1e30 && lv < 1e30 &&
                        hipMalloc(&packed, attn_x3_packed_bytes(D, NH)) == hipSuccess && hipMalloc(&bc, (size_t)D * sizeof(float)) == hipSuccess &&
                        launch_attn_x3_pack(iw, ib, ow, ob, packed, static_cast<float*>(bc), D, NH, ws_in, ws_out, p.h->own_stream) == hipSuccess) {
                        p.h->packed_weights.push_back(packed);
                        p.h->packed_weights.push_back(bc);
                        const float cK = lk > 1e-30 ? (float)f16_pow2_floor(1.0 / lk) : 1.0f, cV = lv > 1e-30 ? (float)f16_pow2_floor(1.0 / lv) : 1.0f;
                        const float w_un = 1.0f / ws_in, o_un = 1.0f / (ws_out * ws_in * cV), qs = 1.0f / std::sqrt((float)(D / NH));
                        p.add("attn_x3:" + q + ".attention (in_proj+softmax(qk)v+out_proj+res) [f16x3]", [=](Run& r) {
                            AttnArgs a{r.buf[hb], r.buf[hb], static_cast<const unsigned char*>(packed), static_cast<const float*>(bc), r.B, T, w_un, cK, cV, o_un, qs};
                            return launch_attn_x3(a, D, NH, r.stream);
                        });
                        attn_done = true;
                    } else {
                        if (packed) (void)hipFree(packed);
                        if (bc) (void)hipFree(bc);
                    }
                }
                if (!attn_done) {
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
                static const int mha_h2 = 1;
                if (mha_mfma && mha_h2 && p.h->f16 && mha_h2_supported(T, D, NH))
                    p.add("mha_h2:" + q + " [f16x3]", [=](Run& r) { return launch_mha_h2(r.buf[big], r.buf[t1], r.B, T, D, NH, r.stream, head_major ? 1 : 0); });
                else if (mha_mfma && mha_mfma_supported(T, D, NH))
                    p.add("mha_mfma:" + q, [=](Run& r) { return launch_mha_mfma(r.buf[big], r.buf[t1], r.B, T, D, NH, r.stream, head_major ? 1 : 0); });
                else
                    p.add("mha_core:" + q, [=](Run& r) { return launch_mha_core(r.buf[big], r.buf[t1], r.B, T, D, NH, r.stream); });
                if (!add_lin_x3(p, q + ".attention.out_proj+res", t1, hb, T, D, D, p.W(q + ".attention.out_proj.weight"), p.W(q + ".attention.out_proj.bias"), 1, hb, 1.0f))
                    add_gemm(p, q + ".attention.out_proj+res", t1, hb, T, D, D, p.W(q + ".attention.out_proj.weight"), p.W(q + ".attention.out_proj.bias"), ACT_NONE, nullptr, nullptr, hb, 1.0f);
                }
                {
                    const std::string m = q + ".conv_module";
                    const float *lw = p.W(m + ".layer_norm.weight"), *lb = p.W(m + ".layer_norm.bias");
                    const float *dw = p.W(m + ".depthwise_conv.weight"), *db = p.W(m + ".depthwise_conv.bias");
                    const float *ba = p.W(m + ".batch_norm.alpha"), *bb = p.W(m + ".batch_norm.beta");
                    // (the whole module as ONE clip-resident launch was built and measured: bit-identical, 0.27 ms against 0.25 for the three launches
                    // below - tools/ubench/convmod_x3.hip)
                    {
                    // LayerNorm + pointwise conv1 + GLU in one launch (lin_x3.hip), else the three separate ones
                    if (!add_lin_x3(p, m + ".layer_norm+conv1(pw)+glu", hb, t1, T, D, D, p.W(m + ".conv1.weight"), p.W(m + ".conv1.bias"), 2, 99, 1.f, lw, lb)) {
                        p.add("layernorm:" + m, [=](Run& r) { return launch_layernorm(r.buf[hb], r.buf[t1], lw, lb, r.B * T, D, ACT_NONE, r.stream); });
                        add_gemm(p, m + ".conv1(pw)", t1, big, T, 2 * D, D, p.W(m + ".conv1.weight"), p.W(m + ".conv1.bias"), ACT_NONE);
                        p.add("glu:" + m, [=](Run& r) { return launch_glu(r.buf[big], r.buf[t1], r.B * T, D, r.stream); });
                    }
                    p.add("dwconv1d+bn+swish:" + m, [=](Run& r) { return launch_dwconv1d_bn_swish(r.buf[t1], dw, db, ba, bb, r.buf[t3], r.B, T, D, 31, r.stream); });
                    }
                }
                // conv2 + residual inside ff2's launch, and behind the LAST block's ff2 its LayerNorm + the sums of the time average
                const bool want_epi = i == nb - 1;
                bool ff2_done = ffn(".ff2", 2, want_epi);
                if (ff2_done && want_epi) last_fused = true;
                if (!ff2_done && want_epi) ff2_done = ffn(".ff2", 2, false);
                if (!ff2_done) {
                    const std::string m = q + ".conv_module";
                    if (!add_lin_x3(p, m + ".conv2(pw)+res", t3, hb, T, D, D, p.W(m + ".conv2.weight"), p.W(m + ".conv2.bias"), 1, hb, 1.0f))
                        add_gemm(p, m + ".conv2(pw)+res", t3, hb, T, D, D, p.W(m + ".conv2.weight"), p.W(m + ".conv2.bias"), ACT_NONE, nullptr, nullptr, hb, 1.0f);
                    ffn(".ff2", 0, false);
                }
                const float *lw = p.W(q + ".layer_norm.weight"), *lb = p.W(q + ".layer_norm.bias");
                // the last block's LayerNorm feeds only the mean over time: one pass for both (NWW_LN_MEAN=0: two launches)
                if (last_fused) {
                } else if (i == nb - 1 && D <= 256) {
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
        const PlanCtx pc = p;                                   // (the DNN body's pointers, by value)
        p.add("tail:" + p.tail_name + "+classifier", [=](Run& r) {
            TailArgs t{src(r, tin), tK, We, be, E, W0, b0, w3, b3, r.emb, r.logits, r.probs, r.B, act};
            if (pc.dnn_body) {
                t.ln0_w = pc.dnn_ln0_w; t.ln0_b = pc.dnn_ln0_b; t.n_mid = pc.dnn_n_mid;
                for (int i = 0; i < pc.dnn_n_mid; ++i) { t.mid_W[i] = pc.dnn_mid[i][0]; t.mid_b[i] = pc.dnn_mid[i][1]; t.mid_lnw[i] = pc.dnn_mid[i][2]; t.mid_lnb[i] = pc.dnn_mid[i][3]; }
            }
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
    if (!h->plan_error.empty()) return fail(h, NWW_ERR_HIP, "%s", h->plan_error.c_str());
    h->finalized = true;
    return NWW_OK;
}

