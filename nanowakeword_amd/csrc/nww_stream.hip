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

// dense [B][n] <- the first n floats of every clip's slot (stride floats apart): the window of a log-mel ring for heads whose first
// kernel wants dense clips
__global__ void __launch_bounds__(256) stream_gather_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int B, int n4, size_t stride4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * n4) return;
    const int b = (int)(idx / n4), i = (int)(idx - (size_t)b * n4);
    dst[idx] = src[(size_t)b * stride4 + i];
}

static void free_inc(nww_handle* h) {
    if (h->d_lm_ring) (void)hipFree(h->d_lm_ring);
    if (h->d_a2_ring) (void)hipFree(h->d_a2_ring);
    for (int q = 0; q < 2; ++q) { if (h->d_seq[q]) (void)hipFree(h->d_seq[q]); h->d_seq[q] = nullptr; }
    h->seq_cur = 0; h->a3_hi = -1;
    h->d_lm_ring = nullptr; h->d_a2_ring = nullptr;
    h->inc_fe = h->inc_conv = h->primed = false;
    h->lm_rows = h->lm_pos = h->lm_shift = h->a2_rows = h->a2_pos = h->a2_shift = 0;
}

extern "C" int nww_stream_close(nww_handle* h) {
    if (!h) return NWW_ERR_INVALID;
    (void)hipSetDevice(h->cfg.device);
    if (h->d_ring) (void)hipFree(h->d_ring);
    if (h->d_chunk) (void)hipFree(h->d_chunk);
    free_inc(h);
    h->d_ring = nullptr; h->d_chunk = nullptr; h->ring_S = h->ring_W = h->ring_hop = h->ring_pos = 0; h->ring_filled = 0;
    return NWW_OK;
}

// What a hop invalidates.  A hop of k = hop / hop_length frames turns frame t of the old window into frame t - k of the new one, bit
// for bit, unless the frame touches the reflect padding of either window (the first fe_edge_l and last fe_edge_r frames of a centred
// window): per hop only frames [0, fe_edge_l) and [T - fe_edge_r - k, T) are computed, into a ring of log-mel rows per stream.
// Likewise pooled row j of the fused trunk (input rows 4j - 3 .. 4j + 6) equals row j + k / 4 of the previous window when those
// rows are shifted interior frames in both windows: rows [a2_lo, a2_hi] are reused, the rest recomputed from the log-mel ring.
// And pooled row j of the third conv (input rows 8j - 7 .. 8j + 14) equals row j + k / 8 of the previous hop: its sequence output
// alternates between two per-stream buffers, rows [a3_lo, a3_hi] copied over, the others computed.
// NWW_STREAM_INC = 0: every hop re-scores the whole window (rounds 1-3), 1: frontend only, 2: + the fused trunk's rows, 3 (default):
// + the third conv's rows.
static int plan_incremental(nww_handle* h, int S, int W, int hop) {
    static const int mode = [] { const char* e = getenv("NWW_STREAM_INC"); return e ? atoi(e) : 3; }();
    const nww_config& c = h->cfg;
    const int T = fe_num_frames(h->fe, W), hl = h->fe.hop;
    const bool frames_major = !c.mel_major_features || h->e2e_transposed;
    static const int mel_env = 2;
    if (mode <= 0 || !frames_major || !fe2_subset_supported(h->fe, mel_env) || hop % hl != 0) return NWW_OK;
    const int k = hop / hl, pad = h->fe.center ? h->fe.n_fft / 2 : 0;
    int el = 0, er = 0;
    while (el < T && el * hl - pad < 0) ++el;
    while (er < T && (T - 1 - er) * hl - pad + h->fe.n_fft > W) ++er;
    if (T - er - k <= el) return NWW_OK;                      // a hop replaces (nearly) the whole window: nothing to keep
    h->fe_edge_l = el; h->fe_edge_r = er; h->lm_shift = k;
    h->lm_rows = (T + k - 1) / k * k;
    HIP_TRY(h, hipMalloc(&h->d_lm_ring, (size_t)S * 2 * h->lm_rows * c.n_mels * sizeof(float) + 16));
    h->inc_fe = true;
    if (mode >= 2 && h->stream_conv && h->stream_H == T && (k % 4) == 0) {
        const int H2 = T / 4, W2 = h->stream_W / 4;
        const int lo = (el + 3 + 3) / 4, hi = (T - 1 - er - k - 6) >= 0 ? (T - 1 - er - k - 6) / 4 : -1;
        // The rows a hop recomputes - [0, lo) and (hi, H2) - go to the fused trunk as explicit strips, and a strip must fit in LDS:
        // each range is cut into the fewest equal pieces that do (at most four strips in all, TrunkArgs::sub_a).  A window / hop pair
        // that would need more keeps the frontend ring only and re-scores the conv rows (hops of 68-84 frames at W = 16000
        // asked for a 165-198 KB strip and every hop after the first full window failed).
        int nsub = 0, sa[4], sb[4];
        bool fits = hi >= lo && hi < H2;
        auto cut = [&](int a, int b) {
            if (a >= b || !fits) return;
            for (int n = 1; n <= 4; ++n) {
                bool ok = nsub + n <= 4 && n <= b - a;
                for (int q = 0; q < n && ok; ++q) ok = trunk_b_rows_fit(T, h->stream_W, a + (b - a) * q / n, a + (b - a) * (q + 1) / n);
                if (ok) {
                    for (int q = 0; q < n; ++q) { sa[nsub] = a + (b - a) * q / n; sb[nsub] = a + (b - a) * (q + 1) / n; ++nsub; }
                    return;
                }
            }
            fits = false;
        };
        cut(0, lo);
        cut(hi + 1, H2);
        if (fits) {
            h->a2_nsub = nsub;
            for (int q = 0; q < nsub; ++q) { h->a2_sub_a[q] = sa[q]; h->a2_sub_b[q] = sb[q]; }
            h->a2_lo = lo; h->a2_hi = hi; h->a2_shift = k / 4; h->a2_rows = H2;
            HIP_TRY(h, hipMalloc(&h->d_a2_ring, (size_t)S * 32 * H2 * W2 * sizeof(float) + 16));
            h->inc_conv = true;
            // pooled rows of the third conv (row j: input rows 8j - 7 .. 8j + 14) carried from hop to hop in its sequence layout
            const int H3 = H2 / 2, lo3 = (el + 7 + 7) / 8, hi3 = (T - 1 - er - k - 14) >= 0 ? (T - 1 - er - k - 14) / 8 : -1;
            if (mode >= 3 && h->stream_seq && (k % 8) == 0 && hi3 >= lo3 && hi3 < H3) {
                for (int q = 0; q < 2; ++q) HIP_TRY(h, hipMalloc(&h->d_seq[q], (size_t)S * h->seq_floats * sizeof(float) + 16));
                h->a3_lo = lo3; h->a3_hi = hi3; h->a3_shift = k / 8;
            }
        }
    }
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
    rc = plan_incremental(h, S, W, hop);
    if (rc) return rc;
    return ensure_ws(h, S, W);
}

extern "C" int nww_stream_reset(nww_handle* h) {
    if (!h || !h->d_ring) return fail(h, NWW_ERR_STATE, "no open stream batch");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemset(h->d_ring, 0, (size_t)h->ring_S * 2 * h->ring_W * sizeof(int16_t)));
    h->ring_pos = 0; h->ring_filled = 0;
    h->primed = false; h->lm_pos = 0; h->a2_pos = 0; h->seq_cur = 0;      // the next full window is computed whole
    return NWW_OK;
}

extern "C" int64_t nww_stream_filled(const nww_handle* h) { return h ? h->ring_filled : 0; }

// One hop on the incremental path: the invalidated frames (all of them for the first full window) -> log-mel ring -> head.
static int stream_hop_incremental(nww_handle* h, float* d_logits, float* d_probs, hipStream_t s) {
    const nww_config& c = h->cfg;
    const int S = h->ring_S, W = h->ring_W, T = fe_num_frames(h->fe, W), k = h->lm_shift;
    int rc = ensure_ws(h, S, W);
    if (rc) return rc;
    prof_begin(h);
    prof_mark(h, s, 0);
    Fe2Sub sub;
    sub.ring_rows = h->lm_rows; sub.row0 = h->lm_pos; sub.out_clip_stride = (size_t)2 * h->lm_rows * c.n_mels;
    if (h->primed) {
        if (h->fe_edge_l > 0) { sub.t0[sub.nr] = 0; sub.t1[sub.nr] = h->fe_edge_l; ++sub.nr; }
        // the new interior frames, then the trailing edge frames as a group of their own (groups of up to eight frames)
        sub.t0[sub.nr] = T - h->fe_edge_r - k; sub.t1[sub.nr] = T - h->fe_edge_r; ++sub.nr;
        if (h->fe_edge_r > 0) { sub.t0[sub.nr] = T - h->fe_edge_r; sub.t1[sub.nr] = T; ++sub.nr; }
    }
    rc = frontend_dev(h, h->d_ring + h->ring_pos, S, W, h->d_lm_ring, nullptr, 1, s, nullptr, (size_t)2 * W, &sub);
    if (rc) return rc;
    const float* win = h->d_lm_ring + (size_t)h->lm_pos * c.n_mels;          // rows [lm_pos, lm_pos + T) of every stream's ring: its window
    const bool conv_inc = h->inc_conv && h->x_stride_ok;
    if (h->x_stride_ok) {
        h->sr.on = true; h->sr.x_stride = sub.out_clip_stride; h->sr.mode = conv_inc ? (h->primed ? 2 : 1) : 0;
        rc = run_head(h, win, S, d_logits, d_probs, s, nullptr, 0, nullptr, false);
    } else {
        const int n4 = T * c.n_mels / 4;
        const size_t total = (size_t)S * n4;
        hipLaunchKernelGGL(stream_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(win),
                           reinterpret_cast<float4*>(h->d_logmel), S, n4, sub.out_clip_stride / 4);
        HIP_TRY(h, hipGetLastError());
        rc = run_head(h, h->d_logmel, S, d_logits, d_probs, s, nullptr, 0, nullptr, c.mel_major_features != 0);
    }
    if (rc) return rc;
    h->lm_pos = (h->lm_pos + k) % h->lm_rows;
    if (h->inc_conv) h->a2_pos = (h->a2_pos + h->a2_shift) % h->a2_rows;
    if (h->d_seq[0]) h->seq_cur ^= 1;
    h->primed = true;
    return NWW_OK;
}

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
    if (h->inc_fe) {
        const int rc = stream_hop_incremental(h, d_logits, d_probs, s);
        if (rc) h->primed = false;            // the rings may hold a half-finished hop: the next hop computes its window whole
        return rc;
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

