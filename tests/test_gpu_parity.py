"""HIP path vs the oracle and the reference-generated goldens, through the C-ABI (run with -m gpu)."""
import numpy as np
import pytest

import oracle
from conftest import head_case_names
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.synth import synth_features, synth_pcm, synth_state_dict
from parity import assert_frontend_amplitude, assert_frontend_close, frontend_errors, logit_bounds

pytestmark = pytest.mark.gpu

FEAT_LOGIT_ATOL = 1e-4        # north_star: logits within 1e-4 (float32)
FEAT_EMB_RTOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    from nanowakeword_amd.session import HipModel, HipSession
    return HipModel, HipSession


def _fe_cfg(n_mels, center):
    return FrontendConfig(n_mels=n_mels, center=center)


def _model(hip, cfg, fe, g, n_mels):
    HipModel, _ = hip
    fb = g["fb64"] if n_mels == 64 else g["fb40"]
    return HipModel(cfg, fe, state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=fb)


@pytest.mark.parametrize("variant", ["64c", "40n"])
def test_frontend_vs_reference_golden(hip, golden_frontend, variant):
    g = golden_frontend
    n_mels, center = (64, True) if variant == "64c" else (40, False)
    cfg = HeadConfig("dnn", (101, 64) if center else (98, 40))
    m = _model(hip, cfg, _fe_cfg(n_mels, center), g, n_mels)
    db, mel = m.frontend(g["pcm"], return_power=True)
    mel_ref, db_ref = (g["mel64"], g["db64"]) if center else (g["mel40"], g["db40"])
    assert db.shape == db_ref.shape                      # frame count bit-exact
    e_db, e_mel, frac = assert_frontend_close(mel, db, mel_ref, db_ref, variant)
    print(f"frontend {variant}: max dB err {e_db:.2e} (well-conditioned {frac:.0%} of bins), mel err {e_mel:.2e} x frame peak")
    # and against the oracle (same tables)
    mo = oracle.mel_power(g["pcm"], g["window"], g["fb64"] if center else g["fb40"], center=center)
    assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), variant + "/oracle")
    # criterion C: every bin - the ones A excludes too - against the same graph in float64
    exact = oracle.mel_power(g["pcm"], g["window"], g["fb64"] if center else g["fb40"], center=center, dtype=np.float64)
    k_all, k_exc = assert_frontend_amplitude(mel, exact, variant)
    print(f"frontend {variant}: amplitude error {k_all:.2f} (all bins) / {k_exc:.2f} (bins below 1e-4 x frame peak) x 2^-24 x frame peak amplitude")
    m.close()


def test_frame_law_and_edges(hip, golden_frontend):
    g = golden_frontend
    mc = _model(hip, HeadConfig("dnn", (101, 64)), _fe_cfg(64, True), g, 64)
    mn = _model(hip, HeadConfig("dnn", (98, 40)), _fe_cfg(40, False), g, 40)
    for n, fc, fn in zip(g["edge_n"], g["edge_frames_center"], g["edge_frames_nocenter"]):
        assert mc.num_frames(int(n)) == int(fc) and mn.num_frames(int(n)) == int(fn)
        x = synth_pcm("noise", 2, int(n), seed=77)
        assert mc.frontend(x).shape == (2, 64, int(fc))
        assert mn.frontend(x).shape == (2, 40, int(fn))
    db = mc.frontend(g["short_pcm"])
    assert np.abs(db - g["short_db64"]).max() <= 1e-4
    with pytest.raises(ValueError):
        mn.frontend(np.zeros((1, 399), np.int16))       # shorter than n_fft
    with pytest.raises(ValueError):
        mc.frontend(np.zeros((1, 200), np.int16))       # reflect pad needs N > n_fft/2
    with pytest.raises(ValueError):
        mc.frontend([1, 2, 3])                          # non-ndarray (nanointerpreter.py:628-629)
    # odd length / misaligned clips take the 2-byte staging path: compare with oracle
    x = synth_pcm("noise", 3, 16001, seed=3)
    db, mel = mc.frontend(x, return_power=True)
    mo = oracle.mel_power(x, g["window"], g["fb64"])
    assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), "odd-length")
    assert_frontend_amplitude(mel, oracle.mel_power(x, g["window"], g["fb64"], dtype=np.float64), "odd-length")
    mc.close(); mn.close()


def test_frontend_linearity_and_builtin_tables(hip):
    """Size-independent properties at full batch: |X|^2 scales with gain^2 (6.0206 dB per doubling),
    and the library's built-in double-precision tables stay within 1e-2 dB of
    the torchaudio-float32 tables."""
    HipModel, _ = hip
    cfg = HeadConfig("dnn", (101, 64))
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
    x = synth_pcm("noise", 256, 16000, seed=11) // 4
    a = m.frontend(x)
    b = m.frontend((x * 2).astype(np.int16))
    assert np.abs((b - a) - 20 * np.log10(2.0)).max() <= 2e-4
    m2 = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg), tables="builtin")
    c = m2.frontend(x)
    assert np.abs(c - a).max() <= 1e-2
    m.close(); m2.close()


@pytest.mark.parametrize("name", head_case_names())
def test_head_vs_oracle_and_golden(hip, golden_heads, golden_frontend, name):
    d, meta = golden_heads
    g = golden_frontend
    cfg = HeadConfig(**meta[name])
    sd = synth_state_dict(cfg)
    n_mels = 40 if cfg.input_shape == (98, 40) else 64
    center = n_mels == 64
    m = _model(hip, cfg, _fe_cfg(n_mels, center), g, n_mels)
    # (i) features -> logits / embedding, vs reference golden and oracle
    feats = synth_features(4, cfg.input_shape)
    logits, probs, emb = m.forward_features(feats, return_embedding=True)
    ref = d[f"{name}/logits_feat"].ravel()
    assert np.abs(logits - ref).max() <= FEAT_LOGIT_ATOL, (name, np.abs(logits - ref).max())
    assert np.abs(probs - oracle.sigmoid(ref)).max() <= 1e-5
    e_or = oracle.head_forward(feats, sd, cfg)
    assert np.abs(emb - e_or).max() <= FEAT_EMB_RTOL * max(1.0, np.abs(e_or).max()), np.abs(emb - e_or).max()
    # ragged batch sizes (tile edges of the MFMA kernels): 1, 33, 70 clips
    for B in (1, 33, 70):
        fx = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(fx)
        lo = oracle.model_forward(fx, sd, cfg).ravel()
        assert np.abs(lg - lo).max() <= FEAT_LOGIT_ATOL, (name, B, np.abs(lg - lo).max())
    # (ii) PCM -> logits through the fused frontend, vs the reference composite
    if f"{name}/logits_pcm" in d:
        rp = d[f"{name}/logits_pcm"].ravel()
        lp, pp = m.forward_pcm(g["pcm"])
        fb = g["fb64"] if n_mels == 64 else g["fb40"]
        lm32 = oracle.frontend_logmel(g["pcm"], g["window"], fb, center=center)
        lm64 = oracle.frontend_logmel(g["pcm"], g["window"], fb, center=center, dtype=np.float64).astype(np.float32)
        if cfg.model_type != "e2e_dnn":
            lm32, lm64 = lm32.transpose(0, 2, 1), lm64.transpose(0, 2, 1)
        l32 = oracle.model_forward(np.ascontiguousarray(lm32), sd, cfg).ravel()
        lx = oracle.model_forward(np.ascontiguousarray(lm64), sd, cfg).ravel()
        bound = logit_bounds(g["names"], rp, l32, lx)
        err = np.abs(lp - rp)
        assert np.all(err <= bound), (name, [f"{n}: {e:.2e} > {b:.2e}" for n, e, b in zip(g["names"], err, bound) if e > b])
        print(f"{name}: max |dlogit| vs reference = {err.max():.2e} (broadband/speech clips: "
              f"{err[[i for i, n in enumerate(g['names']) if not str(n).startswith(('sine', 'chirp'))]].max():.2e})")
        assert np.abs(pp - oracle.sigmoid(lp)).max() <= 1e-6
    m.close()


def test_host_pointer_calls_equal_device_pointer_calls(hip, golden_frontend):
    """Small host-pointer calls take the zero-copy path (kernels read the pinned staging buffer, the classifier tail writes into
    pinned memory and a completion word the host polls; B <= 16 arms the word, inputs up to 1 MiB = 32 clips stay zero-copy,
    larger ones go through copy commands): every size must give the device-pointer entry point's logits bit for bit, call after
    call, and the features entry point (with and without the embedding output) likewise."""
    import torch
    HipModel, _ = hip
    g = golden_frontend
    cfg = HeadConfig("cnn", (101, 64))
    m = _model(hip, cfg, _fe_cfg(64, True), g, 64)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    x = synth_pcm("noise", 40, 16000, seed=11)
    xd = torch.from_numpy(x).to(dev)
    want_l = torch.empty(40, dtype=torch.float32, device=dev)
    want_p = torch.empty(40, dtype=torch.float32, device=dev)
    m.forward_pcm_dev(xd.data_ptr(), 40, 16000, want_l.data_ptr(), want_p.data_ptr(), stream)
    torch.cuda.synchronize()
    wl, wp = want_l.cpu().numpy(), want_p.cpu().numpy()
    for B in (1, 2, 16, 17, 32, 33, 40, 1, 16):                # 16 | 17: completion word armed or not; 32 | 33: 1 MiB staging limit
        lg, pr = m.forward_pcm(x[:B])
        assert np.array_equal(lg, wl[:B]) and np.array_equal(pr, wp[:B]), B
    feats = synth_features(20, cfg.input_shape, seed=4)
    l1, p1 = m.forward_features(feats)
    l2, p2, e2 = m.forward_features(feats, return_embedding=True)      # the embedding output keeps the copy path
    assert np.array_equal(l1, l2) and np.array_equal(p1, p2) and e2.shape == (20, cfg.embedding_dim)
    for B in (1, 16, 17):
        lb, _ = m.forward_features(feats[:B])
        assert np.array_equal(lb, l1[:B]), B
    m.close()


def test_session_protocol_and_errors(hip, golden_frontend):
    HipModel, HipSession = hip
    g = golden_frontend
    cfg = HeadConfig("e2e_dnn", (64, 101))
    m = _model(hip, cfg, _fe_cfg(64, True), g, 64)
    s = HipSession(m, mode="e2e", clip_samples=16000, input_ndim=3)
    inp = s.get_inputs()[0]
    assert inp.name == "input" and inp.shape == [None, 1, 16000]
    clip = (g["pcm"][:3].astype(np.float32) / 32768.0).reshape(3, 1, -1)       # nanointerpreter.py:750,773
    out = s.run(None, {"input": clip})
    assert isinstance(out, list) and out[0].shape == (3, 1, 1) and out[0].dtype == np.float32
    lg, pr = m.forward_pcm(g["pcm"][:3])
    assert np.array_equal(out[0].ravel(), pr)                                  # float and int16 inputs agree bit-for-bit
    assert float(out[0][0].item()) == float(pr[0])                             # .item() as the interpreter calls it (:784)
    with pytest.raises(ValueError):
        s.run(None, {"input": clip * 0.3333})                                  # not int16-representable
    with pytest.raises(ValueError):
        m.forward_features(np.zeros((2, 5, 5), np.float32))
    # state errors
    m2 = HipModel(cfg, _fe_cfg(64, True))
    with pytest.raises(KeyError):
        m2.finalize()                                                          # missing state_dict keys
    with pytest.raises(ValueError):
        m2._load("model.fc1.weight", np.zeros((3, 3), np.float32))             # size mismatch
    with pytest.raises(KeyError):
        m2.load_state_dict({"bogus.weight": np.zeros(3, np.float32)})
    m.close(); m2.close()
    with pytest.raises(ValueError):
        HipModel(HeadConfig("dnn", (16, 96)), FrontendConfig(n_mels=500))


def test_batch_invariance_full_size(hip, golden_frontend):
    """At BASELINE size (B=4096) the oracle is too slow to run everywhere; use properties instead:
    per-clip results do not depend on batch size or position in the batch (bit-exact), and the
    first clips equal the small-batch run that IS checked against the oracle."""
    g = golden_frontend
    cfg = HeadConfig("cnn", (101, 64))
    m = _model(hip, cfg, _fe_cfg(64, True), g, 64)
    x = synth_pcm("noise", 4096, 16000)
    lg, pr = m.forward_pcm(x)
    assert np.isfinite(lg).all()
    l16, _ = m.forward_pcm(x[:16])
    assert np.array_equal(lg[:16], l16)
    for nb in (1, 3, 8):                                       # the interpreter-sized calls take shortcuts (four trunk strips, two-frame
        ls, _ = m.forward_pcm(x[:nb])                          # frontend groups, split-K reduce inside the tail): same bits
        assert np.array_equal(lg[:nb], ls), nb
    perm = np.random.default_rng(0).permutation(4096)
    lp, _ = m.forward_pcm(np.ascontiguousarray(x[perm]))
    assert np.array_equal(lp, lg[perm])
    sd = synth_state_dict(cfg)
    lm = oracle.frontend_logmel(x[:8], g["window"], g["fb64"]).transpose(0, 2, 1)
    lo = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
    assert np.abs(lg[:8] - lo).max() <= 1e-4
    m.close()


@pytest.mark.parametrize("d_model,n_head", [(32, 8), (80, 4), (144, 8), (96, 2), (144, 2), (66, 2), (50, 2), (90, 6), (250, 2), (186, 2)])
def test_conformer_attention_head_dims(hip, d_model, n_head):
    """Compiled attention head dims (4, 20, 18, 48, 72) and ones in between / above (33, 25, 15, 125, 93: zero-padded onto the
    next compiled width; d_model itself need not be a multiple of 4) against the oracle."""
    HipModel, _ = hip
    cfg = HeadConfig("conformer", (16, 24), embedding_dim=16, conformer_d_model=d_model, conformer_n_head=n_head)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    feats = synth_features(5, cfg.input_shape, seed=d_model + n_head)
    logits, _ = m.forward_features(feats)
    ref = oracle.model_forward(feats, sd, cfg).ravel()
    assert np.abs(logits - ref).max() <= FEAT_LOGIT_ATOL, np.abs(logits - ref).max()
    m.close()


@pytest.mark.parametrize("d_model,n_head,shape,B", [(32, 2, (16, 32), 5), (64, 4, (40, 64), 7), (96, 4, (101, 64), 3),
                                                    (128, 4, (33, 64), 6), (144, 4, (101, 64), 3), (144, 4, (130, 64), 2)])
def test_conformer_fused_kernels_every_width(hip, d_model, n_head, shape, B):
    """Every compiled width of the round-2 Conformer kernels (ffn_x3 / lin_x3: d_model 32, 64, 96, 128, 144; input_proj
    K = 32 / 64; mha_h2 (mha_mfma under the other arithmetics) head dims 16, 24, 32, 36 and 1..4 key tiles; T = 130 > 128 falls back to the VALU attention
    core) must be IN the plan and agree with the oracle; row counts that are not multiples of the 128-row tile."""
    HipModel, _ = hip
    cfg = HeadConfig("conformer", shape, embedding_dim=16, conformer_d_model=d_model, conformer_n_head=n_head)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    plan = m.describe_plan()
    # default arithmetic: the whole attention module in one launch at d_model 144 / 4 heads and 64 < T <= 128 (attn_x3), else the two-term attention core
    attn = d_model == 144 and n_head == 4 and 64 < shape[0] <= 128
    assert "ffn_x3:" in plan and "lin_x3:" in plan and ("attn_x3:" in plan) == attn and ("mha_h2:" in plan) == (shape[0] <= 128 and not attn), plan
    feats = synth_features(B, cfg.input_shape, seed=d_model + B)
    logits, _ = m.forward_features(feats)
    ref = oracle.model_forward(feats, sd, cfg).ravel()
    assert np.abs(logits - ref).max() <= FEAT_LOGIT_ATOL, np.abs(logits - ref).max()
    m.close()


@pytest.mark.parametrize("T,B", [(65, 3), (80, 5), (96, 2), (97, 4), (101, 300), (112, 3), (113, 2), (128, 7)])
def test_conformer_fused_attention_module(hip, T, B):
    """attn_x3 (in_proj, per-head softmax(q k^T) v, out_proj, residual in one clip-resident launch; architectures.py:471-493, 512-513) at every
    key-block count it takes (5 .. 8 blocks of 16 keys: the run-time instance and the two compiled ones), query tiles that straddle T, and a
    batch larger than the grid (a workgroup walks several clips: the weight-chunk stream and the row prefetch wrap around); against the oracle,
    and a clip's logit must not depend on the batch it travels in."""
    HipModel, _ = hip
    cfg = HeadConfig("conformer", (T, 64), embedding_dim=16)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    assert "attn_x3:" in m.describe_plan(), m.describe_plan()
    feats = synth_features(B, cfg.input_shape, seed=T + B)
    feats[0, T // 2:] = -100.0                              # half a clip of floor values (identical rows: flat scores)
    logits, _ = m.forward_features(feats)
    k = min(B, 6)
    ref = oracle.model_forward(feats[:k], sd, cfg).ravel()
    assert np.isfinite(logits).all()
    assert np.abs(logits[:k] - ref).max() <= FEAT_LOGIT_ATOL, np.abs(logits[:k] - ref).max()
    l1, _ = m.forward_features(feats[:1])
    l3, _ = m.forward_features(feats[B - 1:])
    assert np.array_equal(l1, logits[:1]) and np.array_equal(l3, logits[B - 1:])
    m.close()


def test_conformer_unsupported_head_dim_is_refused_at_create(hip):
    """a head wider than the widest compiled attention kernel (128) is refused when the model is created, not at the first batch"""
    HipModel, _ = hip
    cfg = HeadConfig("conformer", (16, 24), embedding_dim=16, conformer_d_model=260, conformer_n_head=2)
    with pytest.raises(NotImplementedError, match="head_dim 130"):
        HipModel(cfg, FrontendConfig())


@pytest.mark.parametrize("shape", [(201, 64), (150, 96), (303, 40)])
def test_cnn_trunk_row_strips_for_large_inputs(hip, shape):
    """Inputs whose conv1 output does not fit one CU's LDS go through the fused trunk in row strips (seam rows
    recomputed by both neighbours); results must not depend on where the seams fall."""
    HipModel, _ = hip
    cfg = HeadConfig("cnn", shape, embedding_dim=16)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    assert "trunk" in m.describe_plan()
    feats = synth_features(3, shape, seed=shape[0])
    logits, _, emb = m.forward_features(feats, return_embedding=True)
    e_or = oracle.head_forward(feats, sd, cfg)
    assert np.abs(emb - e_or).max() <= FEAT_EMB_RTOL * max(1.0, np.abs(e_or).max())
    assert np.abs(logits - oracle.model_forward(feats, sd, cfg).ravel()).max() <= FEAT_LOGIT_ATOL
    m.close()


def test_conv_arithmetic_modes_agree(hip, golden_frontend):
    """The fused trunk and fc1 on the float32 MFMA, on the bf16 MFMA from exactly split operands with nine (exact products) or
    six partial products, or on the f16 MFMA from two binary16 terms per operand (three partial products, the default): all
    four are float32-grade - each sits as close to the oracle as the others, and every mode is deterministic and batch invariant."""
    HipModel, _ = hip
    g = golden_frontend
    cfg = HeadConfig("cnn", (101, 64))
    sd = synth_state_dict(cfg)
    feats = synth_features(40, cfg.input_shape, seed=77)
    ref = oracle.model_forward(feats, sd, cfg).ravel()
    out = {}
    for mode in ("f32", "bf16x9", "bf16x6", "f16x3", None):
        m = HipModel(cfg, _fe_cfg(64, True), state_dict=sd, window=g["window"], mel_fb=g["fb64"], conv_arith=mode)
        assert ("trunk_x3:" in m.describe_plan()) == (mode != "f32")
        assert ("[f16x3]" in m.describe_plan()) == (mode in ("f16x3", None))      # the library default is f16x3
        lg, _ = m.forward_features(feats)
        lg2, _ = m.forward_features(feats[::-1].copy())
        assert np.array_equal(lg, lg2[::-1]), mode                     # batch position does not matter, bit for bit
        lg3, _ = m.forward_features(feats[:7])
        assert np.array_equal(lg[:7], lg3), mode
        out[mode] = lg
        m.close()
    err = {k: float(np.abs(v - ref).max()) for k, v in out.items()}
    print("max |dlogit| vs oracle:", err)
    assert max(err.values()) <= 2e-5                                   # 5x tighter than the 1e-4 parity bar
    assert max(err.values()) <= 3 * min(err.values()) + 2e-6           # no mode is meaningfully worse than another
    assert np.abs(out["bf16x6"] - out["f32"]).max() <= 2e-5 and np.abs(out["bf16x9"] - out["f32"]).max() <= 2e-5
    assert np.abs(out["f16x3"] - out["f32"]).max() <= 2e-5 and np.array_equal(out["f16x3"], out[None])
    with pytest.raises(ValueError):
        HipModel(cfg, _fe_cfg(64, True), conv_arith="fp8")


@pytest.mark.parametrize("shape", [(37, 28), (64, 64), (100, 100), (17, 130), (12, 12), (41, 18), (256, 32)])
@pytest.mark.parametrize("arith", ["f16x3", "bf16x6", "f32"])
def test_cnn_trunk_odd_shapes(hip, shape, arith):
    """Edge geometry of the fused trunk: odd sizes (floor pooling drops a row/column), widths that leave partial
    MFMA tiles, strips of unequal height, tiny and tall inputs - in both arithmetics."""
    HipModel, _ = hip
    cfg = HeadConfig("cnn", shape, embedding_dim=16, activation="silu" if shape[0] % 2 else "relu")
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith=arith)
    feats = synth_features(5, shape, seed=shape[0] * 7 + shape[1])
    logits, _, emb = m.forward_features(feats, return_embedding=True)
    e_or = oracle.head_forward(feats, sd, cfg)
    assert np.abs(emb - e_or).max() <= FEAT_EMB_RTOL * max(1.0, np.abs(e_or).max()), (shape, arith)
    assert np.abs(logits - oracle.model_forward(feats, sd, cfg).ravel()).max() <= FEAT_LOGIT_ATOL
    m.close()


_F64_HEADS = (("cnn", (101, 64), {}), ("dnn", (98, 40), {}), ("crnn", (101, 64), {}), ("crnn", (101, 64), {"crnn_rnn_type": "lstm"}),
              ("e2e_dnn", (101, 64), {}), ("conformer", (101, 64), {}), ("gru", (101, 64), {}), ("bcresnet", (101, 64), {}))


def _arith_errors_vs_float64(HipModel, cfg, sd, feats, modes=("f32", "bf16x9", "bf16x6", "f16x3")):
    ref = oracle.model_forward(feats, sd, cfg, dtype=np.float64).ravel()
    err = {}
    for mode in modes:
        m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[1]), state_dict=sd, conv_arith=mode)
        lg, _ = m.forward_features(feats)
        assert np.isfinite(lg).all(), (cfg.model_type, mode)
        err[mode] = float(np.abs(lg.astype(np.float64) - ref).max())
        m.close()
    return err, float(np.abs(ref).max())


_F64_IDS = [h + ("-" + "-".join(map(str, k.values())) if k else "") for h, _, k in _F64_HEADS]


@pytest.mark.parametrize("head,shape,kw", _F64_HEADS, ids=_F64_IDS)
def test_arithmetic_modes_against_float64(hip, head, shape, kw):
    """Every arithmetic against the same network evaluated in float64 (oracle, dtype = float64), EVERY head (VERDICT r04 item 2:
    f16x3 is the default everywhere, so its float64 evidence runs where the driver runs it), 48 clips: the two-term binary16 form
    (default) and the three-term bf16 forms are as close to exact arithmetic as the float32 MFMA path - none is more than
    2x + 2e-6 worse than it."""
    HipModel, _ = hip
    cfg = HeadConfig(head, shape, **kw)
    sd = synth_state_dict(cfg)
    feats = synth_features(48, cfg.input_shape, seed=21)
    err, _ = _arith_errors_vs_float64(HipModel, cfg, sd, feats)
    print(head, kw, "max |dlogit| vs float64:", err)
    # every mode 10x inside the 1e-4 bar; the BcResNet head's own float32 noise (ten layers, no normalisation of the residual
    # stream) is 2.3e-5 in EVERY mode, the float32 MFMA path included: 3x inside
    assert max(err.values()) <= (3e-5 if head == "bcresnet" else 1e-5), (head, err)
    for mode in ("bf16x9", "bf16x6", "f16x3"):
        assert err[mode] <= 2.0 * err["f32"] + 2e-6, (head, mode, err)


def _heavy_tailed(sd, factor, frac=0.005, seed=77):
    """Trained-model-like outliers: `frac` of every contraction weight tensor (ndim >= 2) times `factor` - the case where a
    plan-time scale that keeps the LARGEST weight inside binary16 pushes the `lo` terms of the ordinary weights towards
    binary16's subnormals."""
    out = {}
    for k, v in sd.items():
        v = np.array(v, np.float32, copy=True)
        if v.ndim >= 2 and v.size >= 200:
            r = np.random.default_rng([seed, len(k), v.size])
            idx = r.choice(v.size, max(1, int(frac * v.size)), replace=False)
            rms = float(np.sqrt(np.mean(v.astype(np.float64) ** 2)))
            v.reshape(-1)[idx] *= np.float32(factor)
            # same RMS as before (activations stay O(1) through the deep heads): the ordinary weights now sit `factor` below the
            # tensor's largest, which is what the plan-time scale is derived from
            v *= np.float32(rms / float(np.sqrt(np.mean(v.astype(np.float64) ** 2))))
        out[k] = v
    return out


@pytest.mark.parametrize("factor", [2.0 ** 12, 2.0 ** 20])
@pytest.mark.parametrize("head,shape,kw", _F64_HEADS, ids=_F64_IDS)
def test_heavy_tailed_weights_against_float64(hip, head, shape, kw, factor):
    """0.5 % of every weight tensor x 2^12 / x 2^20 (outlier weights of a trained model), default arithmetic against float64,
    beside the float32 MFMA path on the same weights: relative to the largest logit the two-term form is no more than 2x + 2e-6
    worse (where a layer's bound / typical-magnitude window says the two-term form would lose bits the plan keeps it on bf16x6 -
    nww_plan.hip F16Range - and this test is what holds that guard to account)."""
    HipModel, _ = hip
    cfg = HeadConfig(head, shape, **kw)
    sd = _heavy_tailed(synth_state_dict(cfg), factor)
    feats = synth_features(24, cfg.input_shape, seed=22)
    err, scale = _arith_errors_vs_float64(HipModel, cfg, sd, feats, modes=("f32", "f16x3"))
    rel = {k: v / max(1.0, scale) for k, v in err.items()}
    print(head, kw, factor, "max |dlogit| / max(1, |logit|max) vs float64:", rel, "scale", scale)
    assert rel["f16x3"] <= 2.0 * rel["f32"] + 2e-6, (head, factor, rel)
    assert rel["f16x3"] <= 1e-4, (head, factor, rel)


def test_bcresnet_shapes_activations_and_storage(hip):
    """The BcResNet head's round-4 kernels (two-term front kernel, blocks chained with the next depthwise in bc_chain.hip, two-term
    dual_x3 with per-pixel scales) on shapes whose planes are ragged against the 32-pixel work items, with every activation, odd
    batch sizes (fewer clips than persistent workgroups, and more), against the oracle: float32 storage at the head tolerance,
    binary16 storage (act_dtype = f16) at 1e-2.  Weight scales 2^-10 .. 2^10 per layer (the plan-time scales move with them) change
    nothing beyond float32 rounding."""
    HipModel, _ = hip
    for shape, kw in (((101, 64), {}), ((48, 40), {"activation": "gelu"}), ((61, 36), {"activation": "silu"}), ((32, 40), {}),
                      ((200, 64), {}), ((101, 80), {"activation": "gelu"})):
        cfg = HeadConfig("bcresnet", shape, **kw)
        sd = synth_state_dict(cfg)
        for B in (5, 300):
            feats = synth_features(B, cfg.input_shape, seed=B)
            ref = oracle.model_forward(feats[:24], sd, cfg).ravel()
            for act_dtype, tol in ((None, FEAT_LOGIT_ATOL), ("f16", 1e-2)):
                m = HipModel(cfg, FrontendConfig(n_mels=shape[1]), state_dict=sd, act_dtype=act_dtype)
                lg, _ = m.forward_features(feats)
                d = float(np.abs(lg[:24] - ref).max())
                assert d <= tol, (shape, kw, B, act_dtype, d, m.describe_plan())
                m.close()
    cfg = HeadConfig("bcresnet", (101, 64))
    base = synth_state_dict(cfg)
    feats = synth_features(6, cfg.input_shape, seed=9)
    # negative BatchNorm factors in the init conv: the front kernel's pooling then needs the window's minimum too (its plan-time
    # shortcut for all-non-negative factors must not be taken)
    sd = {k: v.copy() for k, v in base.items()}
    sd["model.init_conv.1.weight"][::3] *= np.float32(-1.0)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    lg, _ = m.forward_features(feats)
    d = float(np.abs(lg - oracle.model_forward(feats, sd, cfg).ravel()).max())
    assert d <= FEAT_LOGIT_ATOL, d
    m.close()
    ref = oracle.model_forward(feats, base, cfg).ravel()
    for e0, e1, e2, e3 in ((10, -10, 0, 0), (-10, 0, 10, 0), (0, 10, 0, -10), (4, 4, -4, -4)):
        sd = {k: v.copy() for k, v in base.items()}
        # scaling a conv's weights in front of a BatchNorm is absorbed by scaling the BN statistics: the network's function is unchanged
        for name, bn, e in (("model.init_conv.0", "model.init_conv.1", e0), ("model.block1.pointwise", "model.block1.bn1", e1),
                            ("model.block2.shortcut.0", "model.block2.shortcut.1", e2), ("model.block3.pointwise", "model.block3.bn1", e3)):
            f = np.float32(2.0 ** e)
            sd[name + ".weight"] = (sd[name + ".weight"] * f).astype(np.float32)
            sd[bn + ".running_mean"] = (sd[bn + ".running_mean"] * f).astype(np.float32)
            sd[bn + ".running_var"] = ((sd[bn + ".running_var"] + np.float32(1e-5)) * f * f - np.float32(1e-5)).astype(np.float32)
        for act_dtype, tol in ((None, 2 * FEAT_LOGIT_ATOL), ("f16", 1e-2)):
            m = HipModel(cfg, FrontendConfig(), state_dict=sd, act_dtype=act_dtype)
            lg, _ = m.forward_features(feats)
            d = float(np.abs(lg - ref).max())
            assert d <= tol, ((e0, e1, e2, e3), act_dtype, d)
            m.close()


def test_f16x3_scales_clamp_and_activations(hip):
    """The two-term binary16 arithmetic (nww_config.conv_arith = NWW_ARITH_F16X3, the default): power-of-two scales fixed at plan
    time from bounds on the tensors.  (i) every activation / BatchNorm form of the fused trunk (CNN: bias only; CRNN / E2E: folded
    BN) against the oracle; (ii) weights scaled by 2^-12 .. 2^12 (the bounds, hence the scales, move with them) change nothing
    beyond float32 rounding; (iii) features far outside the log-mel range stay finite - inputs are clamped to +-8192, the bound
    the scales were derived from - and features inside it are not touched by the clamp."""
    HipModel, _ = hip
    for head, shape, kw in (("cnn", (101, 64), {}), ("cnn", (40, 36), {"activation": "gelu"}), ("cnn", (44, 40), {"activation": "silu"}),
                            ("crnn", (101, 64), {}), ("crnn", (48, 40), {"activation": "gelu"}), ("e2e_dnn", (64, 101), {}),
                            ("e2e_dnn", (40, 61), {"activation": "silu"}), ("dnn", (98, 40), {}), ("dnn", (16, 96), {})):
        cfg = HeadConfig(head, shape, **kw)
        sd = synth_state_dict(cfg)
        feats = synth_features(9, cfg.input_shape, seed=5)
        m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith="f16x3")
        assert "[f16x3]" in m.describe_plan(), m.describe_plan()
        lg, _ = m.forward_features(feats)
        ref = oracle.model_forward(feats, sd, cfg).ravel()
        assert np.abs(lg - ref).max() <= FEAT_LOGIT_ATOL, (head, shape, kw, float(np.abs(lg - ref).max()))
        m.close()
    cfg = HeadConfig("cnn", (101, 64))
    base = synth_state_dict(cfg)
    feats = synth_features(6, cfg.input_shape, seed=9)
    for e1, e2, e3 in ((-12, 0, 12), (12, -12, 0), (0, 12, -12), (6, 6, -12)):
        sd = {k: v.copy() for k, v in base.items()}
        for name, e in (("model.conv1", e1), ("model.conv2", e2), ("model.fc1", e3)):       # total scale 2^(e1 + e2 + e3) = 1 on the logits
            sd[name + ".weight"] = (sd[name + ".weight"] * np.float32(2.0 ** e)).astype(np.float32)
        sd["model.conv1.bias"] = (base["model.conv1.bias"] * np.float32(2.0 ** e1)).astype(np.float32)
        sd["model.conv2.bias"] = (base["model.conv2.bias"] * np.float32(2.0 ** (e1 + e2))).astype(np.float32)
        sd["model.fc1.bias"] = (base["model.fc1.bias"] * np.float32(2.0 ** (e1 + e2 + e3))).astype(np.float32)
        m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith="f16x3")
        lg, _ = m.forward_features(feats)
        ref = oracle.model_forward(feats, sd, cfg).ravel()
        assert np.abs(lg - ref).max() <= FEAT_LOGIT_ATOL, ((e1, e2, e3), float(np.abs(lg - ref).max()))
        m.close()
    # (iv) a layer whose plan-time bound is out of all proportion to its typical magnitude (here: a bias of 1e7 on conv1) is not
    # given to the two-term form - the bound would push the values that matter below binary16's precision - and runs on three bf16 terms
    sd = {k: v.copy() for k, v in base.items()}
    sd["model.conv1.bias"] = (sd["model.conv1.bias"] + np.float32(1e7)).astype(np.float32)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith="f16x3")
    assert "trunk_x3:" in m.describe_plan() and "[f16x3]" not in m.describe_plan().split("gemm:")[0], m.describe_plan()
    lg, _ = m.forward_features(feats)
    ref = oracle.model_forward(feats, sd, cfg, dtype=np.float64).ravel()
    assert np.abs(lg - ref).max() <= 1e-5 * np.abs(ref).max(), float(np.abs(lg - ref).max() / np.abs(ref).max())
    m.close()
    m = HipModel(cfg, FrontendConfig(), state_dict=base, conv_arith="f16x3")
    wild = feats.copy()
    wild[0] *= 1e6; wild[1, 3, 5] = 3e38; wild[2, :, 0] = -1e9
    lg, _ = m.forward_features(wild)
    assert np.isfinite(lg).all()
    ref = oracle.model_forward(np.clip(wild, -8192.0, 8192.0), base, cfg).ravel()
    assert np.abs(lg - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), float(np.abs(lg - ref).max())
    assert np.array_equal(lg[3:], m.forward_features(feats)[0][3:])                                   # untouched clips: bit-identical
    m.close()


def test_split_operand_gemm_on_every_shape_subprocess():
    """NWW_GEMM_X3=2 routes EVERY Linear / 1x1 conv with N, K >= 32 through the split-operand GEMM (normally only
    long-K layers use it).  The knob is read once per process, hence the subprocess.  Heads with many different
    (M, N, K): DNN (K = 6464), Conformer (K = 144 / 576, N = 144 .. 576), GRU (K = 64, N = 384), BcResNet (1x1 convs)."""
    import os, subprocess, sys
    script = r'''
import numpy as np, oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict
worst = 0.0
for cfg in (HeadConfig("dnn", (101, 64)), HeadConfig("conformer", (40, 32), embedding_dim=16, conformer_d_model=96, conformer_n_head=4),
            HeadConfig("gru", (30, 64), layer_dim=64), HeadConfig("bcresnet", (32, 40), embedding_dim=16)):
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    for B in (3, 70):
        x = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(x)
        worst = max(worst, float(np.abs(lg - oracle.model_forward(x, sd, cfg).ravel()).max()))
    m.close()
print("WORST", worst)
assert worst <= 1e-4, worst
'''
    env = dict(os.environ, NWW_GEMM_X3="2", PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "WORST" in r.stdout


def test_feature_clamp_is_surfaced(hip):
    """The default arithmetic clamps the head input to +-8192 (the reference does not): nww_feature_clamp says so, the host layer warns
    when a feature array crosses it, and conv_arith = bf16x6 (no clamp) follows the oracle on the same outliers."""
    import warnings
    HipModel, _ = hip
    cfg = HeadConfig("dnn", (16, 96))
    sd = synth_state_dict(cfg)
    feats = synth_features(6, cfg.input_shape, seed=3)
    feats[2, 5, 7] = 3.0e4
    feats[4, 0, 0] = -1.0e5
    ref = oracle.model_forward(feats, sd, cfg).ravel()
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    assert m.feature_clamp == 8192.0
    with pytest.warns(RuntimeWarning, match="clamps the head input"):
        lg, _ = m.forward_features(feats)
    ok = [0, 1, 3, 5]
    assert np.abs(lg[ok] - ref[ok]).max() <= FEAT_LOGIT_ATOL          # clips inside the bound are untouched
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.forward_features(feats[ok])                                  # no warning without outliers
    m.close()
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith="bf16x6")
    assert m.feature_clamp == 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lg, _ = m.forward_features(feats)
    assert np.abs(lg - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), np.abs(lg - ref)
    m.close()
    # heads whose first layer scales every row by its own power of two clamp nothing, whatever their step names say
    for kw in (dict(model_type="conformer", input_shape=(16, 24), embedding_dim=16, conformer_d_model=32, conformer_n_head=2),
               dict(model_type="gru", input_shape=(12, 64), layer_dim=256)):
        m = HipModel(HeadConfig(**kw), FrontendConfig(), state_dict=synth_state_dict(HeadConfig(**kw)))
        assert m.feature_clamp == 0.0, (kw, m.describe_plan())
        m.close()
    for kw in (dict(model_type="cnn", input_shape=(101, 64)), dict(model_type="gru", input_shape=(101, 64)), dict(model_type="bcresnet", input_shape=(32, 40), embedding_dim=16)):
        m = HipModel(HeadConfig(**kw), FrontendConfig(), state_dict=synth_state_dict(HeadConfig(**kw)))
        assert m.feature_clamp == 8192.0, (kw, m.describe_plan())
        m.close()
