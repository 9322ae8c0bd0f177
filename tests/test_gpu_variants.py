"""GPU parity for everything that is not the default path of a default-config head (run with -m gpu):
round-2 reference goldens (CRNN with the reference's default LSTM backend, other conv stacks and recurrent widths,
the native E2E composite at 1.5 s / 2 s clips), every fallback / variant kernel behind an environment knob, the C4
streaming configuration at full size, and batch invariance at the per-GPU sizes of BASELINE configs 3 and 5."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import head_case_names_r02, head_case_names_r04, head_case_names_r06
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.synth import synth_features, synth_pcm, synth_state_dict
from parity import logit_bounds

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def HipModel():
    from nanowakeword_amd.session import HipModel
    return HipModel


@pytest.mark.parametrize("name", head_case_names_r02())
def test_round2_heads_vs_reference(HipModel, golden_heads_r02, golden_frontend, name):
    d, meta = golden_heads_r02
    g = golden_frontend
    cfg = HeadConfig(**meta[name])
    sd = synth_state_dict(cfg)
    n_mels = 40 if cfg.input_shape == (98, 40) else 64
    fe = FrontendConfig(n_mels=n_mels, center=n_mels == 64)
    m = HipModel(cfg, fe, state_dict=sd, window=g["window"], mel_fb=g["fb64"] if n_mels == 64 else g["fb40"])
    if cfg.model_type == "e2e_dnn":
        # clip length 24 000 / 32 000: general export-form average pool (a17), through the PCM entry point
        pcm = d[f"{name}/pcm"]
        lp, pp = m.forward_pcm(pcm)
        rp = d[f"{name}/logits_pcm"].ravel()
        lm32 = oracle.frontend_logmel(pcm, g["window"], g["fb64"])
        lm64 = oracle.frontend_logmel(pcm, g["window"], g["fb64"], dtype=np.float64).astype(np.float32)
        bound = logit_bounds(["noise0", "noise1", "speechlike0", "speechlike1", "loud0", "zeros0"], rp,
                             oracle.model_forward(lm32, sd, cfg).ravel(), oracle.model_forward(lm64, sd, cfg).ravel())
        assert np.all(np.abs(lp - rp) <= bound), (name, np.abs(lp - rp), bound)
        assert np.all(np.abs(pp - d[f"{name}/probs_pcm_export"].ravel()) <= bound)
        m.close()
        return
    feats = synth_features(4, cfg.input_shape)
    logits, probs, emb = m.forward_features(feats, return_embedding=True)
    ref = d[f"{name}/logits_feat"].ravel()
    assert np.abs(logits - ref).max() <= 1e-4, (name, np.abs(logits - ref).max())
    e_ref = d[f"{name}/emb_feat"]
    assert np.abs(emb - e_ref).max() <= 1e-4 * max(1.0, np.abs(e_ref).max())
    for B in (1, 17, 50):                                   # ragged batches: partial 16- / 32-clip recurrent workgroups
        fx = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(fx)
        assert np.abs(lg - oracle.model_forward(fx, sd, cfg).ravel()).max() <= 1e-4, (name, B)
    if f"{name}/logits_pcm" in d:
        rp = d[f"{name}/logits_pcm"].ravel()
        lp, _ = m.forward_pcm(g["pcm"])
        lm32 = oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"]).transpose(0, 2, 1)
        lm64 = oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"], dtype=np.float64).astype(np.float32).transpose(0, 2, 1)
        bound = logit_bounds(g["names"], rp, oracle.model_forward(np.ascontiguousarray(lm32), sd, cfg).ravel(),
                             oracle.model_forward(np.ascontiguousarray(lm64), sd, cfg).ravel())
        assert np.all(np.abs(lp - rp) <= bound), (name, np.abs(lp - rp), bound)
    m.close()


# Every knob is read once per process, hence one subprocess per setting.  Each runs the heads whose plan the knob changes
# against the oracle (features) and, for the frontend knobs, against the reference goldens.
_KNOB_SCRIPT = r'''
import json, os, sys
import numpy as np, oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict
heads, want_in_plan, not_in_plan, check_fe = json.loads(sys.argv[1])
worst = 0.0
for spec in heads:
    cfg = HeadConfig(**spec)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, **({"conv_arith": os.environ["TEST_CONV_ARITH"]} if "TEST_CONV_ARITH" in os.environ else {}))
    plan = m.describe_plan()
    for w in want_in_plan:
        assert w in plan, (w, plan)
    for w in not_in_plan:
        assert w not in plan, (w, plan)
    for B in (3, 40):
        x = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(x)
        worst = max(worst, float(np.abs(lg - oracle.model_forward(x, sd, cfg).ravel()).max()))
    m.close()
if check_fe:
    sys.path.insert(0, os.path.join(os.environ["NWW_ROOT"], "tests"))
    from parity import assert_frontend_amplitude, assert_frontend_close
    g = dict(np.load(os.path.join(os.environ["NWW_ROOT"], "tests", "golden", "frontend.npz")))
    for n_mels, center, mk, dk, fk in ((64, True, "mel64", "db64", "fb64"), (40, False, "mel40", "db40", "fb40")):
        cfg = HeadConfig("dnn", (101, 64) if center else (98, 40))
        m = HipModel(cfg, FrontendConfig(n_mels=n_mels, center=center), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g[fk])
        db, mel = m.frontend(g["pcm"], return_power=True)
        assert_frontend_close(mel, db, g[mk], g[dk], "knob")
        assert_frontend_amplitude(mel, oracle.mel_power(g["pcm"], g["window"], g[fk], center=center, dtype=np.float64), "knob")
        lg, _ = m.forward_pcm(g["pcm"])                     # frames-major output path of the same kernel
        lm = np.ascontiguousarray(oracle.frontend_logmel(g["pcm"], g["window"], g[fk], center=center).transpose(0, 2, 1))
        assert np.abs(lg - oracle.model_forward(lm, synth_state_dict(cfg), cfg).ravel())[:4].max() <= 1e-4
        m.close()
print("WORST", worst)
assert worst <= 1e-4, worst
'''

_CNN = dict(model_type="cnn", input_shape=(101, 64))
_CRNN = dict(model_type="crnn", input_shape=(101, 64))
_CRNN4 = dict(model_type="crnn", input_shape=(32, 96), crnn_cnn_channels=[16, 32, 64, 64])
_CRNN_LSTM = dict(model_type="crnn", input_shape=(16, 96), crnn_rnn_type="lstm")
_E2E = dict(model_type="e2e_dnn", input_shape=(64, 101))
_GRU = dict(model_type="gru", input_shape=(30, 64), layer_dim=64)
_BC = dict(model_type="bcresnet", input_shape=(32, 40), embedding_dim=16)
_CONF = dict(model_type="conformer", input_shape=(40, 32), embedding_dim=16, conformer_d_model=96, conformer_n_head=4)


@pytest.mark.parametrize("env,heads,want,unwanted,check_fe", [
    ({"NWW_TRUNK": "0"}, [_CNN, _CRNN, _E2E], [], ["trunk"], False),                    # unfused conv1 / conv2 kernels
    ({"NWW_CONV_MFMA": "0"}, [_CRNN, _CRNN4, _E2E, _BC], [], ["conv3x3_mfma", "conv1_mfma"], False),   # VALU 3x3 convs
    ({"TEST_CONV_ARITH": "f32"}, [_CNN, _E2E], ["trunk:"], ["trunk_x3"], False),        # float32-MFMA fused trunk (nww_config.conv_arith)
    ({"TEST_CONV_ARITH": "f32"}, [_CRNN, _GRU, _CRNN_LSTM, dict(model_type="gru", input_shape=(30, 64), layer_dim=96), dict(model_type="gru", input_shape=(12, 64), layer_dim=256), _BC, _CONF],
     [], ["[f16x3]", "dual_x3"], False),                                                 # ... the float32 recurrent kernels (register-resident and streaming), float32-MFMA BcResNet dual GEMM, Conformer GEMMs
    ({"TEST_CONV_ARITH": "bf16x9"}, [_CNN, _E2E, _CRNN], ["trunk_x3"], ["[f16x3]"], False),      # all nine partial products
    ({"TEST_CONV_ARITH": "bf16x6"}, [_CNN, _E2E, _CRNN], ["trunk_x3"], ["[f16x3]"], False),      # three bf16 terms, six partial products (the default of rounds 2-3)
    ({"TEST_CONV_ARITH": "bf16x6"}, [_GRU, _CRNN_LSTM, _CRNN4, dict(model_type="gru", input_shape=(30, 64), layer_dim=96), dict(model_type="gru", input_shape=(12, 64), layer_dim=256),
                                     dict(model_type="gru", input_shape=(101, 64)), _BC], [], ["[f16x3]", "streamed"], False),   # ... the three-term forms of the recurrences (four waves of 32 units; widths beside 32 / 64 / 128 on the 32-clips-per-workgroup kernels), k-split conv passes, BcResNet front / block products
    ({"TEST_CONV_ARITH": "bf16x6"}, [dict(model_type="dnn", input_shape=(98, 40))], ["gemm:layer1"], ["[f16x3]"], False),   # ... DNN layer1
    ({"NWW_CONV3_X3": "0"}, [_CRNN, _E2E], ["conv3x3_mfma"], ["conv3_x3"], False),      # float32-MFMA third conv stage
    ({"NWW_FFN_FUSED": "0"}, [_CONF], ["layernorm:", "linear1+swish"], ["ffn_x3"], False),   # feed-forward as LayerNorm + two GEMMs
    ({"NWW_MHA_MFMA": "0"}, [_CONF], ["mha_core"], ["mha_mfma", "mha_h2", "head-major"], False),  # one-lane-per-query attention core
    ({}, [_GRU, dict(model_type="gru", input_shape=(101, 64)), dict(model_type="gru", input_shape=(20, 32), layer_dim=32)], ["+ input projection [f16x3]"], ["lin_x3:model.gru.ih_l0 "], False),   # (default) ... fused into the recurrence
    ({}, [_CRNN4, dict(model_type="crnn", input_shape=(96, 64), crnn_cnn_channels=[16, 32, 64, 64]), dict(model_type="crnn", input_shape=(101, 64), crnn_cnn_channels=[16, 32, 64, 32], activation="gelu")],
     ["conv3_x3:model.cnn.12 (input channels 0-31, raw sums)", "conv3_x3+seq:model.cnn.12 (input channels 32-63, epilogue)"], ["conv3x3:model.cnn.12"], False),   # (default) a fourth CRNN stage with 64 input channels: two passes of the 32-channel instance
    ({}, [dict(model_type="crnn", input_shape=(64, 64), crnn_cnn_channels=[16, 32, 96, 32]), dict(model_type="crnn", input_shape=(48, 96), crnn_cnn_channels=[16, 32, 64, 128], crnn_rnn_type="lstm")],
     ["epilogue)"], ["conv3x3:"], False),                                                # ... three passes (96 input channels); 128 output channels
    ({}, [dict(model_type="crnn", input_shape=(151, 64)), dict(model_type="crnn", input_shape=(201, 64), crnn_rnn_type="lstm", activation="gelu"), dict(model_type="crnn", input_shape=(150, 96), layer_dim=64)],
     ["conv3_x3+seq:model.cnn.8 (strips of"], ["conv3x3:"], False),                      # clips longer than ~1.3 s: the third stage's plane (> 512 pixels) in strips of rows
    ({}, [dict(model_type="crnn", input_shape=(101, 64), crnn_cnn_channels=[32, 64, 64]), dict(model_type="crnn", input_shape=(61, 40), crnn_cnn_channels=[32, 32])],
     ["conv3_x3", "(strips of"], ["conv3x3:model.cnn.4", "conv3x3:model.cnn.8"], False),   # stacks the fused trunk does not take (first stage not 16 channels): later stages in strips / k-split passes
    ({}, [dict(model_type="crnn", input_shape=(201, 64), crnn_cnn_channels=[32, 64, 64], layer_dim=64), dict(model_type="crnn", input_shape=(151, 96), crnn_cnn_channels=[16, 32, 64, 64], crnn_rnn_type="lstm")],
     ["raw sums)", "(strips of"], ["conv3x3:model.cnn.4", "conv3x3:model.cnn.8", "conv3x3:model.cnn.12"], False),   # k-split passes that also run in strips (a 64-channel stage on an 800-pixel plane)
    ({"NWW_F16_RANGE_LOG2": "40"}, [_CRNN4, dict(model_type="crnn", input_shape=(96, 64), crnn_cnn_channels=[16, 32, 64, 64], activation="silu")],
     ["conv3_x3+seq:model.cnn.12 [f16x3]"], ["raw sums", "conv3x3:"], False),            # ... the wide two-term instance (taken when the plan-time bound of the stage's input is tight enough)
    ({}, [dict(model_type="gru", input_shape=(30, 64), layer_dim=96), dict(model_type="gru", input_shape=(20, 32), layer_dim=20, n_blocks=2),
          dict(model_type="crnn", input_shape=(16, 96), crnn_rnn_type="lstm", layer_dim=48), dict(model_type="crnn", input_shape=(32, 96), layer_dim=100)],
     ["+ first reverse step [f16x3]"], [], False),                                    # (default) recurrent widths between the register-resident ones: zero-padded instances
    ({}, [dict(model_type="gru", input_shape=(12, 64), layer_dim=256), dict(model_type="gru", input_shape=(10, 40), layer_dim=160, n_blocks=2),
          dict(model_type="crnn", input_shape=(32, 96), crnn_rnn_type="lstm", layer_dim=200)],
     ["+ first reverse step [f16x3, W_hh streamed]"], [], False),                        # (default) recurrent widths above 128: W_hh streamed from L2 every step (rnn_stream.hip)
    ({"NWW_ATTN_FUSED": "0"}, [dict(model_type="conformer", input_shape=(101, 64)), dict(model_type="conformer", input_shape=(70, 40), embedding_dim=16)],
     ["in_proj(head-major)", "mha_h2:", "out_proj+res"], ["attn_x3"], False),                # the attention module as three launches (in_proj, core, out_proj + residual) at the fused kernel's shape
    ({}, [dict(model_type="conformer", input_shape=(101, 64)), dict(model_type="conformer", input_shape=(70, 40), embedding_dim=16, n_blocks=2)],
     ["attn_x3:"], ["mha_h2:", ".attention.in_proj", ".attention.out_proj"], False),                                          # (default) ... in one clip-resident launch (attn_x3.hip)
    ({"TEST_CONV_ARITH": "bf16x6"}, [_CONF], ["mha_mfma", "ffn_x3"], ["[f16x3]"], False),   # Conformer on three bf16 terms
    ({"NWW_LIN_X3": "0"}, [_CONF], ["glu:", "gemm:input_proj"], ["lin_x3"], False),     # short-K Linears on the general GEMM
    ({"NWW_BC_FRONT": "0"}, [_BC], ["conv1_mfma:init_conv", "dwconv3x3_nhwc:model.block1"], ["conv1_dw_mfma", "conv1_dw_x3"], False),   # init conv and block1 depthwise apart
    ({"NWW_BC_FRONT": "2"}, [_BC], ["conv1_dw_mfma"], ["conv1_dw_x3"], False),          # fused front kernel on the float32 MFMA
    ({"TEST_CONV_ARITH": "bf16x9"}, [_BC], ["conv1_dw_x3"], [], False),                 # fused front kernel, all nine partial products
    ({"NWW_BC_CHAIN": "0"}, [_BC], ["dwconv3x3_nhwc:model.block2", "dwconv3x3_nhwc:model.block3", "[f16x3]"], ["bc_chain"], False),   # blocks unchained
    ({"NWW_GEMM_X3": "0"}, [_CNN], [], [], False),                                      # fc1 on the float32-MFMA GEMM
    ({"NWW_TAIL": "0"}, [_CNN, _GRU], [], ["tail:"], False),                            # separate GEMMs + sigmoid instead of the fused tail
], ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else None)
def test_knob_variants_in_subprocess(env, heads, want, unwanted, check_fe):
    import json
    e = dict(os.environ, NWW_ROOT=ROOT, PYTHONPATH=os.pathsep.join([ROOT] + sys.path), **env)
    r = subprocess.run([sys.executable, "-c", _KNOB_SCRIPT, json.dumps([heads, want, unwanted, check_fe])], env=e,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "WORST" in r.stdout


def test_c4_streaming_full_size(HipModel, golden_frontend):
    """BASELINE config 4 at full size: 1024 lock-step streams x 125 hops of 80 ms (10 s) on the CRNN-GRU head.
    Properties over all streams (the oracle cannot follow 128 000 window scores): duplicated streams score
    identically whatever their slot, scores are finite probabilities, nothing fires before the window is full; and
    three hops of eight streams are checked against the oracle on the exact 1 s windows."""
    from nanowakeword_amd.interpreter import StreamBatch
    g = golden_frontend
    cfg = HeadConfig("crnn", (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    S, hop, n_hops = 1024, 1280, 125
    base = np.stack([synth_pcm("speechlike" if s % 2 else "noise", 1, hop * n_hops, seed=100 + s)[0] for s in range(64)])
    idx = np.random.default_rng(4).integers(0, 64, S)
    idx[:64] = np.arange(64)
    streams = base[idx]                                       # every distinct stream appears in several slots
    sb = StreamBatch(m, S, 16000, hop)
    raw = np.zeros((n_hops, S), np.float32)
    for i in range(n_hops):
        sb.push(np.ascontiguousarray(streams[:, i * hop:(i + 1) * hop]))
        raw[i] = sb.raw_scores
    assert np.isfinite(raw).all() and (raw >= 0).all() and (raw <= 1).all()
    assert not raw[:12].any() and raw[12:].all()              # 12.5 hops fill the 1 s window
    for s in range(64, S):
        assert np.array_equal(raw[:, s], raw[:, idx[s]]), s   # slot independence, bit for bit
    for i in (12, 60, 124):                                   # first full window, mid-stream, last hop
        end = (i + 1) * hop
        win = streams[:8, end - 16000:end]
        lm = oracle.frontend_logmel(win, g["window"], g["fb64"]).transpose(0, 2, 1)
        want = oracle.sigmoid(oracle.model_forward(np.ascontiguousarray(lm), sd, cfg)).ravel()
        assert np.abs(raw[i, :8] - want).max() <= 1e-4, (i, np.abs(raw[i, :8] - want).max())
    sb.close(); m.close()


def test_bcresnet_bf16_batch_invariance_at_baseline_size(HipModel, golden_frontend):
    """BASELINE config 3 as written (bf16 activations) at its per-GPU batch of 8192: a clip's logit does not depend on the batch it
    travels in or on its slot (bit-exact), and stays within the mode's tolerance of the float32 path on broadband clips."""
    g = golden_frontend
    cfg = HeadConfig("bcresnet", (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"], act_dtype="bf16")
    B = 8192
    x = synth_pcm("noise", B, 16000, seed=7)
    lg, _ = m.forward_pcm(x)
    assert np.isfinite(lg).all()
    l16, _ = m.forward_pcm(x[:16])
    assert np.array_equal(lg[:16], l16)
    perm = np.random.default_rng(0).permutation(B)
    lp, _ = m.forward_pcm(np.ascontiguousarray(x[perm]))
    assert np.array_equal(lp, lg[perm])
    m32 = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    l32, _ = m32.forward_pcm(x[:64])
    assert np.abs(lg[:64] - l32).max() <= 2e-2, float(np.abs(lg[:64] - l32).max())
    m.close(); m32.close()


@pytest.mark.parametrize("head,B", [("bcresnet", 8192), ("conformer", 2048)])
def test_batch_invariance_at_baseline_sizes(HipModel, golden_frontend, head, B):
    """BASELINE configs 3 (BcResNet, 65 536 / 8 GPUs) and 5 (Conformer, 16 384 / 8 GPUs) at their per-GPU batch:
    a clip's logit does not depend on the batch size or its position (bit-exact), and the first clips equal the
    small batch that IS checked against the oracle."""
    g = golden_frontend
    cfg = HeadConfig(head, (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    x = synth_pcm("noise", B, 16000, seed=7)
    lg, _ = m.forward_pcm(x)
    assert np.isfinite(lg).all()
    l16, _ = m.forward_pcm(x[:16])
    assert np.array_equal(lg[:16], l16)
    perm = np.random.default_rng(1).permutation(B)
    lp, _ = m.forward_pcm(np.ascontiguousarray(x[perm]))
    assert np.array_equal(lp, lg[perm])
    lm = oracle.frontend_logmel(x[:6], g["window"], g["fb64"]).transpose(0, 2, 1)
    lo = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
    assert np.abs(lg[:6] - lo).max() <= 1e-4
    m.close()


def test_capi_communicator_world1(HipModel, golden_frontend):
    """RCCL through the C-ABI (nww_comm_*): a one-rank communicator on the 1-GPU box exercises the run-time binding, the
    by-value 128-byte id and the in-place all-gather on the kernels' stream; the gathered vector must be the logits."""
    import torch
    g = golden_frontend
    cfg = HeadConfig("cnn", (101, 64))
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g["fb64"])
    uid = HipModel.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    m.comm_init(0, 1, uid)
    dev = torch.device("cuda", 0)
    x = synth_pcm("noise", 64, 16000, seed=2)
    pcm = torch.from_numpy(x).to(dev)
    out = torch.zeros(64, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    m.forward_pcm_gather_dev(pcm.data_ptr(), 64, 16000, out.data_ptr(), stream)
    torch.cuda.synchronize()
    want, _ = m.forward_pcm(x)
    assert np.array_equal(out.cpu().numpy(), want)
    send = torch.arange(10, dtype=torch.float32, device=dev)
    recv = torch.zeros(10, dtype=torch.float32, device=dev)
    m.all_gather_logits_dev(send.data_ptr(), recv.data_ptr(), 10, stream)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    # the asynchronous form: the gather runs on the handle's own stream behind an event, two buffers in flight; the gathered
    # vectors are the same bits, and the previous step's gather ENDS AFTER the next step was free to start (it is off the kernels' path)
    outs = [torch.zeros(64, dtype=torch.float32, device=dev) for _ in range(2)]
    for k in range(6):                                        # back to back: the host runs ahead of the device, as in a serving loop
        m.forward_pcm_gather_async_dev(pcm.data_ptr(), 64, 16000, outs[k & 1].data_ptr(), stream)
    m.gather_fence(stream)
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), want) and np.array_equal(outs[1].cpu().numpy(), want)
    # a one-rank in-place all-gather is a no-op, so the gather is made visible: a 300 us spin on the gather's stream in front of it
    # (test hook).  The next step's start event must then come BEFORE the previous step's gather ends - by about the spin.
    # (the hook is read once per communicator, at nww_comm_init: a fresh one-rank communicator is made with it set)
    m.comm_destroy()
    os.environ["NWW_GATHER_TEST_DELAY_US"] = "300"
    try:
        m.comm_init(0, 1, HipModel.comm_unique_id())
    finally:
        del os.environ["NWW_GATHER_TEST_DELAY_US"]
    for k in range(4):
        m.forward_pcm_gather_async_dev(pcm.data_ptr(), 64, 16000, outs[k & 1].data_ptr(), stream)
    overlap = m.gather_overlap_ms()
    # out of phase on purpose: the same buffer twice in a row must still come out right (each call waits for the gather that last wrote it)
    m.forward_pcm_gather_async_dev(pcm.data_ptr(), 64, 16000, outs[1].data_ptr(), stream)
    m.forward_pcm_gather_async_dev(pcm.data_ptr(), 64, 16000, outs[1].data_ptr(), stream)
    m.gather_fence(stream)
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), want) and np.array_equal(outs[1].cpu().numpy(), want)
    print(f"the next step started {overlap:.3f} ms before the previous step's (delayed) gather ended")
    assert overlap > 0.1, overlap
    with pytest.raises(ValueError):
        m.comm_init(1, 1, uid)                                # rank outside the world
    m.comm_destroy()
    with pytest.raises(Exception, match="communicator"):
        m.forward_pcm_gather_dev(pcm.data_ptr(), 64, 16000, out.data_ptr(), stream)
    m.close()


def test_capi_gather_two_ranks():
    """The C-ABI RCCL path with a real second rank (rank > 0 slot offset, cross-rank agreement): needs two GPUs, skipped on
    the 1-GPU box (the driver's multi-GPU node runs it)."""
    import socket
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_capi_gather_two_ranks_one_device():
    """the rank > 0 slot offset with BOTH ranks on cuda:0 of the 1-GPU box, if RCCL allows a communicator
    with two ranks on one device.  It does not (ncclCommInitRank refuses duplicate devices): the test then records that answer
    instead of the two-rank result - the real second rank stays with test_capi_gather_two_ranks on the driver's multi-GPU node."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--one-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if "one-device communicator refused" in r.stdout:
        pytest.skip("RCCL refuses two ranks on one device: " + [l for l in r.stdout.splitlines() if "refused" in l][0][-160:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_recurrent_kernels_every_size_and_arithmetic(HipModel):
    """rnn_x3.hip (split-operand recurrent product, H = 32 / 64 / 128, 6 and 9 partial products) and the float32-MFMA
    recurrences behind conv_arith = f32, GRU head (T = 30 steps, two directions) and CRNN with the LSTM backend, against the
    oracle; batches that leave the last 16-clip workgroup ragged.  Each arithmetic is batch invariant."""
    worst = 0.0
    for H in (32, 64, 128):
        for cfg in (HeadConfig("gru", (30, 64), layer_dim=H), HeadConfig("crnn", (16, 96), layer_dim=H, crnn_rnn_type="lstm"),
                    HeadConfig("crnn", (32, 64), layer_dim=H)):
            sd = synth_state_dict(cfg)
            x = synth_features(37, cfg.input_shape, seed=H)
            want = oracle.model_forward(x, sd, cfg).ravel()
            for arith in (None, "f32", "bf16x9"):
                m = HipModel(cfg, FrontendConfig(), state_dict=sd, **({"conv_arith": arith} if arith else {}))
                names = m.describe_plan()
                assert ("gru:" in names) or ("lstm:" in names), names
                lg, _ = m.forward_features(x)
                worst = max(worst, float(np.abs(lg - want).max()))
                assert np.abs(lg - want).max() <= 1e-4, (H, cfg.model_type, arith, float(np.abs(lg - want).max()))
                l5, _ = m.forward_features(x[:5])
                assert np.array_equal(l5, lg[:5]), (H, cfg.model_type, arith)
                m.close()
    print("worst |dlogit| over recurrent instances:", worst)


def test_bcresnet_front_kernel_odd_shapes(HipModel):
    """bc_front_b_kernel away from (101, 64): widths that are not multiples of four (scalar staging path), more than 32 pooled
    columns (two 32-pixel conv groups per row), planes smaller than one strip; six and nine partial products; against the oracle."""
    for shape in ((33, 42), (50, 70), (21, 18), (64, 101)):
        cfg = HeadConfig("bcresnet", shape, embedding_dim=16)
        sd = synth_state_dict(cfg)
        x = synth_features(5, cfg.input_shape, seed=shape[0])
        want = oracle.model_forward(x, sd, cfg).ravel()
        for arith in (None, "bf16x9"):
            m = HipModel(cfg, FrontendConfig(), state_dict=sd, **({"conv_arith": arith} if arith else {}))
            assert "conv1_dw_x3" in m.describe_plan(), m.describe_plan()
            lg, _ = m.forward_features(x)
            assert np.abs(lg - want).max() <= 1e-4, (shape, arith, float(np.abs(lg - want).max()))
            m.close()


def test_wide_recurrent_layers_are_refused_loudly(HipModel):
    """layer_dim > 512 has no recurrent kernel: nww_create must say so (not fail at the first launch)."""
    for mt in ("gru", "crnn"):
        with pytest.raises(Exception, match="512"):
            HipModel(HeadConfig(mt, (16, 96), layer_dim=520), FrontendConfig())


@pytest.mark.parametrize("mt,rnn,H,nb,shape", [("gru", "gru", 384, 1, (16, 96)), ("gru", "gru", 37, 2, (20, 40)), ("crnn", "lstm", 260, 1, (32, 64)),
                                               ("crnn", "gru", 512, 1, (16, 96)), ("crnn", "lstm", 6, 2, (16, 96)), ("crnn", "lstm", 512, 1, (16, 32))])
def test_any_recurrent_width(HipModel, mt, rnn, H, nb, shape):
    """nn.GRU / nn.LSTM take any hidden_size (architectures.py:132-145,238-254): widths that are not a multiple of 4 or exceed 256
    run on the any-width kernel (zero-padded W_hh rows, two hidden-unit tiles per wave) - against the oracle, ragged batches."""
    cfg = HeadConfig(mt, shape, layer_dim=H, n_blocks=nb, crnn_rnn_type=rnn, embedding_dim=16)
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(n_mels=min(shape[1], 64)), state_dict=sd)
    for B in (1, 5, 33, 70):
        x = synth_features(B, cfg.input_shape, seed=H + B)
        lg, _ = m.forward_features(x)
        want = oracle.model_forward(x, sd, cfg).ravel()
        assert np.abs(lg - want).max() <= 1e-4, (mt, rnn, H, B, float(np.abs(lg - want).max()))
    m.close()


def test_bcresnet_bf16_activations(HipModel, golden_frontend):
    """BASELINE config 3 as written: nww_config.act_dtype = bf16 stores every activation tensor between the BcResNet head's
    kernels as bf16 (float32 products and accumulation).  Tolerance 2e-2 on logits (SURVEY section 7 / BASELINE.md) against
    the float32 oracle on all 16 golden clips through the PCM path and on synthetic features; the default float32 path is
    untouched (bit-identical before and after a bf16 model was created) and other heads refuse the option."""
    g = golden_frontend
    cfg = HeadConfig("bcresnet", (101, 64))
    sd = synth_state_dict(cfg)
    m32 = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    l32, _ = m32.forward_pcm(g["pcm"])
    mbf = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"], act_dtype="bf16")
    assert "bf16" in mbf.describe_plan()
    lbf, pbf = mbf.forward_pcm(g["pcm"])
    lm = np.ascontiguousarray(oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"]).transpose(0, 2, 1))
    ref = oracle.model_forward(lm, sd, cfg).ravel()
    from parity import is_tonal
    tonal = is_tonal(g["names"])
    e = np.abs(lbf - ref)
    print("bcresnet bf16 activations, |dlogit| per clip:", {str(n): float(f"{v:.2e}") for n, v in zip(g["names"], e)},
          f"(float32 path: {np.abs(l32 - ref).max():.2e})")
    # broadband and real-speech clips: 2e-2 (observed <= 6.8e-3).  Two kinds of input are outside what 8 significant bits can hold to
    # 2e-2 and are bounded separately: the three synthetic pure-tone / chirp clips, on which this head's logit
    # already moves by 7.7e-3 under a re-association of float32 sums (observed <= 3.2e-2 since the blocks are chained - the depthwise
    # reads unrounded planes - 4.7e-2 before; bound 6e-2), and digital silence, whose log-mel is -100 dB in every bin: every pixel rounds
    # the same way (observed 0.079, 0.36 in round 3; bound 0.15).  act_dtype = "f16" is the mode without such carve-outs.
    silent = np.array([str(n).startswith("zeros") for n in g["names"]])
    assert e[~tonal & ~silent].max() <= 2e-2, e
    assert e[tonal].max() <= 6e-2, e
    assert e[silent].max() <= 0.15, e
    assert np.abs(pbf - 1.0 / (1.0 + np.exp(-lbf.astype(np.float64)))).max() <= 1e-6
    for B in (3, 40):
        x = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = mbf.forward_features(x)
        assert np.abs(lg - oracle.model_forward(x, sd, cfg).ravel()).max() <= 5e-2      # synthetic N(0, 1)-scale features: observed 3.1e-2
    # batch invariance holds in this mode too
    lb1, _ = mbf.forward_pcm(g["pcm"][:1])
    assert np.array_equal(lb1, lbf[:1])
    l32b, _ = m32.forward_pcm(g["pcm"])
    assert np.array_equal(l32, l32b)
    mbf.close(); m32.close()
    with pytest.raises(Exception, match="BcResNet"):
        HipModel(HeadConfig("cnn", (101, 64)), FrontendConfig(), act_dtype="bf16")


_CHAIN_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, os.environ["NWW_ROOT"])
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
out = {}
for shape, n in (((101, 64), 16000), ((32, 40), None)):
    for act_dtype in (None, "f16", "bf16"):
        cfg = HeadConfig("bcresnet", shape)
        m = HipModel(cfg, FrontendConfig(n_mels=shape[1]), state_dict=synth_state_dict(cfg), act_dtype=act_dtype)
        plan = m.describe_plan()
        assert ("bc_chain" in plan) == (os.environ.get("NWW_BC_CHAIN", "1") != "0"), plan
        if n:
            lg, _ = m.forward_pcm(synth_pcm("speechlike", 300, n, seed=3))
        else:
            from nanowakeword_amd.synth import synth_features
            lg, _ = m.forward_features(synth_features(37, shape, seed=2))
        out[f"{shape[0]}_{act_dtype}"] = lg
        m.close()
np.savez(sys.argv[1], **out)
"""


def test_bcresnet_chained_blocks_equal_unchained(tmp_path):
    """bc_chain.hip keeps a block's output in LDS and runs the next block's depthwise on it; the same arithmetic in the same order as
    dual_x3 + dwconv3x3_nhwc, so the float32 logits are bit-identical with the chain switched off (NWW_BC_CHAIN = 0), at the
    BASELINE shape (13 / 8 work items per clip, ragged last pixel group) and at a small one; the f16-activation logits (whose unchained
    path rounds the block output to binary16 before the depthwise, the chained one after) agree to 1e-2 (observed 6e-3 on N(0,1)-scale
    features, whose first-layer activations are far from the log-mel statistics the scales assume)."""
    res = {}
    for chain in ("1", "0"):
        out = str(tmp_path / f"chain{chain}.npz")
        e = dict(os.environ, NWW_ROOT=ROOT, PYTHONPATH=os.pathsep.join([ROOT] + sys.path), NWW_BC_CHAIN=chain)
        r = subprocess.run([sys.executable, "-c", _CHAIN_SCRIPT, out], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[chain] = dict(np.load(out))
    for k in res["1"]:
        print(k, float(np.abs(res["1"][k] - res["0"][k]).max()))
        if k.endswith("None"):
            assert np.array_equal(res["1"][k], res["0"][k]), (k, float(np.abs(res["1"][k] - res["0"][k]).max()))
        elif k.endswith("bf16"):
            assert np.abs(res["1"][k] - res["0"][k]).max() <= 0.15, (k, float(np.abs(res["1"][k] - res["0"][k]).max()))
        else:
            assert np.abs(res["1"][k] - res["0"][k]).max() <= 1e-2, (k, float(np.abs(res["1"][k] - res["0"][k]).max()))


def test_bcresnet_f16_activations(HipModel, golden_frontend):
    """nww_config.act_dtype = f16: the tensors between the BcResNet head's kernels as binary16 of value x a plan-time power of two
    (same bytes as bf16, 11 significant bits instead of 8; block weights as two binary16 terms, float32 accumulation).  Unlike bf16
    the mode needs no carve-outs: 1e-2 against the float32 oracle on ALL 16 golden clips - tones, chirp and digital silence
    included (CPU emulation of the storage rounding alone: <= 4.2e-3) - and on synthetic features; batch-invariant; out-of-range
    features saturate instead of producing infinities; the float32 path is untouched."""
    g = golden_frontend
    cfg = HeadConfig("bcresnet", (101, 64))
    sd = synth_state_dict(cfg)
    m32 = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    l32, _ = m32.forward_pcm(g["pcm"])
    mh = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"], act_dtype="f16")
    plan = mh.describe_plan()
    assert "f16 out" in plan and plan.count("(f16 activations)") == 3 and "global_avg_pool" in plan, plan
    lh, ph = mh.forward_pcm(g["pcm"])
    lm = np.ascontiguousarray(oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"]).transpose(0, 2, 1))
    ref = oracle.model_forward(lm, sd, cfg).ravel()
    e = np.abs(lh - ref)
    print("bcresnet f16 activations, |dlogit| per clip:", {str(n): float(f"{v:.2e}") for n, v in zip(g["names"], e)},
          f"(float32 path: {np.abs(l32 - ref).max():.2e})")
    assert e.max() <= 1e-2, e
    assert np.abs(ph - 1.0 / (1.0 + np.exp(-lh.astype(np.float64)))).max() <= 1e-6
    for B in (3, 40):
        x = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = mh.forward_features(x)
        d = np.abs(lg - oracle.model_forward(x, sd, cfg).ravel()).max()
        assert d <= 5e-3, d
    lb1, _ = mh.forward_pcm(g["pcm"][:1])
    assert np.array_equal(lb1, lh[:1])
    # features far outside the assumed +-8192: finite (saturated) logits, no infinities / NaNs
    big = synth_features(4, cfg.input_shape, seed=1) * 1e6
    lg, _ = mh.forward_features(big)
    assert np.isfinite(lg).all()
    l32b, _ = m32.forward_pcm(g["pcm"])
    assert np.array_equal(l32, l32b)
    mh.close(); m32.close()
    with pytest.raises(Exception, match="BcResNet"):
        HipModel(HeadConfig("cnn", (101, 64)), FrontendConfig(), act_dtype="f16")


def test_bcresnet_f16_batch_invariance_at_baseline_size(HipModel, golden_frontend):
    """act_dtype = f16 at BASELINE config 3's per-GPU batch of 8192 (the eight-wave kernels): slot- and batch-independent logits,
    within 5e-3 of the float32 path."""
    g = golden_frontend
    cfg = HeadConfig("bcresnet", (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"], act_dtype="f16")
    B = 8192
    x = synth_pcm("noise", B, 16000, seed=7)
    lg, _ = m.forward_pcm(x)
    assert np.isfinite(lg).all()
    l16, _ = m.forward_pcm(x[:16])
    assert np.array_equal(lg[:16], l16)
    perm = np.random.default_rng(0).permutation(B)
    lp, _ = m.forward_pcm(np.ascontiguousarray(x[perm]))
    assert np.array_equal(lp, lg[perm])
    m32 = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    l32, _ = m32.forward_pcm(x[:64])
    assert np.abs(lg[:64] - l32).max() <= 5e-3, float(np.abs(lg[:64] - l32).max())
    m.close(); m32.close()


R04_REFUSED = set()          # every shape of heads_r04.npz runs (round 4: any recurrent width <= 512, any attention head dim <= 128)


@pytest.mark.parametrize("name", head_case_names_r04())
def test_round4_heads_vs_reference(HipModel, golden_heads_r04, golden_frontend, name):
    """The reference's distilled lite gate (distill.py:45-76: DNN 8 / 1 / 8), DNN inputs with K % 4 != 0 (the split-K / VALU seam),
    recurrent widths and attention head dims beyond the first kernel set - against goldens made by the reference classes."""
    d, meta = golden_heads_r04
    g = golden_frontend
    cfg = HeadConfig(**meta[name])
    sd = synth_state_dict(cfg)
    n_mels = 40 if cfg.input_shape == (98, 40) else 64
    fe = FrontendConfig(n_mels=n_mels, center=n_mels == 64)
    if name in R04_REFUSED:
        with pytest.raises(NotImplementedError):
            HipModel(cfg, fe, state_dict=sd, window=g["window"], mel_fb=g["fb64"] if n_mels == 64 else g["fb40"])
        return
    m = HipModel(cfg, fe, state_dict=sd, window=g["window"], mel_fb=g["fb64"] if n_mels == 64 else g["fb40"])
    feats = synth_features(4, cfg.input_shape)
    logits, probs, emb = m.forward_features(feats, return_embedding=True)
    ref = d[f"{name}/logits_feat"].ravel()
    assert np.abs(logits - ref).max() <= 1e-4, (name, np.abs(logits - ref).max())
    assert np.abs(probs - oracle.sigmoid(ref)).max() <= 1e-5
    e_ref = d[f"{name}/emb_feat"]
    assert np.abs(emb - e_ref).max() <= 1e-4 * max(1.0, np.abs(e_ref).max())
    for B in (1, 9, 33, 130):                               # ragged batches; B <= 8 and > 8 take different split-K reduce paths
        fx = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(fx)
        assert np.abs(lg - oracle.model_forward(fx, sd, cfg).ravel()).max() <= 1e-4, (name, B)
        l1, _ = m.forward_features(fx[:1])
        assert l1[0] == lg[0], (name, B, "batch dependence")
    if f"{name}/logits_pcm" in d and cfg.input_shape in ((101, 64), (98, 40)):
        lp, _ = m.forward_pcm(g["pcm"])
        rp = d[f"{name}/logits_pcm"].ravel()
        center = n_mels == 64
        fb = g["fb64"] if center else g["fb40"]
        lm32 = oracle.frontend_logmel(g["pcm"], g["window"], fb, n_mels=n_mels, center=center).transpose(0, 2, 1)
        lm64 = oracle.frontend_logmel(g["pcm"], g["window"], fb, n_mels=n_mels, center=center, dtype=np.float64).astype(np.float32).transpose(0, 2, 1)
        bound = logit_bounds([str(n) for n in g["names"]], rp, oracle.model_forward(np.ascontiguousarray(lm32), sd, cfg).ravel(),
                             oracle.model_forward(np.ascontiguousarray(lm64), sd, cfg).ravel())
        assert np.all(np.abs(lp - rp) <= bound), (name, np.abs(lp - rp), bound)
    m.close()


def test_lite_gate_cascade(tmp_path, golden_heads_r04):
    """`load_model(cascade=True)` finds `<name>_lite` next to the main model (nanointerpreter.py:310-325) - here the reference's
    distilled student shape (distill.py:45-76) - evaluates it first and skips the verifier while the gate is closed."""
    from nanowakeword_amd.interpreter import HipInterpreter
    from nanowakeword_amd.weights import infer_head_config, save_bundle
    d, meta = golden_heads_r04
    main = HeadConfig("dnn", (16, 96))
    lite = HeadConfig(**meta["lite_dnn_16x96"])
    sd_main, sd_lite = synth_state_dict(main), synth_state_dict(lite)
    inferred = infer_head_config(sd_lite, input_shape=(16, 96))
    assert (inferred.layer_dim, inferred.n_blocks, inferred.embedding_dim) == (8, 1, 8)
    save_bundle(os.path.join(tmp_path, "kw.nww.npz"), main, sd_main, mode="features")
    save_bundle(os.path.join(tmp_path, "kw_lite.nww.npz"), inferred, sd_lite, mode="features")

    class Pre:                                              # AudioFeatures protocol with scripted features
        def __init__(self):
            self.feature_buffer = np.zeros((0, 96), np.float32); self.n = 0; self.k = 0

        def __call__(self, x):
            self.n += len(x)
            k, self.n = divmod(self.n, 1280)
            for _ in range(k):
                self.feature_buffer = np.vstack([self.feature_buffer, synth_features(1, (1, 96), seed=100 + self.k)[0]])[-120:]
                self.k += 1
            return k * 1280

        def get_features(self, n):
            return self.feature_buffer[-n:][None]

        def reset(self):
            self.__init__()

    x = synth_pcm("noise", 1, 1280)[0]
    for thr, closed in ((0.0, False), (1.1, True)):
        pre = Pre()
        it = HipInterpreter.load_model(os.path.join(tmp_path, "kw.nww.npz"), cascade=True, gate_threshold=thr, preprocessor=pre)
        assert it.is_cascade and it.gate_name == "kw_lite" and list(it.models) == ["kw_lite", "kw"]
        for _ in range(24):
            r = it.predict(x)
        feats = pre.get_features(16)
        g_ref = float(oracle.sigmoid(oracle.model_forward(feats, sd_lite, lite))[0, 0])
        v_ref = float(oracle.sigmoid(oracle.model_forward(feats, sd_main, main))[0, 0])
        assert abs(r.gate_score - g_ref) <= 1e-5
        assert (r.score == 0.0 and it.raw_scores["kw"] == 0.0) if closed else abs(r.score - v_ref) <= 1e-5


@pytest.mark.parametrize("head,B", [("cnn", 65536), ("bcresnet", 65536), ("conformer", 16384), ("dnn", 131072), ("dnn", 524288), ("cnn", 262144)])
def test_global_batch_on_one_gpu(HipModel, head, B):
    """Maximum sizes: BASELINE's GLOBAL batches (65 536 BcResNet / CNN clips, 16 384 Conformer clips - what eight GPUs share) and twice that for
    the DNN head, on ONE GPU, then 4-8 x those (16.8 GB of PCM, 3.4e9 log-mel floats: element offsets beyond 2^31 and 2^32 in every plane).  The oracle cannot run these; the
    property that can be checked everywhere is batch invariance: the batch is 256 distinct clips tiled, and EVERY clip's logit must be the
    bits the same clip gets in a 256-clip call (which test_head_vs_oracle_and_golden ties to the oracle)."""
    import torch
    cfg = HeadConfig(head, (101, 64))
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
    base = synth_pcm("speechlike", 256, 16000, seed=11)
    base[7] = 0                                                # one silent clip (-100 dB floor rows)
    want, _ = m.forward_pcm(base)
    dev = torch.device("cuda", 0)
    pcm = torch.from_numpy(base).to(dev).repeat(B // 256, 1).contiguous()
    assert pcm.shape == (B, 16000)
    out = torch.empty(B, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    m.reserve(B, 16000)
    m.forward_pcm_dev(pcm.data_ptr(), B, 16000, out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(B // 256, 256)
    assert np.isfinite(got).all()
    bad = np.argwhere(got != want[None, :])
    assert bad.size == 0, (head, B, bad[:5], len(bad))
    del pcm, out
    torch.cuda.empty_cache()
    m.close()


@pytest.mark.parametrize("head", ["cnn", "crnn", "dnn"])
def test_ten_second_clips(HipModel, golden_frontend, head):
    """Long inputs: 10 s clips (160 000 samples -> 1001 frames; BASELINE config 4's stream length as ONE window) through the frontend
    (criteria A / B against the oracle) and a head built for (1001, 64): many row strips in the fused trunk, K = 128 000 / 64 064 in the
    first Linear, a 4 000-feature sequence step for the CRNN - against the oracle at 1e-4."""
    from parity import assert_frontend_close
    g = golden_frontend
    cfg = HeadConfig(head, (1001, 64), **({"layer_dim": 64} if head == "dnn" else {}))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    x = np.concatenate([synth_pcm("speechlike", 2, 160000, seed=5), synth_pcm("noise", 1, 160000, seed=6)])
    x[1, 40000:90000] = 0                                     # three seconds of digital silence inside a clip
    assert m.num_frames(160000) == 1001
    db, mel = m.frontend(x, return_power=True)
    mo = oracle.mel_power(x, g["window"], g["fb64"])
    assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), "10 s")
    lg, _ = m.forward_pcm(x)
    lm = oracle.logmel_db(mo).transpose(0, 2, 1)
    ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
    assert np.abs(lg - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), (head, np.abs(lg - ref), ref)
    l1, _ = m.forward_pcm(x[1:2])
    assert np.array_equal(l1, lg[1:2])
    m.close()


@pytest.mark.parametrize("n_mels,center", [(32, True), (48, False), (80, True), (96, True), (128, False)])
def test_frontend_other_mel_widths(HipModel, golden_frontend, n_mels, center):
    """The mel contraction has three forms chosen by the filterbank (register-resident filters; the sparse LDS loop for filters wider than
    the registers take - 32 bins; MFMA tiles for 80 .. 128 bins): each against the pinned oracle on the golden clips, with torchaudio's own
    tables (criteria A / B of tests/parity.py, and C against the float64 graph).  (Rounds 2-5 reached the other two forms through NWW_FE_MEL.)"""
    from nanowakeword_amd.session import torchaudio_tables
    from parity import assert_frontend_amplitude, assert_frontend_close
    g = golden_frontend
    fe = FrontendConfig(n_mels=n_mels, center=center)
    window, fb = torchaudio_tables(fe)
    T = oracle.frame_count(g["pcm"].shape[1], center=center)
    cfg = HeadConfig("dnn", (T, n_mels))
    m = HipModel(cfg, fe, state_dict=synth_state_dict(cfg), window=window, mel_fb=fb)
    db, mel = m.frontend(g["pcm"], return_power=True)
    ref_mel = oracle.mel_power(g["pcm"], window, fb, center=center)
    ref_db = oracle.frontend_logmel(g["pcm"], window, fb, n_mels=n_mels, center=center)
    assert_frontend_close(mel, db, ref_mel, ref_db, f"n_mels={n_mels}")
    assert_frontend_amplitude(mel, oracle.mel_power(g["pcm"], window, fb, center=center, dtype=np.float64), f"n_mels={n_mels}")
    m.close()


def test_streamed_recurrence_two_tiles_per_workgroup(HipModel):
    """rnn_stream.hip takes two 16-clip tiles per workgroup beyond 16 x CUs clips (67.6 KB of dynamic LDS at width 256): a clip's logit is the
    same bits in a batch of 4200 as in a batch of 16, and agrees with the oracle.  (Rounds 4-5 forced this form with NWW_RNN_STREAM_TILES.)"""
    for kw in (dict(model_type="gru", input_shape=(10, 64), layer_dim=256), dict(model_type="crnn", input_shape=(16, 96), crnn_rnn_type="lstm", layer_dim=192)):
        cfg = HeadConfig(**kw)
        sd = synth_state_dict(cfg)
        m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[1]), state_dict=sd)
        assert "W_hh streamed" in m.describe_plan(), m.describe_plan()
        x = synth_features(4200, cfg.input_shape, seed=5)
        lg, _ = m.forward_features(x)
        l16, _ = m.forward_features(x[:16])
        assert np.isfinite(lg).all() and np.array_equal(lg[:16], l16)
        assert np.abs(l16 - oracle.model_forward(x[:16], sd, cfg).ravel()).max() <= 1e-4
        m.close()


@pytest.mark.parametrize("name", head_case_names_r06())
def test_round6_heads_vs_reference(HipModel, golden_heads_r06, golden_frontend, name):
    """Conformer d_model 256 / 4 heads and 192 / 4 (head dims 64 / 48) must run on the fused kernels (ffn_x3, lin_x3, the two-term attention core -
    rounds 2-5 dropped them onto the general GEMMs and the float32 attention: 6.9 ms at B = 2048), and the default width at 70 / 128 frames on
    the one-launch attention module - against goldens made by the reference's own Model, ragged batches against the oracle, batch invariance."""
    d, meta = golden_heads_r06
    g = golden_frontend
    cfg = HeadConfig(**meta[name])
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[1]), state_dict=sd, **({"window": g["window"], "mel_fb": g["fb64"]} if cfg.input_shape[1] == 64 else {}))
    plan = m.describe_plan()
    assert "ffn_x3:" in plan and "gemm:model.conformer" not in plan, plan
    if cfg.conformer_d_model == 144:
        assert "attn_x3:" in plan, plan
    else:
        assert "mha_h2:" in plan and "lin_x3:model.conformer_blocks.0.attention.in_proj" in plan, plan
    feats = synth_features(4, cfg.input_shape)
    logits, probs, emb = m.forward_features(feats, return_embedding=True)
    ref = d[f"{name}/logits_feat"].ravel()
    assert np.abs(logits - ref).max() <= 1e-4, (name, np.abs(logits - ref).max())
    e_ref = d[f"{name}/emb_feat"]
    assert np.abs(emb - e_ref).max() <= 1e-4 * max(1.0, np.abs(e_ref).max())
    for B in (1, 9, 130):
        fx = synth_features(B, cfg.input_shape, seed=B)
        lg, _ = m.forward_features(fx)
        assert np.abs(lg - oracle.model_forward(fx, sd, cfg).ravel()).max() <= 1e-4, (name, B)
        l1, _ = m.forward_features(fx[:1])
        assert l1[0] == lg[0], (name, B, "batch dependence")
    if f"{name}/logits_pcm" in d:
        lp, _ = m.forward_pcm(g["pcm"])
        rp = d[f"{name}/logits_pcm"].ravel()
        lm32 = oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"]).transpose(0, 2, 1)
        lm64 = oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"], dtype=np.float64).astype(np.float32).transpose(0, 2, 1)
        bound = logit_bounds([str(n) for n in g["names"]], rp, oracle.model_forward(np.ascontiguousarray(lm32), sd, cfg).ravel(),
                             oracle.model_forward(np.ascontiguousarray(lm64), sd, cfg).ravel())
        assert np.all(np.abs(lp - rp) <= bound), (name, np.abs(lp - rp), bound)
    m.close()
