"""Reference .onnx -> HIP path.

Fixtures under tests/golden/onnx/ are real exports: tools/make_goldens.py ran every in-scope head through the
reference's own ``export_onnx_model`` (nanowakeword/_export/onnx.py:157-229, TorchScript exporter, opset 17) and
stored the exported wrapper's logits/probabilities on fixed inputs.  CPU tests pin the dependency-free reader and
the graph walk that undoes Conv+BN folding / Linear->MatMul / GRU gate packing; GPU tests run the ingested model
through the C-ABI and compare with the stored reference outputs.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from nanowakeword_amd.config import HeadConfig, param_spec
from nanowakeword_amd.onnx_reader import read_onnx
from nanowakeword_amd.synth import synth_features, synth_state_dict
from nanowakeword_amd.weights import state_dict_from_onnx

ONNX_DIR = os.path.join(GOLDEN, "onnx")
HEADS = ["dnn", "cnn", "crnn", "crnn_lstm", "gru", "bcresnet", "conformer", "e2e_dnn"]   # crnn_lstm = the reference's default CRNN backend


@pytest.fixture(scope="module")
def expected():
    z = np.load(os.path.join(ONNX_DIR, "expected.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    return d, json.loads(str(d.pop("meta_json")))


def test_reader_decodes_graph_structure():
    g = read_onnx(os.path.join(ONNX_DIR, "crnn.onnx"))
    assert g.opset == 17 and g.producer == "pytorch"
    assert g.inputs == [("input", ["batch_size", 8, 16])]
    assert g.outputs[0][0] == "output"
    ops = [n.op_type for n in g.nodes]
    assert ops.count("Conv") == 3 and ops.count("GRU") == 2 and ops.count("Gemm") == 3 and ops[-1] == "Reshape"
    gru = next(n for n in g.nodes if n.op_type == "GRU")
    assert gru.attrs["hidden_size"] == 16 and gru.attrs["direction"] == b"bidirectional" and gru.attrs["linear_before_reset"] == 1
    conv = next(n for n in g.nodes if n.op_type == "Conv")
    assert conv.attrs["kernel_shape"] == [3, 3] and conv.attrs["pads"] == [1, 1, 1, 1] and conv.attrs["group"] == 1
    w = g.initializers[conv.inputs[1]]
    assert w.shape == (16, 1, 3, 3) and w.dtype == np.float32
    assert g.initializers["trained_model.classifier.3.bias"].shape == (1,)
    # same bytes through the bytes entry point
    with open(os.path.join(ONNX_DIR, "crnn.onnx"), "rb") as f:
        g2 = read_onnx(f.read())
    assert [n.op_type for n in g2.nodes] == ops


def test_reader_rejects_garbage():
    with pytest.raises(ValueError):
        read_onnx(b"\x0a\x03abc")                      # a protobuf without a graph
    with pytest.raises(ValueError):
        read_onnx(b"\x3a\xff\xff\x01")                 # truncated length-delimited field


@pytest.mark.parametrize("name", HEADS)
def test_ingest_recovers_config_and_function(expected, name):
    import oracle
    exp, meta = expected
    cfg = HeadConfig(**{**meta[name], "input_shape": tuple(meta[name]["input_shape"])})
    got_cfg, sd, info = state_dict_from_onnx(os.path.join(ONNX_DIR, name + ".onnx"))
    assert got_cfg.to_dict() == cfg.to_dict()
    assert info["mode"] == ("e2e" if name == "e2e_dnn" else "features") and info["opset"] == 17
    spec = param_spec(cfg)
    assert all(k in sd and tuple(sd[k].shape) == s for k, s in spec.items())
    orig = synth_state_dict(cfg)                       # what the exported model was loaded with
    folded = [k for k in spec if not np.array_equal(orig[k], sd[k])]
    if name in ("dnn", "cnn", "gru"):
        assert not folded                              # nothing for the exporter to fold: bit-identical tensors
    if name == "e2e_dnn":
        fe = info["frontend"]
        assert (fe.n_fft, fe.hop_length, fe.n_mels, fe.center) == (400, 160, 64, True) and info["clip_samples"] == 16000
        window = sd["model.mel_spec.real_basis"][0, 0]
        lm = oracle.frontend_logmel(exp[name + "/pcm"], window, sd["model.mel_spec.mel_fb"])
        logits = oracle.model_forward(lm, sd, got_cfg).reshape(-1)
    else:
        logits = oracle.model_forward(synth_features(4, cfg.input_shape), sd, got_cfg).reshape(-1)
    scale = max(1.0, float(np.abs(exp[name + "/logits"]).max()))
    assert np.abs(logits - exp[name + "/logits"]).max() <= 2e-5 * scale
    assert np.abs(oracle.sigmoid(logits) - exp[name + "/probs"]).max() <= 1e-5


def test_gru_gate_unpacking_matches_original_layout(expected):
    _, meta = expected
    cfg = HeadConfig(**{**meta["gru"], "input_shape": tuple(meta["gru"]["input_shape"])})
    _, sd, _ = state_dict_from_onnx(os.path.join(ONNX_DIR, "gru.onnx"))
    orig = synth_state_dict(cfg)
    for l in range(cfg.n_blocks):
        for sfx in ("", "_reverse"):
            for t in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                k = f"model.gru.{t}_l{l}{sfx}"
                assert np.array_equal(sd[k], orig[k]), k


def test_lstm_gate_unpacking_matches_original_layout(expected):
    """ONNX packs LSTM gates [i, o, f, c]; the graph walk must give back nn.LSTM's [i, f, g, o] rows bit for bit."""
    _, meta = expected
    cfg = HeadConfig(**{**meta["crnn_lstm"], "input_shape": tuple(meta["crnn_lstm"]["input_shape"])})
    assert cfg.crnn_rnn_type == "lstm"
    got, sd, _ = state_dict_from_onnx(os.path.join(ONNX_DIR, "crnn_lstm.onnx"))
    assert got.crnn_rnn_type == "lstm" and got.layer_dim == 16 and got.n_blocks == 2
    orig = synth_state_dict(cfg)
    for l in range(cfg.n_blocks):
        for sfx in ("", "_reverse"):
            for t in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                k = f"model.rnn.{t}_l{l}{sfx}"
                assert sd[k].shape[0] == 4 * 16 and np.array_equal(sd[k], orig[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", HEADS)
def test_onnx_session_matches_reference_outputs(expected, name):
    """load_session('<ref>.onnx') -> HipSession.run(): probabilities of the exported reference model."""
    from nanowakeword_amd.weights import load_session
    exp, meta = expected
    s = load_session(os.path.join(ONNX_DIR, name + ".onnx"))
    assert s.name == name
    if name == "e2e_dnn":
        assert s.get_inputs()[0].shape == [None, 1, 16000]
        x = (exp[name + "/pcm"].astype(np.float32) / 32768.0)[:, None, :]
    else:
        T, F = meta[name]["input_shape"]
        assert s.get_inputs()[0].shape == [None, T, F]
        x = synth_features(4, (T, F))
    probs = s.run(None, {"input": x})[0]
    assert probs.shape == (4, 1, 1)
    assert np.abs(probs.reshape(-1) - exp[name + "/probs"]).max() <= 1e-5
    scale = max(1.0, float(np.abs(exp[name + "/logits"]).max()))
    assert np.abs(s.run_logits({"input": x}).reshape(-1) - exp[name + "/logits"]).max() <= 1e-4 * scale


@pytest.mark.gpu
def test_interpreter_loads_reference_onnx(expected):
    from nanowakeword_amd.interpreter import HipInterpreter
    exp, _ = expected
    it = HipInterpreter.load_model(os.path.join(ONNX_DIR, "e2e_dnn.onnx"))
    pcm = exp["e2e_dnn/pcm"]
    res = it.predict_clip(pcm[0])
    assert len(res) == 1
    assert abs(it.raw_scores["e2e_dnn"] - float(exp["e2e_dnn/probs"][0])) <= 1e-5
