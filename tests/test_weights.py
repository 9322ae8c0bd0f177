"""Weight ingestion: reference .pt state_dict -> config inference -> bundle. CPU-only,
plus one GPU end-to-end test through HipInterpreter."""
import os

import numpy as np
import pytest

from conftest import head_case_names
from nanowakeword_amd.config import FrontendConfig, HeadConfig, param_spec
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
from nanowakeword_amd.weights import infer_head_config, load_bundle, save_bundle, state_dict_from_pt


@pytest.mark.parametrize("name", head_case_names())
def test_infer_head_config_roundtrip(golden_heads, name):
    _, meta = golden_heads
    cfg = HeadConfig(**meta[name])
    sd = synth_state_dict(cfg)
    got = infer_head_config(sd, input_shape=cfg.input_shape, activation=cfg.activation)
    assert got.model_type == cfg.model_type and got.layer_dim == (cfg.layer_dim if cfg.model_type in ("dnn", "crnn", "gru") else got.layer_dim)
    assert got.n_blocks == (cfg.n_blocks if cfg.model_type in ("dnn", "crnn", "gru", "conformer") else got.n_blocks)
    assert got.embedding_dim == cfg.embedding_dim
    assert dict(param_spec(got)) == dict(param_spec(cfg))


def test_infer_rejects_wrong_shape_and_foreign_heads():
    cfg = HeadConfig("cnn", (101, 64))
    sd = synth_state_dict(cfg)
    with pytest.raises(ValueError, match="size mismatch"):
        infer_head_config(sd, input_shape=(98, 40))
    with pytest.raises(ValueError):
        infer_head_config(sd)                                  # cnn needs input_shape
    with pytest.raises(ValueError):
        infer_head_config({"classifier.0.weight": np.zeros((32, 64), np.float32), "model.lstm.weight_ih_l0": np.zeros((4, 4))}, (16, 96))


def test_pt_and_bundle_roundtrip(tmp_path):
    import torch
    cfg = HeadConfig("dnn", (16, 96), layer_dim=32, n_blocks=2, embedding_dim=16, activation="gelu")
    sd = synth_state_dict(cfg)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    tsd["model.fake_bn.num_batches_tracked"] = torch.tensor(3)
    pt = os.path.join(tmp_path, "m.pt")
    torch.save(tsd, pt)                                         # what export_pytorch_model writes (_export/pytorch.py:26-46)
    back = state_dict_from_pt(pt)
    assert set(back) == set(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)
    inferred = infer_head_config(back, (16, 96), "gelu")
    b = os.path.join(tmp_path, "m.nww.npz")
    window = np.hanning(400).astype(np.float32)
    save_bundle(b, inferred, back, FrontendConfig(n_mels=40, center=False), mode="features", window=window)
    head, fe, sd2, extras, meta = load_bundle(b)
    assert head.to_dict() == inferred.to_dict() and fe.n_mels == 40 and fe.center is False and meta["mode"] == "features"
    assert all(np.array_equal(sd2[k], sd[k]) for k in sd) and np.array_equal(extras["frontend.window"], window)
    with pytest.raises(ValueError):
        save_bundle(os.path.join(tmp_path, "x.bin"), inferred, back)


@pytest.mark.gpu
def test_pt_to_interpreter_end_to_end(tmp_path, golden_frontend):
    """reference .pt -> bundle -> HipInterpreter.load_model(path) -> predict(), incl. cascade auto-discovery."""
    import torch
    import oracle
    from nanowakeword_amd.interpreter import HipInterpreter
    g = golden_frontend
    cfg = HeadConfig("e2e_dnn", (64, 101))
    sd = synth_state_dict(cfg)
    pt = os.path.join(tmp_path, "kw.pt")
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, pt)
    back = state_dict_from_pt(pt)
    save_bundle(os.path.join(tmp_path, "kw.nww.npz"), infer_head_config(back), back, window=g["window"], mel_fb=g["fb64"])
    lite = HeadConfig("e2e_dnn", (64, 101), embedding_dim=8)
    save_bundle(os.path.join(tmp_path, "kw_lite.nww.npz"), lite, synth_state_dict(lite, seed=3), window=g["window"], mel_fb=g["fb64"])
    it = HipInterpreter.load_model(os.path.join(tmp_path, "kw.nww.npz"), cascade=True, gate_threshold=0.0)
    assert it.is_cascade and it.gate_name == "kw_lite" and it.model_name == "kw" and list(it.models) == ["kw_lite", "kw"]
    stream = synth_pcm("noise", 1, 16000 * 2, seed=9)[0]
    scores = [it.predict(stream[i:i + 1280]) for i in range(0, len(stream) - 1279, 1280)]
    assert all(s.score == 0.0 for s in scores[:12])             # window not full yet (12.5 hops) -> 0
    last_clip = stream[len(stream) // 1280 * 1280 - 16000: len(stream) // 1280 * 1280]
    lm = oracle.frontend_logmel(last_clip[None], g["window"], g["fb64"])
    ref = float(oracle.sigmoid(oracle.model_forward(lm, sd, cfg))[0, 0])
    assert abs(it.raw_scores["kw"] - ref) <= 1e-5 and abs(scores[-1].score - ref) <= 1e-5
    res = it.predict_clip(stream[:16000])
    assert len(res) == 1
