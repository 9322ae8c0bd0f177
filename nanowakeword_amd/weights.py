"""Weight ingestion for the HIP path.

* ``state_dict_from_pt``  : the reference's final artefact ``<name>.pt`` = ``torch.save(model.state_dict())``
                            (reference: nanowakeword/_export/pytorch.py:26-46) -> {key: float32 ndarray}.
* ``infer_head_config``   : recover Model()'s hyper-parameters from the state_dict keys/shapes
                            (the .pt carries no config; reference: nanowakeword/modules/model.py:67-296).
* ``save_bundle/load_bundle/load_session`` : a self-describing ``*.nww.npz`` (config JSON + tensors +
                            optional frontend tables) that ``HipInterpreter.load_model`` accepts as a path,
                            playing the role the ``.onnx`` file plays for the reference interpreter.
"""
from __future__ import annotations

import json
import re
from typing import Mapping, Optional, Tuple

import numpy as np

from .config import FrontendConfig, HeadConfig, param_spec


def state_dict_from_pt(path: str) -> dict:
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, Mapping):
        raise TypeError(f"{path} does not hold a state_dict")
    if "model_state_dict" in sd:                       # training checkpoints (train_model.py:675-705)
        sd = sd["model_state_dict"]
    return {k: v.detach().cpu().numpy().astype(np.float32) if hasattr(v, "detach") else np.asarray(v, np.float32)
            for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def infer_head_config(sd: Mapping, input_shape: Optional[Tuple[int, int]] = None, activation: str = "relu") -> HeadConfig:
    """Model type / layer_dim / n_blocks / embedding_dim from key names and shapes.  ``input_shape`` is needed
    when the weights do not determine it (DNN/CNN flatten sizes fix only T*F or (T//4)*(F//4))."""
    keys = set(sd)
    shp = lambda k: tuple(np.shape(sd[k]))
    E = shp("classifier.0.weight")[1]
    kw = dict(embedding_dim=E, activation=activation)

    def n_indexed(pattern):
        idx = [int(m.group(1)) for k in keys for m in [re.match(pattern, k)] if m]
        return max(idx) + 1 if idx else 0

    if "model.layer1.weight" in keys:
        L, TF = shp("model.layer1.weight")
        if input_shape is None:
            raise ValueError(f"dnn weights fix T*F={TF} only: pass input_shape=(T, F)")
        cfg = HeadConfig("dnn", input_shape, layer_dim=L, n_blocks=n_indexed(r"model\.blocks\.(\d+)\.fcn_layer\.weight"), **kw)
    elif "model.conv1.weight" in keys:
        if input_shape is None:
            raise ValueError("cnn weights fix (T//4)*(F//4) only: pass input_shape=(T, F)")
        cfg = HeadConfig("cnn", input_shape, **kw)
    elif "model.cnn.0.weight" in keys:
        chans = []
        i = 0
        while f"model.cnn.{4*i}.weight" in keys:
            chans.append(shp(f"model.cnn.{4*i}.weight")[0]); i += 1
        if input_shape is None:
            raise ValueError("crnn: pass input_shape=(T, F)")
        L = shp("model.rnn.weight_hh_l0")[1]
        cfg = HeadConfig("crnn", input_shape, layer_dim=L, n_blocks=n_indexed(r"model\.rnn\.weight_hh_l(\d+)$"),
                         crnn_cnn_channels=chans, crnn_rnn_type="gru", **kw)
        if shp("model.rnn.weight_hh_l0")[0] != 3 * L:
            raise ValueError("crnn with an LSTM backend is out of scope (only crnn_rnn_type='gru')")
    elif "model.gru.weight_hh_l0" in keys:
        L = shp("model.gru.weight_hh_l0")[1]
        F = shp("model.gru.weight_ih_l0")[1]
        if input_shape is None:
            raise ValueError("gru: the sequence length is not in the weights; pass input_shape=(T, F)")
        if input_shape[1] != F:
            raise ValueError(f"input_shape F={input_shape[1]} but the GRU expects {F} features")
        cfg = HeadConfig("gru", input_shape, layer_dim=L, n_blocks=n_indexed(r"model\.gru\.weight_hh_l(\d+)$"), **kw)
    elif "model.init_conv.0.weight" in keys:
        if input_shape is None:
            raise ValueError("bcresnet is shape-independent; pass the input_shape=(T, F) you will feed")
        cfg = HeadConfig("bcresnet", input_shape, **kw)
    elif "model.input_proj.weight" in keys and any(k.startswith("model.conformer_blocks.") for k in keys):
        D, F = shp("model.input_proj.weight")
        if input_shape is None:
            raise ValueError("conformer: pass input_shape=(T, F)")
        nb = n_indexed(r"model\.conformer_blocks\.(\d+)\.layer_norm\.weight")
        cfg = HeadConfig("conformer", input_shape, n_blocks=nb, conformer_d_model=D, **kw)
        cfg.conformer_n_head = 4          # not recoverable from weights; Model()'s default (model.py:252)
    elif "model.conv_block.0.weight" in keys:
        cfg = HeadConfig("e2e_dnn", input_shape or (64, 101), **kw)
    else:
        raise ValueError("state_dict does not belong to an in-scope head (dnn/cnn/crnn-gru/gru/bcresnet/conformer/e2e_dnn)")
    spec = param_spec(cfg)
    for k, s in spec.items():
        if k not in keys:
            raise KeyError(f"Missing key(s) in state_dict: '{k}'")
        if shp(k) != s:
            raise ValueError(f"size mismatch for {k}: checkpoint {shp(k)} vs model {s} (wrong input_shape?)")
    return cfg


def save_bundle(path: str, head: HeadConfig, state_dict: Mapping, frontend: Optional[FrontendConfig] = None,
                mode: str = "e2e", clip_samples: int = 16000, window=None, mel_fb=None):
    if not path.endswith(".npz"):
        raise ValueError("bundle path must end in .npz (convention: <name>.nww.npz)")
    fe = frontend or FrontendConfig()
    meta = {"format": "nww-bundle-1", "head": head.to_dict(), "frontend": fe.__dict__, "mode": mode, "clip_samples": clip_samples}
    arrays = {"__config__": np.array(json.dumps(meta))}
    spec = param_spec(head)
    for k in spec:
        a = state_dict[k]
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        arrays["sd/" + k] = np.asarray(a, np.float32)
    sd_keys = set(state_dict)
    if window is None and "model.mel_spec.real_basis" in sd_keys:
        rb = np.asarray(state_dict["model.mel_spec.real_basis"], np.float32)
        pad = (fe.n_fft - fe.win_length) // 2
        window = rb.reshape(rb.shape[0], -1)[0, pad:pad + fe.win_length]
    if mel_fb is None and "model.mel_spec.mel_fb" in sd_keys:
        mel_fb = np.asarray(state_dict["model.mel_spec.mel_fb"], np.float32)
    if window is not None:
        arrays["frontend.window"] = np.asarray(window, np.float32)
    if mel_fb is not None:
        arrays["frontend.mel_fb"] = np.asarray(mel_fb, np.float32)
    np.savez(path, **arrays)


def load_bundle(path: str):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["__config__"]))
    if meta.get("format") != "nww-bundle-1":
        raise ValueError(f"{path}: not a nww bundle")
    hd = dict(meta["head"]); hd["input_shape"] = tuple(hd["input_shape"])
    head = HeadConfig(**hd)
    fe = FrontendConfig(**meta["frontend"])
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    extras = {k: z[k] for k in ("frontend.window", "frontend.mel_fb") if k in z.files}
    return head, fe, sd, extras, meta


def load_session(path: str, device: int = 0):
    """Bundle -> finalized HipModel -> HipSession (raises if the HIP library or a GPU is missing)."""
    from .session import HipModel, HipSession
    head, fe, sd, extras, meta = load_bundle(path)
    model = HipModel(head, fe, device=device, state_dict=sd, window=extras.get("frontend.window"),
                     mel_fb=extras.get("frontend.mel_fb"))
    name = path.rsplit("/", 1)[-1]
    for ext in (".nww.npz", ".npz"):
        if name.endswith(ext):
            name = name[:-len(ext)]
            break
    s = HipSession(model, mode=meta.get("mode", "e2e"), clip_samples=int(meta.get("clip_samples", 16000)), name=name)
    s.name = name
    return s
