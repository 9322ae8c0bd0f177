"""Weight ingestion for the HIP path.

* ``state_dict_from_pt``  : the reference's final artefact ``<name>.pt`` = ``torch.save(model.state_dict())``
                            (reference: nanowakeword/_export/pytorch.py:26-46) -> {key: float32 ndarray}.
* ``infer_head_config``   : recover Model()'s hyper-parameters from the state_dict keys/shapes
                            (the .pt carries no config; reference: nanowakeword/modules/model.py:67-296).
* ``state_dict_from_onnx`` : the reference's deployable ``<name>.onnx`` (reference: nanowakeword/_export/onnx.py:157-229,
                            TorchScript exporter of torch 2.8) -> (HeadConfig, state_dict, info): undoes what the exporter
                            did to the weights (Conv+BN folding, Linear -> MatMul with transposed weight, GRU gate
                            re-packing) by walking the graph; no ``onnx`` package needed (``onnx_reader``).
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
        gates = shp("model.rnn.weight_hh_l0")[0] // L          # nn.GRU stacks 3 gates, nn.LSTM 4 (architectures.py:238-254)
        if gates not in (3, 4) or shp("model.rnn.weight_hh_l0")[0] != gates * L:
            raise ValueError(f"crnn: unexpected recurrent weight shape {shp('model.rnn.weight_hh_l0')}")
        cfg = HeadConfig("crnn", input_shape, layer_dim=L, n_blocks=n_indexed(r"model\.rnn\.weight_hh_l(\d+)$"),
                         crnn_cnn_channels=chans, crnn_rnn_type="lstm" if gates == 4 else "gru", **kw)
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


# ---------------------------------------------------------------------------------------------------------------------
# .onnx ingestion
_WRAP = "trained_model."                 # InferenceWrapper attribute name, _export/onnx.py:163-172


def _bn_identity(sd, prefix, c, beta=None):
    """BatchNorm tensors that make ``prefix`` the exact map x -> x + beta: the exporter has already folded the trained
    statistics into the preceding Conv.  running_var = 1 - eps so that var + eps rounds to exactly 1.0f."""
    sd[prefix + ".weight"] = np.ones(c, np.float32)
    sd[prefix + ".bias"] = np.zeros(c, np.float32) if beta is None else np.asarray(beta, np.float32)
    sd[prefix + ".running_mean"] = np.zeros(c, np.float32)
    sd[prefix + ".running_var"] = np.full(c, np.float32(1.0) - np.float32(1e-5), np.float32)


def _consumers(g, tensor):
    return [n for n in g.nodes if tensor in n.inputs]


def _activation_after(g, tensor) -> str:
    ops = {n.op_type for n in _consumers(g, tensor)}
    if "Relu" in ops:
        return "relu"
    if "Sigmoid" in ops and "Mul" in ops:
        return "silu"                                  # x * sigmoid(x)
    if "Div" in ops or "Erf" in ops:
        return "gelu"                                  # 0.5 x (1 + erf(x / sqrt 2))
    raise ValueError(f"unrecognised activation pattern after '{tensor}': {sorted(ops)}")


def _unpack_onnx_gru(sd, prefix, layer, W, R, B, hidden):
    """ONNX GRU packs gates [z, r, h] per direction; nn.GRU stores [r, z, n] (torch symbolic for aten::gru)."""
    H = hidden
    perm = np.r_[H:2 * H, 0:H, 2 * H:3 * H]
    for d, sfx in enumerate(("", "_reverse")):
        sd[f"{prefix}.weight_ih_l{layer}{sfx}"] = np.ascontiguousarray(W[d][perm], np.float32)
        sd[f"{prefix}.weight_hh_l{layer}{sfx}"] = np.ascontiguousarray(R[d][perm], np.float32)
        sd[f"{prefix}.bias_ih_l{layer}{sfx}"] = np.ascontiguousarray(B[d][:3 * H][perm], np.float32)
        sd[f"{prefix}.bias_hh_l{layer}{sfx}"] = np.ascontiguousarray(B[d][3 * H:][perm], np.float32)


def _unpack_onnx_lstm(sd, prefix, layer, W, R, B, hidden):
    """ONNX LSTM packs gates [i, o, f, c] per direction; nn.LSTM stores [i, f, g, o] (torch symbolic for aten::lstm)."""
    H = hidden
    perm = np.r_[0:H, 2 * H:3 * H, 3 * H:4 * H, H:2 * H]
    for d, sfx in enumerate(("", "_reverse")):
        sd[f"{prefix}.weight_ih_l{layer}{sfx}"] = np.ascontiguousarray(W[d][perm], np.float32)
        sd[f"{prefix}.weight_hh_l{layer}{sfx}"] = np.ascontiguousarray(R[d][perm], np.float32)
        sd[f"{prefix}.bias_ih_l{layer}{sfx}"] = np.ascontiguousarray(B[d][:4 * H][perm], np.float32)
        sd[f"{prefix}.bias_hh_l{layer}{sfx}"] = np.ascontiguousarray(B[d][4 * H:][perm], np.float32)


def state_dict_from_onnx(path_or_bytes):
    """Recover (HeadConfig, state_dict, info) from a reference-exported ONNX model.

    ``info``: {"mode": "e2e"|"features", "clip_samples", "frontend": FrontendConfig|None, "opset", "producer"}.
    The state_dict uses the reference's ``Model.state_dict()`` keys.  Convs the exporter fused with their BatchNorm
    come back as (folded weight, folded bias, identity BatchNorm) - the same function, so logits agree with the
    original state_dict to float32 rounding of the folding itself.
    """
    from .onnx_reader import read_onnx
    g = read_onnx(path_or_bytes)
    if len(g.inputs) != 1 or g.inputs[0][1] is None:
        raise ValueError("expected one graph input with a static shape (reference exports name it 'input')")
    in_shape = g.inputs[0][1]
    named = {k[len(_WRAP):]: v for k, v in g.initializers.items() if k.startswith(_WRAP)}
    sd = {k: np.asarray(v, np.float32) for k, v in named.items()}
    anon = lambda t: t in g.initializers and not t.startswith(_WRAP)

    # Linear on a 3-D input is exported as MatMul(x, W^T) + Add(bias): the bias keeps its name.
    for n in g.nodes:
        if n.op_type != "MatMul" or len(n.inputs) != 2 or not anon(n.inputs[1]):
            continue
        bias = [t for c in _consumers(g, n.outputs[0]) if c.op_type == "Add" for t in c.inputs if t.startswith(_WRAP)]
        if len(bias) != 1:
            raise ValueError(f"MatMul '{n.name}': cannot identify the Linear it came from")
        b = bias[0][len(_WRAP):]
        key = b[:-len("in_proj_bias")] + "in_proj_weight" if b.endswith("in_proj_bias") else b[:-len("bias")] + "weight"
        sd[key] = np.ascontiguousarray(g.initializers[n.inputs[1]].T, np.float32)

    convs = [n for n in g.nodes if n.op_type == "Conv" and anon(n.inputs[1])]
    grus = [n for n in g.nodes if n.op_type == "GRU"]
    lstms = [n for n in g.nodes if n.op_type == "LSTM"]
    if lstms and not any(n.op_type == "Conv" and anon(n.inputs[1]) for n in g.nodes):
        raise NotImplementedError("model_type='lstm' (LSTMModel) is out of scope; the CRNN's LSTM backend is supported")

    def folded(node, wkey, bias_key, bn_prefix):
        W = np.asarray(g.initializers[node.inputs[1]], np.float32)
        b = np.asarray(g.initializers[node.inputs[2]], np.float32) if len(node.inputs) > 2 else np.zeros(W.shape[0], np.float32)
        sd[wkey] = W
        if bias_key is not None:
            sd[bias_key] = b
            _bn_identity(sd, bn_prefix, W.shape[0])
        else:
            _bn_identity(sd, bn_prefix, W.shape[0], beta=b)

    def gru_layers(prefix):
        for l, n in enumerate(grus):
            if n.attrs.get("direction") != b"bidirectional" or n.attrs.get("linear_before_reset", 0) != 1:
                raise ValueError(f"GRU node '{n.name}': expected the bidirectional nn.GRU export")
            H = int(n.attrs["hidden_size"])
            W, R, B = (np.asarray(g.initializers[t]) for t in n.inputs[1:4])
            _unpack_onnx_gru(sd, prefix, l, W, R, B, H)

    def lstm_layers(prefix):
        for l, n in enumerate(lstms):
            if n.attrs.get("direction") != b"bidirectional":
                raise ValueError(f"LSTM node '{n.name}': expected the bidirectional nn.LSTM export")
            H = int(n.attrs["hidden_size"])
            W, R, B = (np.asarray(g.initializers[t]) for t in n.inputs[1:4])
            _unpack_onnx_lstm(sd, prefix, l, W, R, B, H)

    mode, clip_samples, fe = "features", 16000, None
    if "model.mel_spec.real_basis" in named:                                        # E2E_MelSpectrogram_CNN
        mode = "e2e"
        clip_samples = int(in_shape[-1])
        rb, fb = named["model.mel_spec.real_basis"], named["model.mel_spec.mel_fb"]
        stft = [n for n in g.nodes if n.op_type == "Conv" and n.inputs[1] == _WRAP + "model.mel_spec.real_basis"]
        hop = int(stft[0].attrs["strides"][0])
        center = any(n.op_type == "Pad" and n.attrs.get("mode") == b"reflect" for n in g.nodes)
        fe = FrontendConfig(n_fft=int(rb.shape[-1]), win_length=int(rb.shape[-1]), hop_length=hop, n_mels=int(fb.shape[1]), center=center)
        for i, n in enumerate(convs):
            folded(n, f"model.conv_block.{4*i}.weight", f"model.conv_block.{4*i}.bias", f"model.conv_block.{4*i+1}")
        n_frames = fe.n_frames(clip_samples)
        input_shape = (fe.n_mels, n_frames)
    else:
        if len(in_shape) != 3:
            raise ValueError(f"feature-mode model with input shape {in_shape}: expected [batch, T, F]")
        input_shape = (int(in_shape[1]), int(in_shape[2]))
        if "model.block1.depthwise.weight" in named:                                # BcResNet
            folded(convs[0], "model.init_conv.0.weight", None, "model.init_conv.1")
            for i in (1, 2, 3):
                dw = next(n for n in g.nodes if n.op_type == "Conv" and n.inputs[1] == f"{_WRAP}model.block{i}.depthwise.weight")
                pw = [n for n in convs if n.inputs[0] == dw.outputs[0]]
                sc = [n for n in convs if n.inputs[0] == dw.inputs[0]]
                if len(pw) != 1 or len(sc) != 1:
                    raise ValueError(f"bcresnet block{i}: unexpected graph structure")
                folded(pw[0], f"model.block{i}.pointwise.weight", None, f"model.block{i}.bn1")
                folded(sc[0], f"model.block{i}.shortcut.0.weight", None, f"model.block{i}.shortcut.1")
        elif any(k.startswith("model.conformer_blocks.") for k in named):           # Conformer
            for i, n in enumerate(convs):
                p = f"model.conformer_blocks.{i}.conv_module"
                folded(n, f"{p}.depthwise_conv.weight", f"{p}.depthwise_conv.bias", f"{p}.batch_norm")
        elif (grus or lstms) and convs:                                             # CRNN (GRU or LSTM backend)
            for i, n in enumerate(convs):
                folded(n, f"model.cnn.{4*i}.weight", f"model.cnn.{4*i}.bias", f"model.cnn.{4*i+1}")
            gru_layers("model.rnn") if grus else lstm_layers("model.rnn")
        elif grus:                                                                  # GRU
            gru_layers("model.gru")

    cls = [n for n in g.nodes if n.op_type == "Gemm" and len(n.inputs) > 1 and n.inputs[1] == _WRAP + "classifier.0.weight"]
    if len(cls) != 1:
        raise ValueError("no 'classifier.0' Gemm: not a reference Model export")
    act = _activation_after(g, cls[0].outputs[0])
    cfg = infer_head_config(sd, input_shape=input_shape, activation=act)
    if cfg.model_type == "conformer":
        sm = [n for n in g.nodes if n.op_type == "Softmax"]
        qk = g.producer_of(sm[0].inputs[0]) if sm else None
        sc_node = g.producer_of(qk.inputs[0]) if qk is not None else None
        c = None
        if sc_node is not None and sc_node.op_type == "Mul":
            c = next((g.constant(t) for t in sc_node.inputs if g.constant(t) is not None), None)
        if c is None:
            raise ValueError("conformer: cannot find the attention scale (1/sqrt(head_dim)) in the graph")
        dh = int(round(1.0 / float(np.asarray(c).ravel()[0]) ** 2))
        cfg.conformer_n_head = cfg.conformer_d_model // dh
    if g.metadata.get("mode") == "e2e" and mode != "e2e":
        raise ValueError("ONNX metadata says mode=e2e but the graph has no mel front end")
    info = {"mode": mode, "clip_samples": clip_samples, "frontend": fe, "opset": g.opset, "producer": g.producer,
            "input_ndim": len(in_shape)}
    return cfg, sd, info


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
    """Bundle or reference .onnx -> finalized HipModel -> HipSession (raises if the HIP library or a GPU is missing)."""
    from .session import HipModel, HipSession
    if path.endswith((".pt", ".pth")):
        raise ValueError(f"{path}: a raw PyTorch state_dict carries no architecture; convert it once with "
                         "weights.state_dict_from_pt() + infer_head_config() + save_bundle() "
                         "and load the resulting .nww.npz")
    if path.endswith(".onnx"):                          # the reference's own artefact (nanointerpreter.py:955-959)
        head, sd, info = state_dict_from_onnx(path)
        fe = info["frontend"] or FrontendConfig()
        extras, meta = {}, {"mode": info["mode"], "clip_samples": info["clip_samples"], "input_ndim": info["input_ndim"]}
    else:
        head, fe, sd, extras, meta = load_bundle(path)
    # feature-mode heads never run the frontend and e2e exports carry their own tables (model.mel_spec.*): neither
    # needs torch to rebuild torchaudio's float32 tables
    feature_mode = meta.get("mode", "e2e") == "features"
    own_tables = "frontend.window" in extras or any(k.startswith("model.mel_spec.") for k in sd)
    model = HipModel(head, fe, device=device, state_dict=sd, window=extras.get("frontend.window"),
                     mel_fb=extras.get("frontend.mel_fb"), tables="builtin" if (feature_mode or own_tables) else "torchaudio")
    name = path.rsplit("/", 1)[-1]
    for ext in (".nww.npz", ".npz", ".onnx"):
        if name.endswith(ext):
            name = name[:-len(ext)]
            break
    s = HipSession(model, mode=meta.get("mode", "e2e"), clip_samples=int(meta.get("clip_samples", 16000)),
                   input_ndim=int(meta.get("input_ndim", 3)), name=name)
    s.name = name
    return s
