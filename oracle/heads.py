"""Oracle heads: numpy float32 restatement of the in-scope classifier heads.

Each function cites the reference module it follows (nanowakeword/modules/
architectures.py, nanowakeword/modules/model.py).  Weights come in as a dict
keyed exactly like ``Model.state_dict()``.  Eval-mode semantics: Dropout is
identity, BatchNorm uses running stats (eps 1e-5), LayerNorm eps 1e-5 with
biased variance.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from scipy.special import erf

F32 = np.float32


# ----------------------------------------------------------------------------- primitives
def sigmoid(x):
    x = np.asarray(x)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(x.dtype)


def act(x, kind):
    """model.py:81-87: nn.ReLU / nn.GELU (exact erf form) / nn.SiLU."""
    if kind == "relu":
        return np.maximum(x, 0)
    if kind == "gelu":
        return (0.5 * x * (1.0 + erf(x / np.sqrt(F32(2.0))))).astype(x.dtype)
    if kind == "silu":
        return x * sigmoid(x)
    raise ValueError(kind)


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + F32(eps)) * w + b


def batch_norm(x, sd, prefix, eps=1e-5, axis=1):
    """Eval BatchNorm as PyTorch's CPU kernel evaluates it: alpha = w/sqrt(var+eps),
    beta = b - mean*alpha, y = x*alpha + beta."""
    alpha = sd[prefix + ".weight"] / np.sqrt(sd[prefix + ".running_var"] + F32(eps))
    beta = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * alpha
    shape = [1] * x.ndim
    shape[axis] = -1
    return x * alpha.reshape(shape) + beta.reshape(shape)


def conv2d(x, w, b=None, stride=(1, 1), pad=(1, 1), groups=1):
    """Cross-correlation conv (torch.nn.Conv2d). x [B,C,H,W], w [O,C/groups,kh,kw]."""
    B, C, H, W = x.shape
    O, Cg, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad[0], pad[0]), (pad[1], pad[1])))
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::stride[0], ::stride[1]]  # [B,C,Ho,Wo,kh,kw]
    if groups == 1:
        y = np.tensordot(win, w, axes=([1, 4, 5], [1, 2, 3]))          # [B,Ho,Wo,O]
        y = np.moveaxis(y, 3, 1)
    elif groups == C and Cg == 1:
        y = np.einsum("bchwij,cij->bchw", win, w[:, 0], optimize=True)
        if O != C:
            raise ValueError("depthwise multiplier != 1 unsupported")
    else:
        raise ValueError("groups")
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1)
    return np.ascontiguousarray(y.astype(x.dtype))


def maxpool2(x):
    """nn.MaxPool2d(kernel_size=2, stride=2), floor mode."""
    B, C, H, W = x.shape
    h, w = H // 2, W // 2
    v = x[:, :, :2 * h, :2 * w]
    return np.maximum(np.maximum(v[:, :, 0::2, 0::2], v[:, :, 0::2, 1::2]),
                      np.maximum(v[:, :, 1::2, 0::2], v[:, :, 1::2, 1::2]))


def avgpool_export(x, out_hw):
    """The pool the CPU interpreter executes: make_onnx_safe_adaptive_pool
    (_export/onnx.py:96-154) swaps AdaptiveAvgPool2d(out) for
    AvgPool2d(kernel=in-(out-1)*(in//out), stride=in//out)."""
    B, C, H, W = x.shape
    oh, ow = out_hw
    sh, sw = H // oh, W // ow
    kh, kw = H - (oh - 1) * sh, W - (ow - 1) * sw
    y = np.empty((B, C, oh, ow), x.dtype)
    for i in range(oh):
        for j in range(ow):
            y[:, :, i, j] = x[:, :, i * sh:i * sh + kh, j * sw:j * sw + kw].mean(axis=(2, 3))
    return y


def gru_cell(x_gates, h, w_hh, b_hh, H):
    """torch.nn.GRU cell, gate order (r,z,n); b_hn sits inside r*(...)."""
    hg = h @ w_hh.T + b_hh
    r = sigmoid(x_gates[:, :H] + hg[:, :H])
    z = sigmoid(x_gates[:, H:2 * H] + hg[:, H:2 * H])
    n = np.tanh(x_gates[:, 2 * H:] + r * hg[:, 2 * H:])
    return (1 - z) * n + z * h


def lstm_cell(xg, h, c, w_hh, b_hh, H):
    """torch.nn.LSTM cell, gate order i, f, g, o (CRNNModel's default backend, architectures.py:247-254)."""
    g = xg + h @ w_hh.T + b_hh
    i, f, gg, o = sigmoid(g[:, :H]), sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sigmoid(g[:, 3 * H:])
    c = f * c + i * gg
    return (o * np.tanh(c)).astype(xg.dtype), c.astype(xg.dtype)


def bigru_last(x, sd, prefix, n_layers, H, lstm=False):
    """nn.GRU / nn.LSTM (batch_first, bidirectional)(x)[0][:, -1, :].

    The reference takes ``rnn_out[:, -1, :]`` (architectures.py:142,282): the forward
    direction after all T steps concatenated with the reverse direction's output at
    t = T-1, which is its FIRST step (it has seen only x[T-1]).
    """
    B, T, _ = x.shape
    inp = x
    for l in range(n_layers):
        last = l == n_layers - 1
        outs = {}
        for sfx in ("", "_reverse"):
            w_ih = sd[f"{prefix}.weight_ih_l{l}{sfx}"]; w_hh = sd[f"{prefix}.weight_hh_l{l}{sfx}"]
            b_ih = sd[f"{prefix}.bias_ih_l{l}{sfx}"]; b_hh = sd[f"{prefix}.bias_hh_l{l}{sfx}"]
            xg = inp @ w_ih.T + b_ih                          # [B,T,3H]
            h = np.zeros((B, H), x.dtype)
            c = np.zeros((B, H), x.dtype)
            seq = np.zeros((B, T, H), x.dtype)
            order = range(T) if sfx == "" else range(T - 1, -1, -1)
            for t in order:
                if lstm:
                    h, c = lstm_cell(xg[:, t], h, c, w_hh, b_hh, H)
                else:
                    h = gru_cell(xg[:, t], h, w_hh, b_hh, H).astype(x.dtype)
                seq[:, t] = h
                if last and sfx == "_reverse":
                    break                                      # only t = T-1 is consumed
            outs[sfx] = seq
        inp = np.concatenate([outs[""], outs["_reverse"]], axis=-1)
    return inp[:, -1, :]


def mha(x, sd, prefix, n_head):
    """nn.MultiheadAttention(batch_first=True)(x, x, x)[0] in eval mode."""
    B, T, D = x.shape
    dh = D // n_head
    qkv = x @ sd[prefix + ".in_proj_weight"].T + sd[prefix + ".in_proj_bias"]
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, T, n_head, dh).transpose(0, 2, 1, 3) for i in range(3))
    q = q * F32(1.0 / np.sqrt(dh))                              # torch scales q before QK^T
    s = q @ k.transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
    return o @ sd[prefix + ".out_proj.weight"].T + sd[prefix + ".out_proj.bias"]


# ----------------------------------------------------------------------------- heads
def net_dnn(x, sd, cfg):
    """Net + FCNBlock (architectures.py:102-126)."""
    a = cfg.activation
    h = x.reshape(x.shape[0], -1)
    h = act(layer_norm(linear(h, sd["model.layer1.weight"], sd["model.layer1.bias"]),
                       sd["model.layernorm1.weight"], sd["model.layernorm1.bias"]), a)
    for i in range(cfg.n_blocks):
        p = f"model.blocks.{i}"
        h = act(layer_norm(linear(h, sd[p + ".fcn_layer.weight"], sd[p + ".fcn_layer.bias"]),
                           sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"]), a)
    return linear(h, sd["model.last_layer.weight"], sd["model.last_layer.bias"])


def net_cnn(x, sd, cfg):
    """CNNModel (architectures.py:51-80)."""
    a = cfg.activation
    h = x[:, None]
    h = maxpool2(act(conv2d(h, sd["model.conv1.weight"], sd["model.conv1.bias"]), a))
    h = maxpool2(act(conv2d(h, sd["model.conv2.weight"], sd["model.conv2.bias"]), a))
    h = h.reshape(h.shape[0], -1)
    h = act(linear(h, sd["model.fc1.weight"], sd["model.fc1.bias"]), a)
    return linear(h, sd["model.fc2.weight"], sd["model.fc2.bias"])


def net_crnn(x, sd, cfg):
    """CRNNModel (architectures.py:209-287), rnn_type 'gru' or the default 'lstm'."""
    a = cfg.activation
    h = x[:, None]
    for i in range(len(cfg.crnn_cnn_channels)):
        h = conv2d(h, sd[f"model.cnn.{4*i}.weight"], sd[f"model.cnn.{4*i}.bias"])
        h = maxpool2(act(batch_norm(h, sd, f"model.cnn.{4*i+1}"), a))
    B, C, H, W = h.shape
    seq = h.reshape(B, C * H, W).transpose(0, 2, 1)              # sequence over W (:272-276)
    last = bigru_last(np.ascontiguousarray(seq), sd, "model.rnn", cfg.n_blocks, cfg.layer_dim, lstm=cfg.crnn_rnn_type == "lstm")
    return linear(last, sd["model.fc.weight"], sd["model.fc.bias"])


def net_gru(x, sd, cfg):
    """GRUModel (architectures.py:129-145)."""
    last = bigru_last(x, sd, "model.gru", cfg.n_blocks, cfg.layer_dim)
    return linear(last, sd["model.fc.weight"], sd["model.fc.bias"])


def net_bcresnet(x, sd, cfg):
    """BcResNetModel / BcResNetBlock (architectures.py:620-687); activation BEFORE the residual add."""
    a = cfg.activation
    h = x[:, None]
    h = conv2d(h, sd["model.init_conv.0.weight"])
    h = maxpool2(act(batch_norm(h, sd, "model.init_conv.1"), a))
    for i, stride in ((1, (2, 2)), (2, (2, 2)), (3, (2, 1))):
        p = f"model.block{i}"
        res = batch_norm(conv2d(h, sd[p + ".shortcut.0.weight"], None, stride, (0, 0)), sd, p + ".shortcut.1")
        d = conv2d(h, sd[p + ".depthwise.weight"], None, stride, (1, 1), groups=h.shape[1])
        d = conv2d(d, sd[p + ".pointwise.weight"], None, (1, 1), (0, 0))
        h = act(batch_norm(d, sd, p + ".bn1"), a) + res
    h = h.mean(axis=(2, 3))
    return linear(h, sd["model.fc.weight"], sd["model.fc.bias"])


def _swish(x):
    return x * sigmoid(x)


def _ffn(x, sd, p):
    """FeedForwardModule (architectures.py:479-496)."""
    h = layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
    h = _swish(linear(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]))
    return linear(h, sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])


def _conv_module(x, sd, p):
    """ConvolutionModule (architectures.py:446-477): LN -> pw(2D) -> GLU -> dw k=31 'same' -> BN -> Swish -> pw."""
    B, T, D = x.shape
    h = layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
    h = h @ sd[p + ".conv1.weight"][:, :, 0].T + sd[p + ".conv1.bias"]      # [B,T,2D]
    h = h[..., :D] * sigmoid(h[..., D:])                                       # GLU over channels
    wd = sd[p + ".depthwise_conv.weight"][:, 0, :]                             # [D,31]
    K = wd.shape[1]
    hp = np.pad(h, ((0, 0), (K // 2, K - 1 - K // 2), (0, 0)))
    win = sliding_window_view(hp, K, axis=1)                                   # [B,T,D,K]
    h = np.einsum("btdk,dk->btd", win, wd, optimize=True).astype(x.dtype) + sd[p + ".depthwise_conv.bias"]
    h = _swish(batch_norm(h, sd, p + ".batch_norm", axis=2))
    return h @ sd[p + ".conv2.weight"][:, :, 0].T + sd[p + ".conv2.bias"]


def net_conformer(x, sd, cfg):
    """ConformerModel / ConformerBlock (architectures.py:499-543); no pre-LN on attention (:512-513)."""
    h = linear(x, sd["model.input_proj.weight"], sd["model.input_proj.bias"])
    for i in range(cfg.n_blocks):
        p = f"model.conformer_blocks.{i}"
        h = h + F32(0.5) * _ffn(h, sd, p + ".ff1")
        h = h + mha(h, sd, p + ".attention", cfg.conformer_n_head)
        h = h + _conv_module(h, sd, p + ".conv_module")
        h = h + F32(0.5) * _ffn(h, sd, p + ".ff2")
        h = layer_norm(h, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])
    return linear(h.mean(axis=1), sd["model.output_proj.weight"], sd["model.output_proj.bias"])


def net_e2e_cnn_body(x, sd, cfg):
    """E2E_MelSpectrogram_CNN after the dB stage (architectures.py:840-865,877-889).
    x is log-mel [B, n_mels, frames]."""
    a = cfg.activation
    h = x[:, None]
    for i in range(3):
        h = conv2d(h, sd[f"model.conv_block.{4*i}.weight"], sd[f"model.conv_block.{4*i}.bias"])
        h = act(batch_norm(h, sd, f"model.conv_block.{4*i+1}"), a)
        h = maxpool2(h) if i < 2 else avgpool_export(h, (1, 4))
    h = h.reshape(h.shape[0], -1)
    h = act(batch_norm(linear(h, sd["model.fc1.weight"], sd["model.fc1.bias"]), sd, "model.bn1"), a)
    return linear(h, sd["model.out.weight"], sd["model.out.bias"])


_NETS = {"dnn": net_dnn, "cnn": net_cnn, "crnn": net_crnn, "gru": net_gru,
         "bcresnet": net_bcresnet, "conformer": net_conformer, "e2e_dnn": net_e2e_cnn_body}


def head_forward(x, sd, cfg, dtype=F32):
    """features [B,T,F] float32 -> embedding [B,E].  dtype = numpy.float64 evaluates the same network in double precision
    (every primitive above follows its input's dtype): the yardstick the arithmetic modes of the HIP path are measured against."""
    x = np.ascontiguousarray(x, dtype=dtype)
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    return _NETS[cfg.model_type](x, sd, cfg).astype(dtype)


def model_forward(x, sd, cfg, dtype=F32):
    """Model.forward (model.py:562-571): embedding -> classifier MLP (model.py:291-296) -> logits [B,1]."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    e = head_forward(x, sd, cfg, dtype)
    h = act(linear(e, sd["classifier.0.weight"], sd["classifier.0.bias"]), cfg.activation)
    return linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"]).astype(dtype)
