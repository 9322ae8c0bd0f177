"""Head / frontend configuration and the state_dict key+shape specification.

The hot path accepts exactly the hyper-parameters the reference's ``Model``
factory reads for the in-scope heads (reference: nanowakeword/modules/model.py:67-296)
and the frontend parameters of ``E2E_MelSpectrogram_CNN``'s ``T.MelSpectrogram``
(reference: nanowakeword/modules/architectures.py:830-837).

``param_spec`` lists every tensor of ``Model.state_dict()`` (minus
``num_batches_tracked``) for a head, with the same keys, so that a reference
``state_dict`` can be fed to ``nww_load_tensor`` unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple

HEAD_TYPES = ("dnn", "cnn", "crnn", "gru", "bcresnet", "conformer", "e2e_dnn")
ACTIVATIONS = ("relu", "gelu", "silu")

# integer codes shared with include/nww.h
HEAD_CODE = {"dnn": 0, "cnn": 1, "crnn": 2, "gru": 3, "bcresnet": 4, "conformer": 5, "e2e_dnn": 6}
ACT_CODE = {"relu": 0, "gelu": 1, "silu": 2}


@dataclass
class FrontendConfig:
    """STFT -> mel -> dB parameters (torchaudio MelSpectrogram/AmplitudeToDB semantics)."""
    sample_rate: int = 16000
    n_fft: int = 400
    win_length: int = 400
    hop_length: int = 160
    n_mels: int = 64
    center: bool = True          # reflect-pad n_fft//2 each side (reference e2e path)
    f_min: float = 0.0
    f_max: float = 8000.0
    amin: float = 1e-10          # AmplitudeToDB clamp floor
    db_multiplier: float = 10.0  # stype="power"

    def n_frames(self, n_samples: int) -> int:
        """Frame law (bit-exact requirement). center: 1+N//hop ; else 1+(N-n_fft)//hop."""
        if self.center:
            if n_samples <= self.n_fft // 2:
                raise ValueError("reflect padding needs more than n_fft//2 samples")
            return 1 + n_samples // self.hop_length
        if n_samples < self.n_fft:
            raise ValueError(f"clip shorter than n_fft={self.n_fft} samples")
        return 1 + (n_samples - self.n_fft) // self.hop_length


@dataclass
class HeadConfig:
    """Mirror of the kwargs/config keys Model() reads (model.py:68-69,81-90,212-261)."""
    model_type: str = "dnn"
    input_shape: Tuple[int, int] = (16, 96)   # (T, F) as the reference's input_shape
    layer_dim: int = 128
    n_blocks: int = 1
    embedding_dim: int = 64
    activation: str = "relu"
    crnn_cnn_channels: List[int] = field(default_factory=lambda: [16, 32, 32])
    crnn_rnn_type: str = "gru"
    conformer_d_model: int = 144
    conformer_n_head: int = 4

    def __post_init__(self):
        self.model_type = self.model_type.lower()
        self.activation = self.activation.lower()
        self.input_shape = tuple(int(v) for v in self.input_shape)
        if self.model_type not in HEAD_TYPES:
            raise ValueError(f"Unsupported model_type: '{self.model_type}'.")
        if self.activation not in ACTIVATIONS:
            # the reference silently falls back to ReLU (model.py:86-87)
            self.activation = "relu"
        # CRNNModel: 'gru' -> nn.GRU, anything else -> nn.LSTM (architectures.py:238-254); the reference's default
        # config value is "lstm" (model.py:214), BASELINE config 4 names the GRU - HeadConfig defaults to the latter
        self.crnn_rnn_type = "gru" if str(self.crnn_rnn_type).lower() == "gru" else "lstm"

    def to_dict(self):
        return asdict(self)


def _bn(spec, prefix, c):
    spec[prefix + ".weight"] = (c,)
    spec[prefix + ".bias"] = (c,)
    spec[prefix + ".running_mean"] = (c,)
    spec[prefix + ".running_var"] = (c,)


def _lin(spec, prefix, out_f, in_f):
    spec[prefix + ".weight"] = (out_f, in_f)
    spec[prefix + ".bias"] = (out_f,)


def _ln(spec, prefix, d):
    spec[prefix + ".weight"] = (d,)
    spec[prefix + ".bias"] = (d,)


def _gru(spec, prefix, input_size, hidden, n_layers, gates=3):
    """nn.GRU (gates = 3: r, z, n) / nn.LSTM (gates = 4: i, f, g, o) parameter names, bidirectional."""
    for l in range(n_layers):
        isz = input_size if l == 0 else 2 * hidden
        for sfx in ("", "_reverse"):
            spec[f"{prefix}.weight_ih_l{l}{sfx}"] = (gates * hidden, isz)
            spec[f"{prefix}.weight_hh_l{l}{sfx}"] = (gates * hidden, hidden)
            spec[f"{prefix}.bias_ih_l{l}{sfx}"] = (gates * hidden,)
            spec[f"{prefix}.bias_hh_l{l}{sfx}"] = (gates * hidden,)


def crnn_cnn_out(input_shape, channels):
    """(C, H, W) after the CRNN conv stack: each stage MaxPool2d(2) floor mode."""
    h, w = input_shape
    for _ in channels:
        h, w = h // 2, w // 2
    return channels[-1], h, w


def param_spec(cfg: HeadConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys/shapes of Model.state_dict() for the head (num_batches_tracked omitted)."""
    T, F = cfg.input_shape
    L, E, nb = cfg.layer_dim, cfg.embedding_dim, cfg.n_blocks
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    mt = cfg.model_type
    if mt == "dnn":                       # architectures.py:110-126
        _lin(s, "model.layer1", L, T * F)
        _ln(s, "model.layernorm1", L)
        for i in range(nb):
            _lin(s, f"model.blocks.{i}.fcn_layer", L, L)
            _ln(s, f"model.blocks.{i}.layer_norm", L)
        _lin(s, "model.last_layer", E, L)
    elif mt == "cnn":                     # architectures.py:51-80
        s["model.conv1.weight"] = (16, 1, 3, 3); s["model.conv1.bias"] = (16,)
        s["model.conv2.weight"] = (32, 16, 3, 3); s["model.conv2.bias"] = (32,)
        _lin(s, "model.fc1", 128, 32 * (T // 4) * (F // 4))
        _lin(s, "model.fc2", E, 128)
    elif mt == "crnn":                    # architectures.py:209-287
        cin = 1
        for i, c in enumerate(cfg.crnn_cnn_channels):
            s[f"model.cnn.{4*i}.weight"] = (c, cin, 3, 3); s[f"model.cnn.{4*i}.bias"] = (c,)
            _bn(s, f"model.cnn.{4*i+1}", c)
            cin = c
        C, H, W = crnn_cnn_out((T, F), cfg.crnn_cnn_channels)
        _gru(s, "model.rnn", C * H, L, nb, gates=4 if cfg.crnn_rnn_type == "lstm" else 3)
        _lin(s, "model.fc", E, 2 * L)
    elif mt == "gru":                     # architectures.py:129-145
        _gru(s, "model.gru", F, L, nb)
        _lin(s, "model.fc", E, 2 * L)
    elif mt == "bcresnet":                # architectures.py:620-687
        s["model.init_conv.0.weight"] = (32, 1, 3, 3)
        _bn(s, "model.init_conv.1", 32)
        for i, (ci, co) in enumerate(((32, 64), (64, 128), (128, 256)), start=1):
            s[f"model.block{i}.depthwise.weight"] = (ci, 1, 3, 3)
            s[f"model.block{i}.pointwise.weight"] = (co, ci, 1, 1)
            _bn(s, f"model.block{i}.bn1", co)
            s[f"model.block{i}.shortcut.0.weight"] = (co, ci, 1, 1)
            _bn(s, f"model.block{i}.shortcut.1", co)
        _lin(s, "model.fc", E, 256)
    elif mt == "conformer":               # architectures.py:441-543
        D = cfg.conformer_d_model
        _lin(s, "model.input_proj", D, F)
        for i in range(nb):
            p = f"model.conformer_blocks.{i}"
            for ff in ("ff1", "ff2"):
                _ln(s, f"{p}.{ff}.layer_norm", D)
                _lin(s, f"{p}.{ff}.linear1", 4 * D, D)
                _lin(s, f"{p}.{ff}.linear2", D, 4 * D)
            s[f"{p}.attention.in_proj_weight"] = (3 * D, D)
            s[f"{p}.attention.in_proj_bias"] = (3 * D,)
            _lin(s, f"{p}.attention.out_proj", D, D)
            _ln(s, f"{p}.conv_module.layer_norm", D)
            s[f"{p}.conv_module.conv1.weight"] = (2 * D, D, 1); s[f"{p}.conv_module.conv1.bias"] = (2 * D,)
            s[f"{p}.conv_module.depthwise_conv.weight"] = (D, 1, 31); s[f"{p}.conv_module.depthwise_conv.bias"] = (D,)
            _bn(s, f"{p}.conv_module.batch_norm", D)
            s[f"{p}.conv_module.conv2.weight"] = (D, D, 1); s[f"{p}.conv_module.conv2.bias"] = (D,)
            _ln(s, f"{p}.layer_norm", D)
        _lin(s, "model.output_proj", E, D)
    elif mt == "e2e_dnn":                 # architectures.py:840-865 (E2E_MelSpectrogram_CNN body)
        cin = 1
        for i, c in enumerate((16, 32, 64)):
            s[f"model.conv_block.{4*i}.weight"] = (c, cin, 3, 3); s[f"model.conv_block.{4*i}.bias"] = (c,)
            _bn(s, f"model.conv_block.{4*i+1}", c)
            cin = c
        _lin(s, "model.fc1", 128, 256)
        _bn(s, "model.bn1", 128)
        _lin(s, "model.out", E, 128)
    # Model.classifier: model.py:291-296
    _lin(s, "classifier.0", E // 2, E)
    _lin(s, "classifier.3", 1, E // 2)
    return s


def head_macs(cfg: HeadConfig) -> int:
    """Multiply-accumulates per clip of the head's contractions (SURVEY.md §8a figures)."""
    T, F = cfg.input_shape
    L, E, nb = cfg.layer_dim, cfg.embedding_dim, cfg.n_blocks
    mt = cfg.model_type
    m = E * (E // 2) + E // 2
    if mt == "dnn":
        m += T * F * L + nb * L * L + L * E
    elif mt == "cnn":
        m += 9 * 16 * T * F + 9 * 16 * 32 * (T // 2) * (F // 2) + 32 * (T // 4) * (F // 4) * 128 + 128 * E
    elif mt == "e2e_dnn":
        h, w = T, F   # here input_shape = (n_mels, frames)
        m += 9 * 16 * h * w + 9 * 16 * 32 * (h // 2) * (w // 2) + 9 * 32 * 64 * (h // 4) * (w // 4) + 256 * 128 + 128 * E
    elif mt == "crnn":
        h, w, cin = T, F, 1
        for c in cfg.crnn_cnn_channels:
            m += 9 * cin * c * h * w
            h, w, cin = h // 2, w // 2, c
        I = cin * h
        for l in range(nb):
            isz = I if l == 0 else 2 * L
            steps_rev = w if l < nb - 1 else 1
            ng = 4 if cfg.crnn_rnn_type == "lstm" else 3
            m += w * ng * L * (isz + L) + steps_rev * ng * L * (isz + L)
        m += 2 * L * E
    elif mt == "gru":
        for l in range(nb):
            isz = F if l == 0 else 2 * L
            steps_rev = T if l < nb - 1 else 1
            m += T * 3 * L * (isz + L) + steps_rev * 3 * L * (isz + L)
        m += 2 * L * E
    elif mt == "bcresnet":
        h, w = T, F
        m += 9 * 32 * h * w
        h, w = h // 2, w // 2
        for ci, co, sh, sw in ((32, 64, 2, 2), (64, 128, 2, 2), (128, 256, 2, 1)):
            ho, wo = (h - 1) // sh + 1, (w - 1) // sw + 1
            m += 9 * ci * ho * wo + 2 * ci * co * ho * wo
            h, w = ho, wo
        m += 256 * E
    elif mt == "conformer":
        D = cfg.conformer_d_model
        m += T * F * D + D * E
        per = 2 * (2 * T * D * 4 * D) + T * 3 * D * D + 2 * T * T * D + T * D * D \
            + T * D * 2 * D + 31 * T * D + T * D * D
        m += nb * per
    return int(m)
