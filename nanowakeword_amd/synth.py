"""Deterministic synthetic inputs and weights (no network, no trained checkpoints).

Everything here is a pure function of (seed, key, shape) through numpy's PCG64
stream, so the golden generator (tools/make_goldens.py, run where the reference
is importable), the tests and bench.py (run on the GPU box, where it is not)
all see bit-identical PCM and weights without shipping multi-MB state_dicts.
SURVEY.md §8d: noise = default_rng(10).integers(-8192, 8192) (~ -12 dBFS).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import HeadConfig, param_spec

SEED = 10  # the reference's global SEED (nanowakeword/modules/model.py:49)


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def synth_pcm(kind: str, n_clips: int, n_samples: int = 16000, seed: int = SEED) -> np.ndarray:
    """int16 [n_clips, n_samples] test signals."""
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    if kind == "noise":
        return _rng(seed, "pcm.noise").integers(-8192, 8192, size=(n_clips, n_samples)).astype(np.int16)
    if kind == "loud":      # full-range white noise, exercises int16 extremes
        return _rng(seed, "pcm.loud").integers(-32768, 32768, size=(n_clips, n_samples)).astype(np.int16)
    if kind == "zeros":     # exercises the -100 dB clamp floor
        return np.zeros((n_clips, n_samples), np.int16)
    if kind == "sine":
        out = np.empty((n_clips, n_samples), np.int16)
        for i in range(n_clips):
            f = 440.0 * (1 + i)
            out[i] = np.round(0.5 * 32767 * np.sin(2 * np.pi * f * t)).astype(np.int16)
        return out
    if kind == "chirp":     # linear 50 -> 7900 Hz
        dur = n_samples / 16000.0
        ph = 2 * np.pi * (50.0 * t + 0.5 * (7900.0 - 50.0) / dur * t * t)
        row = np.round(0.5 * 32767 * np.sin(ph)).astype(np.int16)
        return np.repeat(row[None], n_clips, 0)
    if kind == "square":    # +-32767 square, 100 Hz
        row = np.where((np.floor(t * 200.0).astype(np.int64) & 1) == 0, 32767, -32767).astype(np.int16)
        return np.repeat(row[None], n_clips, 0)
    if kind == "speechlike":  # amplitude-modulated band noise: well-conditioned, non-stationary
        r = _rng(seed, "pcm.speechlike")
        x = r.standard_normal((n_clips, n_samples))
        env = 0.15 + 0.85 * np.abs(np.sin(2 * np.pi * 3.0 * t + r.uniform(0, 6.28, (n_clips, 1))))
        y = x + 0.6 * np.roll(x, 1, axis=1) + 0.3 * np.roll(x, 2, axis=1)
        y = y / np.abs(y).max(axis=1, keepdims=True)
        return np.round(0.6 * 32767 * env * y).astype(np.int16)
    raise ValueError(f"unknown pcm kind {kind!r}")


def synth_features(n_clips: int, shape, seed: int = SEED) -> np.ndarray:
    """float32 [n_clips, T, F] stand-in features with log-mel-dB-like statistics (-30 +- 20 dB)."""
    r = _rng(seed, "features")
    return (-30.0 + 20.0 * r.standard_normal((n_clips,) + tuple(shape))).astype(np.float32)


def _tensor_for(key: str, shape, seed: int) -> np.ndarray:
    r = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 1)[0]
    is_norm = ("norm" in parent) or ("bn" in parent.rsplit(".", 1)[-1]) or leaf.startswith("running_") \
        or _is_seq_bn(key)
    if leaf == "running_mean":
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if is_norm and leaf == "weight":
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if is_norm and leaf == "bias":
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    # contraction weights: variance-preserving uniform (He for the ReLU-family stacks) so that
    # activations and logits stay O(1) through the head; first layers (which see raw dB-scale
    # log-mel, roughly 32x unit scale) carry a 1/32 gain as a trained model's would.
    if len(shape) == 1:
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    recurrent = ("weight_ih" in leaf) or ("weight_hh" in leaf) or leaf == "in_proj_weight"
    k = np.sqrt((3.0 if recurrent else 6.0) / fan_in)
    if key in _FIRST_LAYER_KEYS:
        k /= 32.0
    return r.uniform(-k, k, shape).astype(np.float32)


_FIRST_LAYER_KEYS = {
    "model.layer1.weight", "model.conv1.weight", "model.cnn.0.weight", "model.gru.weight_ih_l0",
    "model.gru.weight_ih_l0_reverse", "model.init_conv.0.weight", "model.input_proj.weight",
    "model.conv_block.0.weight",
}


def _is_seq_bn(key: str) -> bool:
    """BatchNorm layers that sit in nn.Sequential containers have numeric names
    (model.cnn.1, model.conv_block.5, model.init_conv.1, model.blockN.shortcut.1)."""
    parts = key.split(".")
    if len(parts) < 3:
        return False
    parent_leaf, cont = parts[-2], parts[-3]
    if not parent_leaf.isdigit():
        return False
    n = int(parent_leaf)
    if cont in ("cnn", "conv_block"):
        return n % 4 == 1
    if cont in ("init_conv", "shortcut"):
        return n == 1
    return False


def synth_state_dict(cfg: HeadConfig, seed: int = SEED) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 weights for every key of param_spec(cfg)."""
    sd = OrderedDict()
    for key, shape in param_spec(cfg).items():
        sd[key] = _tensor_for(key, shape, seed)
    return sd


def state_dict_checksum(sd) -> str:
    """Order-independent digest so fixtures can pin the generated weights."""
    h = 0
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k], dtype=np.float32)
        h = zlib.crc32(k.encode(), h)
        h = zlib.crc32(a.tobytes(), h)
    return f"{h:08x}"
