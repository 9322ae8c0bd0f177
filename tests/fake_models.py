"""Deterministic stand-ins for the reference's opaque ONNX models (melspectrogram.onnx, embedding_model.onnx),
shared by tools/make_goldens.py (wrapped as fake onnxruntime sessions) and tests/test_audio_features.py."""
import numpy as np


def fake_mel(x: np.ndarray) -> np.ndarray:
    """float32 [B, n] (int16-valued) -> [B, 1, frames, 32], frames = 1 + (n - 512)//160 (the ONNX model's law,
    SURVEY.md §8: 97 frames per second, 8 frames per 1280+480 samples)."""
    x = np.atleast_2d(np.asarray(x, np.float32))
    B, n = x.shape
    frames = 1 + (n - 512) // 160
    out = np.empty((B, 1, frames, 32), np.float32)
    band = 1.0 + np.arange(32, dtype=np.float32) / 32.0
    for f in range(frames):
        e = np.abs(x[:, 160 * f:160 * f + 512]).mean(axis=1) / 1000.0
        out[:, 0, f, :] = e[:, None] * band[None, :] - 20.0
    return out


def fake_embed(batch: np.ndarray) -> np.ndarray:
    """float32 [W, 76, 32, 1] -> [W, 1, 1, 96]"""
    b = np.asarray(batch, np.float32)[..., 0]
    W = b.shape[0]
    m = b.mean(axis=(1, 2))
    out = m[:, None] * (np.arange(96, dtype=np.float32) + 1.0) / 96.0 + b[:, 0, np.arange(96) % 32]
    return out.reshape(W, 1, 1, 96).astype(np.float32)
