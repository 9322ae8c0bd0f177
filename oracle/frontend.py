"""Oracle frontend: int16 PCM -> frames -> windowed DFT -> power -> mel -> dB.

Follows the form the reference's CPU interpreter executes, i.e. the exported
``ONNXSafeMelSpectrogram`` (reference: nanowakeword/_export/onnx.py:27-83) fed
by ``T.MelSpectrogram(16000, n_fft=400, win_length=400, hop_length=160,
n_mels=64)`` + ``T.AmplitudeToDB()`` (nanowakeword/modules/architectures.py:830-837,
873-875), after the interpreter's ``x.astype(float32)/32768.0``
(nanowakeword/interpreter/nanointerpreter.py:750).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np


def frame_count(n_samples: int, n_fft=400, hop=160, center=True) -> int:
    """conv1d output length over the (optionally reflect-padded) signal.

    onnx.py:70-75: pad n_fft//2 both sides when center, then conv1d(kernel=n_fft,
    stride=hop): L_out = (L_in - n_fft)//hop + 1.
    """
    L = n_samples + (2 * (n_fft // 2) if center else 0)
    if L < n_fft:
        raise ValueError("clip shorter than n_fft")
    return (L - n_fft) // hop + 1


def default_tables(sample_rate=16000, n_fft=400, win_length=400, n_mels=64, f_min=0.0, f_max=None):
    """Hann(periodic) window and HTK triangular mel filterbank, torchaudio semantics.

    torchaudio.transforms.MelSpectrogram defaults relied on at architectures.py:830-836:
    window_fn=torch.hann_window (periodic), f_min=0, f_max=sr/2, mel_scale="htk", norm=None.
    Filterbank = torchaudio.functional.melscale_fbanks (published formula):
      all_freqs = linspace(0, sr//2, n_fft//2+1); m = 2595 log10(1+f/700);
      f_pts = mel^-1(linspace(m(f_min), m(f_max), n_mels+2));
      fb = max(0, min(-slopes[:, :-2]/df[:-1], slopes[:, 2:]/df[1:])).
    Evaluated in float64 and rounded once to float32.
    """
    if f_max is None:
        f_max = float(sample_rate // 2)
    n = np.arange(win_length, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(np.float32)
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up)).astype(np.float32)
    return window, fb


def dft_bases(window: np.ndarray, n_fft=400):
    """real/imag conv1d kernels [n_fft//2+1, n_fft] (onnx.py:42-63).

    angle = -2*pi*k*n/n_fft in Python double, cos/sin stored into float32 tensors,
    then multiplied (float32) by the centre-padded window.
    """
    win_length = window.shape[0]
    k = np.arange(n_fft, dtype=np.float64)[:, None]
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    angle = -2.0 * math.pi * k * n / n_fft
    real = np.cos(angle).astype(np.float32)
    imag = np.sin(angle).astype(np.float32)
    wp = np.zeros(n_fft, np.float32)
    pad_left = (n_fft - win_length) // 2
    wp[pad_left:pad_left + win_length] = window.astype(np.float32)
    n_bins = n_fft // 2 + 1
    return (real * wp[None, :])[:n_bins], (imag * wp[None, :])[:n_bins]


def frame_signal(x: np.ndarray, n_fft=400, hop=160, center=True) -> np.ndarray:
    """[B, N] float -> [B, frames, n_fft] strided frames (onnx.py:69-75).

    center: torch.nn.functional.pad(x, (n_fft//2, n_fft//2), mode="reflect"), i.e.
    padded[i] = x[|i - p|] on the left and x[2(N-1) - (i - p)] on the right.
    """
    if center:
        p = n_fft // 2
        x = np.pad(x, ((0, 0), (p, p)), mode="reflect")
    T = (x.shape[1] - n_fft) // hop + 1
    idx = (np.arange(T) * hop)[:, None] + np.arange(n_fft)[None, :]
    return x[:, idx]


def mel_power(pcm: np.ndarray, window: np.ndarray, mel_fb: np.ndarray, n_fft=400, hop=160,
              center=True, dtype=np.float32) -> np.ndarray:
    """int16 [B, N] -> mel power [B, n_mels, frames].

    nanointerpreter.py:750 (x/32768.0), onnx.py:66-83: conv1d with real/imag bases,
    real**2 + imag**2, matmul(power^T, mel_fb), transposed back.
    ``dtype=np.float64`` gives the exact-arithmetic variant used to judge which of
    two float32 results is closer to the truth.
    """
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[None]
    x = pcm.astype(np.float32) / np.float32(32768.0)
    real, imag = dft_bases(window, n_fft)
    frames = frame_signal(x, n_fft, hop, center).astype(dtype)          # [B,T,n_fft]
    B, T, _ = frames.shape
    f2 = frames.reshape(B * T, n_fft)                                     # one contiguous GEMM per basis
    re = f2 @ np.ascontiguousarray(real.astype(dtype).T)                  # [B*T,bins]
    im = f2 @ np.ascontiguousarray(imag.astype(dtype).T)
    power = re * re + im * im
    mel = (power @ mel_fb.astype(dtype)).reshape(B, T, -1)                # [B,T,n_mels]
    return np.ascontiguousarray(np.swapaxes(mel, 1, 2))


def logmel_db(mel: np.ndarray, amin=1e-10, multiplier=10.0) -> np.ndarray:
    """torchaudio AmplitudeToDB(stype='power', ref=1.0, top_db=None) (architectures.py:837,875):
    multiplier*log10(clamp(x, min=amin)) - multiplier*log10(max(amin, ref)); the second term is 0."""
    dt = mel.dtype
    return (dt.type(multiplier) * np.log10(np.maximum(mel, dt.type(amin)))).astype(dt)


def frontend_logmel(pcm, window=None, mel_fb=None, n_mels=64, n_fft=400, hop=160, center=True,
                    dtype=np.float32):
    """PCM -> log-mel dB [B, n_mels, frames]."""
    if window is None or mel_fb is None:
        w, fb = default_tables(n_fft=n_fft, win_length=n_fft, n_mels=n_mels)
        window = w if window is None else window
        mel_fb = fb if mel_fb is None else mel_fb
    return logmel_db(mel_power(pcm, window, mel_fb, n_fft, hop, center, dtype))
