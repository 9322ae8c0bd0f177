"""Batched feature generation on the GPU: the caller side of the frontend.

Mirrors the batch interface the reference's feature job uses (``AudioFeatures.embed_clips`` /
``get_embedding_shape``; reference: nanowakeword/data/AudioFeatures.py:182-186,297-322) and the job itself
(``_process_embedding_generation_job``, nanowakeword/transform_clips.py:389-464): int16 audio batches ->
float32 features -> rows of an ``np.lib.format.open_memmap`` file of shape (total_clips, frames, F), then the
trailing all-zero rows are trimmed (``trim_mmap``, nanowakeword/data/trim_mmap.py:27-89) - the layout the
training datasets read (nanowakeword/data/data_sampler.py:136-140,198).

Only the in-source frontend is computed (log-mel dB, frames-major (frames, n_mels) like Model(input_shape=
(frames, n_mels)) consumes); the reference's default 96-d speech embedding is an un-vendored ONNX model and is
not restated.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import numpy as np
from numpy.lib.format import open_memmap

from .config import FrontendConfig, HeadConfig


class HipFeatures:
    """Log-mel feature extractor on one MI355X with AudioFeatures' batch interface."""

    def __init__(self, frontend: Optional[FrontendConfig] = None, device: int = 0, window=None, mel_fb=None):
        from .session import HipModel
        from .synth import synth_state_dict
        self.fe = frontend or FrontendConfig()
        # the C-ABI handle always carries a head; a minimal DNN stands in (never evaluated by frontend calls)
        head = HeadConfig("dnn", (self.fe.n_frames(16000), self.fe.n_mels), layer_dim=4, n_blocks=0, embedding_dim=2)
        self._model = HipModel(head, self.fe, device=device, state_dict=synth_state_dict(head), window=window, mel_fb=mel_fb)

    def get_embedding_shape(self, audio_length_sec: float, sr: int = 16000):
        """(frames, n_mels) for a clip of that duration (AudioFeatures.py:182-186 returns its (frames, 96))."""
        return (self.fe.n_frames(int(sr * audio_length_sec)), self.fe.n_mels)

    def embed_clips(self, x: np.ndarray, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """int16 [N, samples] -> float32 [N, frames, n_mels].  ``ncpu`` is accepted for signature
        compatibility (AudioFeatures.py:297) and unused: the whole batch is one kernel launch."""
        if not isinstance(x, np.ndarray) or x.dtype != np.int16 or x.ndim != 2:
            raise ValueError("`x` must be an int16 array of shape (N, samples)")
        if x.shape[1] < self.fe.n_fft:
            raise ValueError(f"clips must have at least {self.fe.n_fft} samples")       # cf. AudioFeatures.py:390-391
        out = []
        step = max(int(batch_size), 1)
        for i in range(0, x.shape[0], step):
            lm = self._model.frontend(np.ascontiguousarray(x[i:i + step]))              # [b, n_mels, frames]
            out.append(np.ascontiguousarray(lm.transpose(0, 2, 1)))
        return np.concatenate(out, axis=0) if out else np.zeros((0,) + self.get_embedding_shape(x.shape[1] / 16000), np.float32)

    def close(self):
        self._model.close()


def trim_mmap(path: str, block_rows: int = 1024) -> int:
    """Drop trailing all-zero rows of a 3-D float32 .npy memmap in place; returns the rows kept."""
    src = np.load(path, mmap_mode="r")
    if src.ndim != 3:
        raise ValueError("expected a (rows, frames, features) array")
    rows = src.shape[0]
    while rows > 0 and not np.any(src[rows - 1]):
        rows -= 1
    tmp = path[:-4] + "_tmp.npy" if path.endswith(".npy") else path + "_tmp"
    dst = open_memmap(tmp, mode="w+", dtype=np.float32, shape=(rows,) + tuple(src.shape[1:]))
    for lo in range(0, rows, block_rows):
        hi = min(lo + block_rows, rows)
        dst[lo:hi] = src[lo:hi]
    dst.flush()
    del src, dst
    os.replace(tmp, path)
    return rows


def generate_features(audio_batches: Iterable[np.ndarray], total_clips: int, output_path: str, extractor,
                      clip_seconds: float, overwrite: bool = True) -> Optional[int]:
    """Run one feature-generation job: batches of int16 [B, samples] -> memmap rows -> trim.
    Returns the number of rows written, or None if the file exists and ``overwrite`` is False."""
    if os.path.exists(output_path) and not overwrite:
        return None
    frames, feat = extractor.get_embedding_shape(clip_seconds)
    fp = open_memmap(output_path, mode="w+", dtype=np.float32, shape=(int(total_clips), frames, feat))
    row = 0
    for batch in audio_batches:
        if row >= total_clips:
            break
        f = extractor.embed_clips(batch, batch_size=len(batch))
        end = min(row + f.shape[0], total_clips)
        fp[row:end] = f[:end - row]
        row = end
        fp.flush()
    del fp
    return trim_mmap(output_path)
