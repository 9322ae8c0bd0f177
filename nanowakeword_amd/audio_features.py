"""Embedding-mode preprocessor: interface and windowing semantics of the reference's ``AudioFeatures``
(reference: nanowakeword/data/AudioFeatures.py) around two pluggable models.

The reference computes a 32-bin mel spectrogram and Google's 96-d speech embedding with two ONNX files that are
downloaded at first use (nanowakeword/interpreter/models/_registry.py:34-47) - un-vendored, unobtainable here, so
their ARITHMETIC is outside this build.  What is reference code, and is mirrored here, is everything
around them: the ``x/10 + 2`` mel transform (:124,146), 76-frame windows every 8 frames (:168-179,261-272), the
80 ms streaming state machine with remainder carry (:406-449), the 10 s raw / 970-frame mel / 120-row feature caps
(:106-112,397-398,446), ``get_features`` (:451-457), ``embed_clips`` with -80 padding (:188-227,297-322) and the error
conventions (:252-253,390-391).  ``mel_fn(x float32 [B,n]) -> [B,1,frames,bins]`` and
``embed_fn(windows float32 [W,76,bins,1]) -> [W,1,1,D]`` stand where the ONNX sessions stood; an instance plugs into
``HipInterpreter(sessions, preprocessor=...)`` exactly like ``AudioFeatures`` plugs into ``NanoInterpreter``.
The host-only mirror used to check this class lives with the other checkers (``oracle/audio_features.py``).
"""
from __future__ import annotations

from typing import Callable

import numpy as np

WINDOW_FRAMES = 76          # AudioFeatures.py:168
WINDOW_STEP = 8
CHUNK = 1280                # 80 ms
MEL_CONTEXT = 160 * 3       # extra samples fed to the mel model per update (:394)


class DeviceWindowedFeatures:
    """The reference's ``AudioFeatures`` preprocessor with the buffers and the windowing on the GPU (C-ABI ``nww_emb_*``, csrc/emb_stream.hip)
    for ``n_streams`` lock-step streams: the mel ring (ones((76,32)) at reset, x/10 + 2, newest 970 frames), the
    76-frame windows of the new 80 ms chunks, the 120-row feature ring and ``get_features`` live on the device next
    to the head, whose forward reads the last T rows of every stream without a host hop (``scores()``).  The two
    models stay pluggable host callables; the raw-audio bookkeeping in front of the
    mel model (remainder carry, the ``n + 480`` samples it is fed, AudioFeatures.py:394,406-424) stays on the host
    because that is where the pluggable mel model takes its input.

    ``backend`` is a finalized feature-mode ``HipModel`` with input_shape (T, emb_dim).  With ``n_streams == 1`` an
    instance is a drop-in for ``AudioFeatures`` in ``HipInterpreter(sessions, preprocessor=...)``; with more streams
    ``__call__`` takes int16 [S, n] (every stream gets the same number of samples per call)."""

    def __init__(self, backend, mel_fn: Callable, embed_fn: Callable, n_streams: int = 1, sr: int = 16000, mel_bins: int = 32,
                 emb_dim: int = 96):
        self.backend, self.mel_fn, self.embed_fn, self.S, self.sr = backend, mel_fn, embed_fn, int(n_streams), sr
        self.bins, self.D = int(mel_bins), int(emb_dim)
        self.raw_max = sr * 10
        self.melspectrogram_max_len = 10 * 97
        self.feature_buffer_max_len = 120
        backend.emb_open(self.S, self.bins, self.D, self.melspectrogram_max_len, self.feature_buffer_max_len)
        self.reset()

    # the models, with the reference's shaping (one call per stream keeps a host mel model's batch semantics trivial)
    def _mel_raw(self, x) -> np.ndarray:
        """float32 [S, n] -> RAW mel [S, frames, bins] (x/10 + 2 is applied on the device)"""
        out = [np.squeeze(self.mel_fn(np.asarray(x[s], np.float32)[None])) for s in range(x.shape[0])]
        return np.ascontiguousarray(np.stack(out), np.float32)

    def _embed(self, windows: np.ndarray) -> np.ndarray:
        """[S, W, 76, bins] -> [S, W, D]"""
        S, W = windows.shape[:2]
        e = np.asarray(self.embed_fn(windows.reshape(S * W, WINDOW_FRAMES, self.bins, 1).astype(np.float32)))
        return np.ascontiguousarray(e.reshape(S, W, -1), np.float32)

    def reset(self):
        self._raw = np.zeros((self.S, 0), np.float64)
        self.accumulated_samples = 0
        self.raw_data_remainder = np.zeros((self.S, 0))
        self.backend.emb_reset()
        # the reference warms the feature buffer with embeddings of 4 s of random noise (:112,121)
        noise = np.random.randint(-1000, 1000, (self.S, 16000 * 4)).astype(np.int16)
        spec = self._mel_raw(noise.astype(np.float32)) / 10 + 2
        wins = self.backend.emb_window_batch(spec)
        self.backend.emb_push_features(self._embed(wins))

    @property
    def feature_buffer(self) -> np.ndarray:
        f = self.backend.emb_get_features(self.feature_buffer_max_len)
        return f[0] if self.S == 1 else f

    @property
    def melspectrogram_frames(self) -> int:
        return self.backend.emb_state()[0]

    def _buffer_raw(self, x):
        self._raw = np.concatenate([self._raw, np.asarray(x, np.float64)], axis=1)[:, -self.raw_max:]

    def _streaming_features(self, x: np.ndarray) -> int:
        """AudioFeatures.py:406-449 as arithmetic on a sample counter: samples are committed to the raw buffer in whole 80 ms
        chunks once at least one chunk is pending (the tail is carried to the next call), and every committed run of whole
        chunks is turned into mel frames, windows and feature rows on the device."""
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None]
        if x.shape[0] != self.S:
            raise ValueError(f"expected audio for {self.S} stream(s), got {x.shape}")
        if self.raw_data_remainder.shape[1]:
            x = np.concatenate((self.raw_data_remainder, x), axis=1)
        pending = self.accumulated_samples + x.shape[1]
        take = x.shape[1] - (pending % CHUNK if pending >= CHUNK else 0)
        self._buffer_raw(x[:, :take])
        self.raw_data_remainder = x[:, take:]
        self.accumulated_samples += take
        n_chunks, partial = divmod(self.accumulated_samples, CHUNK)
        if n_chunks == 0 or partial:
            return self.accumulated_samples
        if self._raw.shape[1] < 400:
            raise ValueError("The number of input frames must be at least 400 samples @ 16khz (25 ms)!")
        self.backend.emb_push_mel(self._mel_raw(self._raw[:, -self.accumulated_samples - MEL_CONTEXT:]), raw=True)
        wins = self.backend.emb_windows(n_chunks)                           # [S, n_valid, 76, bins], oldest chunk first
        if wins.shape[1]:
            self.backend.emb_push_features(self._embed(wins))
        done, self.accumulated_samples = self.accumulated_samples, 0
        return done

    def __call__(self, x):
        return self._streaming_features(x)

    def get_features(self, n_feature_frames: int = 16, start_ndx: int = -1) -> np.ndarray:
        if start_ndx != -1:
            buf = self.backend.emb_get_features(self.feature_buffer_max_len)
            end = start_ndx + int(n_feature_frames) if start_ndx + n_feature_frames != 0 else buf.shape[1]
            return buf[:, start_ndx:end, :].astype(np.float32)
        return self.backend.emb_get_features(int(n_feature_frames))

    def scores(self):
        """(logits [S], probs [S]): the head on the last T feature rows of every stream, features never leave the GPU."""
        return self.backend.emb_forward()

    def embed_clips(self, x: np.ndarray, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """int16 [N, samples] -> float32 [N, (frames-76)//8+1, D]: -80 padding and window gather on the device."""
        specs = [np.squeeze(self.mel_fn(np.asarray(s, np.float32)[None])).astype(np.float32) for s in x]
        # the reference transforms each spectrogram (x/10 + 2, :146) and THEN pads with -80 (:221): one device pass
        mel = self.backend.emb_pad_batch(specs, pad=-80.0, raw=True)
        wins = self.backend.emb_window_batch(mel)                            # raises below 76 frames like the reference (:252-253)
        out = [self._embed(wins[i:i + 1])[0] for i in range(wins.shape[0])]
        return np.stack(out).astype(np.float32)

    def close(self):
        self.backend.emb_close()
