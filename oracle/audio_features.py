"""Host mirror of the reference's ``AudioFeatures`` state machine (nanowakeword/data/AudioFeatures.py) around two pluggable
models - the checker for ``nanowakeword_amd.audio_features.DeviceWindowedFeatures`` (whose buffers live on the GPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): a statement-level restatement of the reference's embedding-mode bookkeeping -
the ``x/10 + 2`` mel transform (:124,146), 76-frame windows every 8 frames (:168-179,261-272), the 80 ms streaming state machine
with remainder carry (:406-449), the 10 s raw / 970-frame mel / 120-row feature caps (:106-112,397-398,446), ``get_features``
(:451-457), ``embed_clips`` with -80 padding (:188-227,297-322) and the error conventions (:252-253,390-391).  Pinned by traces of
the reference class driven by deterministic fake ORT sessions (tests/golden/audio_features_trace.json, tools/make_goldens.py).
The two ONNX models themselves are un-vendored binaries: ``mel_fn`` / ``embed_fn`` stand where the sessions stood.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

WINDOW_FRAMES = 76          # AudioFeatures.py:168
WINDOW_STEP = 8
CHUNK = 1280                # 80 ms
MEL_CONTEXT = 160 * 3       # extra samples fed to the mel model per update (:394)


class WindowedFeatures:
    def __init__(self, mel_fn: Callable, embed_fn: Callable, sr: int = 16000):
        self.mel_fn, self.embed_fn, self.sr = mel_fn, embed_fn, sr
        self.raw_max = sr * 10
        self.melspectrogram_max_len = 10 * 97
        self.feature_buffer_max_len = 120
        self.reset()

    # ------------------------------------------------------------------ model calls with the reference's shaping
    def _mel(self, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.float32)
        if x.ndim < 2:
            x = x[None]
        spec = np.squeeze(self.mel_fn(x))
        return spec / 10 + 2                                   # melspec_transform default (:124)

    def _embed(self, windows: np.ndarray) -> np.ndarray:
        return np.asarray(self.embed_fn(windows)).squeeze()    # (:103) (W,96), or (96,) for a single window

    def _get_embeddings(self, x: np.ndarray) -> np.ndarray:
        spec = self._mel(x)
        wins = [spec[i:i + WINDOW_FRAMES] for i in range(0, spec.shape[0], WINDOW_STEP)
                if spec[i:i + WINDOW_FRAMES].shape[0] == WINDOW_FRAMES]
        return self._embed(np.expand_dims(np.array(wins), axis=-1).astype(np.float32))

    def get_embedding_shape(self, audio_length: float, sr: int = 16000):
        x = (np.random.uniform(-1, 1, int(audio_length * sr)) * 32767).astype(np.int16)
        return self._get_embeddings(x).shape

    # ------------------------------------------------------------------ state
    def reset(self):
        self._raw = np.zeros(0, np.float64)                    # last <= 10 s of samples (a deque of Python numbers in the reference)
        self.melspectrogram_buffer = np.ones((WINDOW_FRAMES, 32))
        self.accumulated_samples = 0
        self.raw_data_remainder = np.empty(0)
        # the reference warms the feature buffer with embeddings of 4 s of random noise (:112,121)
        self.feature_buffer = self._get_embeddings(np.random.randint(-1000, 1000, 16000 * 4).astype(np.int16))

    @property
    def raw_data_buffer(self):
        return self._raw

    def _buffer_raw(self, x):
        self._raw = np.concatenate([self._raw, np.asarray(x, np.float64)])[-self.raw_max:]

    def _streaming_mel(self, n_samples: int):
        if len(self._raw) < 400:
            raise ValueError("The number of input frames must be at least 400 samples @ 16khz (25 ms)!")
        new = self._mel(list(self._raw[-n_samples - MEL_CONTEXT:]))
        self.melspectrogram_buffer = np.vstack((self.melspectrogram_buffer, new))[-self.melspectrogram_max_len:]

    def _streaming_features(self, x: np.ndarray) -> int:
        processed = 0
        if self.raw_data_remainder.shape[0] != 0:
            x = np.concatenate((self.raw_data_remainder, x))
            self.raw_data_remainder = np.empty(0)
        total = self.accumulated_samples + x.shape[0]
        if total >= CHUNK:
            rem = total % CHUNK
            if rem:
                even = x[:-rem]
                self._buffer_raw(even)
                self.accumulated_samples += len(even)
                self.raw_data_remainder = x[-rem:]
            else:
                self._buffer_raw(x)
                self.accumulated_samples += x.shape[0]
        else:
            self.accumulated_samples += x.shape[0]
            self._buffer_raw(x)
        if self.accumulated_samples >= CHUNK and self.accumulated_samples % CHUNK == 0:
            self._streaming_mel(self.accumulated_samples)
            for i in range(self.accumulated_samples // CHUNK - 1, -1, -1):     # oldest new chunk first
                end = len(self.melspectrogram_buffer) - WINDOW_STEP * i
                win = self.melspectrogram_buffer[end - WINDOW_FRAMES:end].astype(np.float32)[None, :, :, None]
                if win.shape[1] == WINDOW_FRAMES:
                    self.feature_buffer = np.vstack((self.feature_buffer, self._embed(win)))
            processed = self.accumulated_samples
            self.accumulated_samples = 0
        if self.feature_buffer.shape[0] > self.feature_buffer_max_len:
            self.feature_buffer = self.feature_buffer[-self.feature_buffer_max_len:, :]
        return processed if processed != 0 else self.accumulated_samples

    def __call__(self, x):
        return self._streaming_features(x)

    def get_features(self, n_feature_frames: int = 16, start_ndx: int = -1) -> np.ndarray:
        if start_ndx != -1:
            end = start_ndx + int(n_feature_frames) if start_ndx + n_feature_frames != 0 else len(self.feature_buffer)
            return self.feature_buffer[start_ndx:end, :][None].astype(np.float32)
        return self.feature_buffer[int(-1 * n_feature_frames):, :][None].astype(np.float32)

    # ------------------------------------------------------------------ batch path (transform_clips.py:455)
    def embed_clips(self, x: np.ndarray, batch_size: int = 128, ncpu: int = 1) -> np.ndarray:
        """int16 [N, samples] -> float32 [N, (frames-76)//8+1, D]; mel padded to the longest clip with -80 (:221)."""
        specs = []
        for i in range(0, x.shape[0], batch_size):
            specs.extend(np.squeeze(self._mel(s)) for s in x[i:i + batch_size])
        frames = max(s.shape[0] for s in specs)
        mel = np.full((len(specs), frames, specs[0].shape[1]), -80.0, np.float32)
        for i, s in enumerate(specs):
            mel[i, :s.shape[0]] = s
        if mel.shape[1] < WINDOW_FRAMES:
            raise ValueError("Embedding model requires the input melspectrograms to have at least 76 frames")
        n_frames = (mel.shape[1] - WINDOW_FRAMES) // WINDOW_STEP + 1
        out = None
        for n in range(mel.shape[0]):
            wins = np.stack([mel[n, i:i + WINDOW_FRAMES] for i in range(0, mel.shape[1], WINDOW_STEP)
                             if i + WINDOW_FRAMES <= mel.shape[1]])[..., None].astype(np.float32)
            e = np.asarray(self.embed_fn(wins)).reshape(wins.shape[0], -1)
            if out is None:
                out = np.empty((mel.shape[0], n_frames, e.shape[1]), np.float32)
            out[n] = e[:n_frames]
        return out
