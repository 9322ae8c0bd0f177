"""Host-side mirror of the reference's inference API for the hot path.

``HipInterpreter`` keeps the public surface of ``nanowakeword.interpreter.NanoInterpreter``
(reference: nanowakeword/interpreter/nanointerpreter.py): ``load_model`` / ``predict`` /
``predict_clip`` / ``reset`` / ``detected`` and the ``score`` / ``gate_score`` / ``verifier_score`` /
``model_name`` / ``gate_name`` / ``is_cascade`` / ``info`` / ``raw_scores`` accessors, plus
``DetectionResult`` (:45-115).  Scores come from any object that follows the session protocol
(``get_inputs()``, ``run(None, {"input": x})``) - normally a ``HipSession`` on an MI355X.

Kept on the host, per stream: the raw-audio window, the first-5-predictions
zeroing (:689-691,789-790), cascade gating (:660-669,758-769) and the patience / debounce filters
(:1034-1064).  Not mirrored (out of scope): VAD, noise reduction, the WebSocket
remote verifier, the microphone loop.
"""
from __future__ import annotations

import math
import os
import wave
from collections import deque
from typing import Dict, List, Mapping, Optional, Union

import numpy as np

N_WARMUP_PREDICTIONS = 5        # nanointerpreter.py:690,789
PREDICTION_HISTORY = 30         # nanointerpreter.py:1002
HOP_SAMPLES = 1280              # 80 ms @ 16 kHz (AudioFeatures.py:414)


class DetectionResult:
    """Attribute + dict-style view of one predict() call (nanointerpreter.py:45-115)."""

    __slots__ = ("scores", "model_name", "gate_name", "threshold")

    def __init__(self, scores: dict, model_name: str, gate_name: Optional[str], threshold: float = 0.0):
        self.scores, self.model_name, self.gate_name, self.threshold = scores, model_name, gate_name, threshold

    @property
    def score(self) -> float:
        return self.scores.get(self.model_name, 0.0)

    @property
    def gate_score(self) -> float:
        return self.scores.get(self.gate_name, 0.0) if self.gate_name else 0.0

    @property
    def detected(self) -> bool:
        return self.score >= self.threshold if self.threshold > 0 else False

    def get(self, model_name: str, default: float = 0.0) -> float:
        return self.scores.get(model_name, default)

    def __getitem__(self, key: str) -> float:
        return self.scores[key]

    def __contains__(self, key: str) -> bool:
        return key in self.scores

    def __repr__(self) -> str:
        out = [f"score={self.score:.4f}"]
        if self.gate_name:
            out.append(f"gate={self.gate_score:.4f}")
        if self.threshold > 0:
            out.append(f"detected={self.detected}")
        return "DetectionResult(" + ", ".join(out) + ")"


class _AudioWindow:
    """Last `size` int16 samples of a stream (the reference keeps a deque(maxlen) of floats, :181,751-756)."""

    def __init__(self, size: int):
        self.size = int(size)
        self.data = np.zeros(self.size, np.int16)
        self.filled = 0          # total samples seen (saturating semantics only matter vs size)

    def push(self, x: np.ndarray):
        n = len(x)
        if n >= self.size:
            self.data[:] = x[n - self.size:]
        elif n:
            self.data[:-n] = self.data[n:]
            self.data[-n:] = x
        self.filled += n

    def clear(self):
        self.data[:] = 0
        self.filled = 0


def _is_e2e_session(session) -> bool:
    """Mode detection (nanointerpreter.py:968-992): explicit metadata first, then the rank heuristic."""
    meta = getattr(session, "metadata", None)
    if isinstance(meta, Mapping) and "mode" in meta:
        return meta["mode"] == "e2e"
    shape = session.get_inputs()[0].shape
    if len(shape) <= 2:
        return True
    return len(shape) == 3 and shape[-1] not in (32, 64, 96)


class HipInterpreter:
    """Stateful wake-word scoring over a stream of int16 chunks (one instance per stream; not thread-safe,
    like the reference)."""

    def __init__(self, sessions: Mapping[str, object], preprocessor=None, vad_threshold: float = 0,
                 enable_noise_reduction: bool = False):
        if not sessions:
            raise ValueError("at least one model session is required")
        if vad_threshold and vad_threshold > 0:
            raise NotImplementedError("VAD post-filter is outside the accelerated path (SURVEY.md §2 #9)")
        if enable_noise_reduction:
            raise NotImplementedError("noise reduction is outside the accelerated path")
        self.models: Dict[str, object] = dict(sessions)
        self.model_input_names = {n: [i.name for i in s.get_inputs()] for n, s in self.models.items()}
        self.model_feature_length = {n: s.get_inputs()[0].shape[1] for n, s in self.models.items()}
        self.class_mapping = {n: {"0": n} for n in self.models}
        self.raw_scores = {n: 0.0 for n in self.models}
        self.post_processed_scores = {n: 0.0 for n in self.models}
        self.is_e2e = {n: _is_e2e_session(s) for n, s in self.models.items()}
        self.e2e_clip_samples, self.e2e_input_ndim, self._windows = {}, {}, {}
        for n, s in self.models.items():
            if self.is_e2e[n]:
                shape = s.get_inputs()[0].shape
                self.e2e_clip_samples[n] = int(shape[-1])
                self.e2e_input_ndim[n] = len(shape)
                self._windows[n] = _AudioWindow(shape[-1])
        self.prediction_buffer: Dict[str, deque] = {}
        all_e2e = all(self.is_e2e.values())
        if not all_e2e and preprocessor is None:
            raise ValueError("feature-input models need a preprocessor object (AudioFeatures protocol: "
                             "__call__, feature_buffer, get_features, reset)")
        self.preprocessor = None if all_e2e else preprocessor
        self.cascade_config: dict = {}
        self.vad_threshold = 0

    # ------------------------------------------------------------------ construction
    @classmethod
    def load_model(cls, model: Union[str, List[str], Mapping[str, object], object, None] = None, cascade: bool = False,
                   gate_model=None, gate_threshold: float = 0.3, device: int = 0, **kwargs) -> "HipInterpreter":
        """Same calling convention as NanoInterpreter.load_model (:310-325).  ``model`` may be a path (or
        list of paths) to the reference's own exported ``*.onnx`` files (weights recovered by
        ``weights.state_dict_from_onnx``), to bundles written by ``weights.save_bundle`` (``*.nww.npz``),
        or ready session objects / a {name: session} mapping.  ``cascade=True`` looks for ``<name>_lite``
        next to the main bundle; ``gate_model`` names the gate explicitly (implies cascade)."""
        from .weights import load_session
        if model is None:
            raise ValueError("`model` is required (remote verifier mode is not part of the accelerated path)")
        sessions: Dict[str, object] = {}
        paths: List[str] = []
        if isinstance(model, Mapping):
            sessions.update(model)
        elif isinstance(model, str):
            paths = [model]
        elif isinstance(model, list) and all(isinstance(m, str) for m in model):
            paths = list(model)
        elif isinstance(model, (list, tuple)):
            for i, s in enumerate(model):
                sessions[getattr(s, "name", f"model{i}")] = s
        elif hasattr(model, "run") and hasattr(model, "get_inputs"):
            sessions[getattr(model, "name", "model")] = model
        else:
            raise TypeError("`model` must be a string, list of strings, a session or a mapping of sessions.")
        for p in paths:
            if not os.path.exists(p):
                raise FileNotFoundError(f"Model file not found: {p}")

        def stem(p):
            b = os.path.basename(p)
            for ext in (".nww.npz", ".npz", ".onnx", ".pt"):
                if b.endswith(ext):
                    return b[:-len(ext)]
            return os.path.splitext(b)[0]

        cascade_cfg: dict = {}
        gate_session = None
        if gate_model is not None and not isinstance(gate_model, str):
            gate_session, gate_name = gate_model, getattr(gate_model, "name", "gate")
        if (cascade or gate_model is not None) and (len(paths) == 1 or (len(sessions) == 1 and gate_session is not None)):
            main_name = stem(paths[0]) if paths else next(iter(sessions))
            if gate_session is not None:
                pass
            elif gate_model is not None:
                if not os.path.exists(gate_model):
                    raise FileNotFoundError(f"The specified gate model does not exist: {gate_model}")
                gate_name, gate_session = stem(gate_model), load_session(gate_model, device=device)
            else:
                d = os.path.dirname(os.path.abspath(paths[0]))
                gate_name = main_name + "_lite"
                cand = [os.path.join(d, gate_name + ext) for ext in (".nww.npz", ".npz", ".onnx")]
                hit = next((c for c in cand if os.path.exists(c)), None)
                gate_session = load_session(hit, device=device) if hit else None    # else: single-model mode (:481-487)
            if gate_session is not None:
                ordered = {gate_name: gate_session}              # the gate is evaluated first
                for p in paths:
                    ordered[stem(p)] = load_session(p, device=device)
                ordered.update(sessions)
                sessions, paths = ordered, []
                cascade_cfg = {"gate": gate_name, "verifier": main_name, "gate_threshold": gate_threshold}
        for p in paths:
            name = stem(p)
            if name not in sessions:
                sessions[name] = load_session(p, device=device)
        inst = cls(sessions, **kwargs)
        inst.cascade_config = cascade_cfg
        return inst

    # ------------------------------------------------------------------ accessors (:196-295)
    @property
    def is_cascade(self) -> bool:
        return bool(self.cascade_config)

    @property
    def model_name(self) -> str:
        return self.cascade_config["verifier"] if self.is_cascade else next(iter(self.models))

    @property
    def gate_name(self) -> Optional[str]:
        return self.cascade_config.get("gate")

    @property
    def gate_score(self) -> float:
        return self.post_processed_scores.get(self.gate_name, 0.0) if self.gate_name else 0.0

    @property
    def verifier_score(self) -> float:
        return self.post_processed_scores.get(self.model_name, 0.0)

    @property
    def score(self) -> float:
        return self.verifier_score

    @property
    def info(self) -> dict:
        return {"model_name": self.model_name, "is_cascade": self.is_cascade, "is_remote": False,
                "gate_name": self.gate_name, "gate_threshold": self.cascade_config.get("gate_threshold"),
                "loaded_models": list(self.models), "score": self.score, "gate_score": self.gate_score,
                "raw_scores": dict(self.raw_scores)}

    def detected(self, threshold: float, model: Optional[str] = None) -> bool:
        return self.post_processed_scores.get(model or self.model_name, 0.0) >= threshold

    def __repr__(self) -> str:
        if self.is_cascade:
            return (f"HipInterpreter(model='{self.model_name}', gate='{self.gate_name}', "
                    f"gate_threshold={self.cascade_config.get('gate_threshold', 0.3)})")
        names = list(self.models)
        return f"HipInterpreter(model='{names[0]}')" if len(names) == 1 else f"HipInterpreter(models={names})"

    # ------------------------------------------------------------------ scoring
    def _history(self, name) -> deque:
        buf = self.prediction_buffer.get(name)
        if buf is None:
            buf = self.prediction_buffer[name] = deque(maxlen=PREDICTION_HISTORY)
        return buf

    def _gate_closed(self, name, current) -> bool:
        c = self.cascade_config
        return bool(c) and name == c["verifier"] and current.get(c["gate"], 0.0) < c["gate_threshold"]

    def _run_e2e(self, name, session) -> float:
        win = self._windows[name]
        clip = win.data
        if getattr(session, "accepts_int16", False):
            x = clip.reshape(1, 1, -1) if self.e2e_input_ndim[name] == 3 else clip.reshape(1, -1)
        else:   # the reference's float path: x/32768.0, (1,1,N) or (1,N)  (:750,771-775)
            f = clip.astype(np.float32) / np.float32(32768.0)
            x = f.reshape(1, 1, -1) if self.e2e_input_ndim[name] == 3 else f.reshape(1, -1)
        return float(session.run(None, {"input": x})[0].item())

    def predict(self, x: np.ndarray, patience: dict = {}, threshold: dict = {}, debounce_time: float = 0.0) -> DetectionResult:
        """One chunk of 16-bit PCM in, per-model scores out (nanointerpreter.py:606-717, 735-814)."""
        if not isinstance(x, np.ndarray):
            raise ValueError("Input audio `x` must be a Numpy array.")
        current: Dict[str, float] = {}
        if self.preprocessor is None:                       # ---- E2E: raw audio windows per model
            xi = x if x.dtype == np.int16 else x.astype(np.int16)
            for name, session in self.models.items():
                win = self._windows[name]
                win.push(xi)
                if win.filled >= win.size:
                    if self._gate_closed(name, current):
                        current[name] = 0.0
                        self.raw_scores[name] = 0.0
                        continue
                    score = self._run_e2e(name, session)
                else:
                    score = 0.0
                self.raw_scores[name] = score
                if len(self.prediction_buffer.get(name, ())) < N_WARMUP_PREDICTIONS:
                    score = 0.0
                current[name] = score
            n_samples = len(x)
        else:                                               # ---- feature mode: preprocessor protocol
            n_samples = self.preprocessor(x)
            if n_samples < HOP_SAMPLES:
                return DetectionResult(dict(self.post_processed_scores), self.model_name, self.gate_name)
            for name, session in self.models.items():
                frames = self.model_feature_length[name]
                if self.preprocessor.feature_buffer.shape[0] < frames or self._gate_closed(name, current):
                    current[name] = 0.0
                    continue
                score = float(session.run(None, {"input": self.preprocessor.get_features(frames)})[0].item())
                self.raw_scores[name] = score
                if len(self.prediction_buffer.get(name, ())) < N_WARMUP_PREDICTIONS:
                    score = 0.0
                current[name] = score
        final = dict(current)
        self._post_filters(final, patience, threshold, debounce_time, n_samples)
        for name, s in final.items():
            self._history(name).append(s)
            self.post_processed_scores[name] = s
        return DetectionResult(dict(final), self.model_name, self.gate_name)

    def _post_filters(self, preds, patience, threshold, debounce_time, n_samples):
        """Patience (N consecutive frames >= threshold) or debounce (suppress repeats), :1034-1064."""
        if not patience and debounce_time <= 0:
            return
        if not threshold:
            raise ValueError("`threshold` must be provided when using `patience` or `debounce_time`.")
        if patience and debounce_time > 0:
            raise ValueError("`patience` and `debounce_time` cannot be used together.")
        for name, s in list(preds.items()):
            if s == 0.0:
                continue
            hist = list(self._history(name))
            if name in patience:
                need = patience[name]
                if len(hist) < need:
                    preds[name] = 0.0
                    continue
                tail = hist[-(need - 1):] if need > 1 else hist      # reference slices buffer[-(need-1):]; need==1 -> [-0:] = all
                hits = sum(1 for v in tail + [s] if v >= threshold[name])
                if hits < need:
                    preds[name] = 0.0
            elif debounce_time > 0 and name in threshold:
                dur = n_samples / 16000.0
                if dur <= 0:
                    continue
                k = int(math.ceil(debounce_time / dur))
                if s >= threshold[name] and any(v >= threshold[name] for v in hist[-k:]):
                    preds[name] = 0.0

    def predict_clip(self, clip: Union[str, np.ndarray], chunk_size: int = HOP_SAMPLES, **kwargs) -> list:
        """Whole clip (path or array) -> list of DetectionResult (:816-833). E2E mode: one prediction."""
        if isinstance(clip, str):
            with wave.open(clip, mode="rb") as f:
                if f.getframerate() != 16000 or f.getsampwidth() != 2 or f.getnchannels() != 1:
                    raise ValueError("Audio clip must be a 16kHz, 16-bit, single-channel WAV file.")
                data = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
        elif isinstance(clip, np.ndarray):
            data = clip
        else:
            raise TypeError("`clip` must be a file path (string) or a numpy array.")
        if self.preprocessor is None:
            return [self.predict(data, **kwargs)]
        return [self.predict(data[i:i + chunk_size], **kwargs) for i in range(0, len(data), chunk_size)]

    def reset(self):
        """New stream: clear histories, windows and scores (:719-733)."""
        self.prediction_buffer.clear()
        if self.preprocessor is not None:
            self.preprocessor.reset()
        for name in self.raw_scores:
            self.raw_scores[name] = 0.0
            self.post_processed_scores[name] = 0.0
        for w in self._windows.values():
            w.clear()

    # reference-compatible view of the e2e buffers
    @property
    def e2e_buffer_samples(self) -> Dict[str, int]:
        return {n: w.filled for n, w in self._windows.items()}


class StreamBatch:
    """S lock-step audio streams scored together on one GPU: the batched form of the E2E branch of
    ``NanoInterpreter.predict`` (nanointerpreter.py:735-814).  The raw-audio windows live in device rings
    (``nww_stream_*``); per-stream filter state stays here on the host, vectorised over streams:
    raw score 0 until a stream has a full window (:785-786), first five predictions zeroed (:789-790),
    patience / debounce (:1034-1064), 30-entry prediction history (:1002).  Every stream receives exactly
    ``hop`` new samples per push (streams never migrate)."""

    def __init__(self, backend, n_streams: int, window_samples: int = 16000, hop_samples: int = HOP_SAMPLES,
                 name: str = "model"):
        self.backend, self.S, self.window, self.hop, self.name = backend, int(n_streams), int(window_samples), int(hop_samples), name
        backend.stream_open(self.S, self.window, self.hop)
        self.reset(_device=False)

    def reset(self, _device: bool = True):
        if _device:
            self.backend.stream_reset()
        self.history = np.zeros((0, self.S), np.float32)          # most recent last, at most PREDICTION_HISTORY rows
        self.raw_scores = np.zeros(self.S, np.float32)
        self.post_processed_scores = np.zeros(self.S, np.float32)

    def push(self, chunk: np.ndarray, patience: int = 0, threshold: float = 0.0, debounce_time: float = 0.0) -> np.ndarray:
        """chunk int16 [S, hop] -> post-processed scores [S] (what DetectionResult.score is per stream)."""
        if not isinstance(chunk, np.ndarray):
            raise ValueError("Input audio `x` must be a Numpy array.")
        _, probs = self.backend.stream_push(chunk)
        raw = probs.astype(np.float32)
        self.raw_scores = raw
        score = raw.copy()
        if self.history.shape[0] < N_WARMUP_PREDICTIONS:
            score[:] = 0.0
        if patience or debounce_time > 0:
            if not threshold:
                raise ValueError("`threshold` must be provided when using `patience` or `debounce_time`.")
            if patience and debounce_time > 0:
                raise ValueError("`patience` and `debounce_time` cannot be used together.")
            nz = score != 0.0
            if patience:
                if self.history.shape[0] < patience:
                    score[nz] = 0.0
                else:
                    tail = self.history[-(patience - 1):] if patience > 1 else self.history
                    hits = (tail >= threshold).sum(axis=0) + (score >= threshold)
                    score[nz & (hits < patience)] = 0.0
            else:
                k = int(math.ceil(debounce_time / (self.hop / 16000.0)))
                recent = (self.history[-k:] >= threshold).any(axis=0) if self.history.shape[0] else np.zeros(self.S, bool)
                score[nz & (score >= threshold) & recent] = 0.0
        self.history = np.vstack([self.history, score[None]])[-PREDICTION_HISTORY:]
        self.post_processed_scores = score
        return score

    def close(self):
        self.backend.stream_close()
