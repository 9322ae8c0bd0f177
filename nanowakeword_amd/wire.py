"""RemoteVerifier wire format in front of a HIP session - messages only, no transport.

The reference's edge<->server protocol (reference: nanowakeword/interpreter/remote_verifier.py:89-105,147-158,
415-455): every request starts with a 1-byte tag,
    0x01 features  "<Biii" (tag, batch, time, feat) + batch*time*feat float32
    0x02 mel       "<Biii" (tag, batch, frames, mel_bins) + float32            (needs the opaque embedding model: not served)
    0x03 audio     "<Bi"   (tag, n_samples) + n_samples int16
and every reply is JSON ``{"score": <float>}``.  The WebSocket server, TLS and auth layers are out of scope
(SURVEY.md §2 #7,#8); what belongs to the hot path is turning MANY clients' requests into ONE GPU batch:
``MicroBatcher`` decodes requests, groups them, scores each group with a single session call and returns the
per-request replies the reference server would have sent one by one.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, List, Tuple

import numpy as np

TAG_FEATURES, TAG_MEL, TAG_AUDIO = 0x01, 0x02, 0x03


def encode_features(features: np.ndarray) -> bytes:
    b, t, f = features.shape
    return struct.pack("<Biii", TAG_FEATURES, b, t, f) + np.ascontiguousarray(features, dtype=np.float32).tobytes()


def encode_audio(audio: np.ndarray) -> bytes:
    return struct.pack("<Bi", TAG_AUDIO, len(audio)) + np.ascontiguousarray(audio, dtype=np.int16).tobytes()


def decode_message(message: bytes) -> Tuple[str, np.ndarray]:
    """-> ("features", float32 [B,T,F]) | ("mel", float32 [B,frames,mel]) | ("audio", int16 [n])."""
    if not message:
        raise ValueError("empty message")
    tag = message[0]
    if tag in (TAG_FEATURES, TAG_MEL):
        if len(message) < 13:
            raise ValueError("truncated feature header")
        b, t, f = struct.unpack("<iii", message[1:13])
        n = b * t * f * 4
        if b <= 0 or t <= 0 or f <= 0 or len(message) < 13 + n:
            raise ValueError("truncated or malformed feature payload")
        arr = np.frombuffer(message[13:13 + n], dtype=np.float32).reshape(b, t, f)
        return ("features" if tag == TAG_FEATURES else "mel"), arr
    if tag == TAG_AUDIO:
        if len(message) < 5:
            raise ValueError("truncated audio header")
        n = struct.unpack("<i", message[1:5])[0]
        if n < 0 or len(message) < 5 + 2 * n:
            raise ValueError("truncated or malformed audio payload")
        return "audio", np.frombuffer(message[5:5 + 2 * n], dtype=np.int16)
    raise ValueError(f"unknown wire tag 0x{tag:02x}")


def encode_reply(score: float) -> str:
    return json.dumps({"score": float(score)})


def decode_reply(reply: str) -> float:
    return float(json.loads(reply)["score"])


class MicroBatcher:
    """Collect requests from many edge clients, score them as one GPU batch.

    ``feature_session`` : session protocol object for 0x01 requests (feature mode, input (B,T,F)).
    ``audio_backend``   : object with ``forward_pcm(int16 [B,N]) -> (logits, probs)`` (a HipModel) for 0x03
                          requests in the e2e pipeline.  Per client, the reference server's state machine
                          (remote_verifier.py:377,436-453) is mirrored: a deque of ``clip_samples`` samples plus an
                          ``accumulated`` counter that is reset after every scoring, so once the deque is full a
                          client is scored once per ``clip_samples`` of NEW audio and gets 0.0 in between (a client
                          that sends one full clip per message - the reference's own client - is scored every time).
                          One deliberate deviation: while the deque is still filling the reference scores the
                          partial clip it holds (``accumulated >= len(buffer)`` is true on the first message), which
                          a fixed-length e2e model cannot run; here that case replies 0.0 and resets the counter
                          exactly as the reference's scoring branch would.
    """

    def __init__(self, feature_session=None, audio_backend=None, clip_samples: int = 16000):
        self.feature_session, self.audio_backend, self.clip_samples = feature_session, audio_backend, int(clip_samples)
        self._pending: List[Tuple[int, str, str, np.ndarray]] = []
        self._windows: Dict[str, np.ndarray] = {}
        self._acc: Dict[str, int] = {}      # samples since the last scoring (reference: e2e_state["accumulated"])
        self._fill: Dict[str, int] = {}     # len(e2e_state["buffer"])
        self._ticket = 0

    def submit(self, client_id: str, message: bytes) -> int:
        kind, arr = decode_message(message)
        if kind == "mel":
            raise NotImplementedError("0x02 needs the un-vendored speech-embedding model (SURVEY.md §8c)")
        if kind == "features" and self.feature_session is None:
            raise ValueError("no feature session configured")
        if kind == "audio" and self.audio_backend is None:
            raise ValueError("no audio backend configured")
        self._ticket += 1
        self._pending.append((self._ticket, client_id, kind, arr))
        return self._ticket

    def drop_client(self, client_id: str):
        self._windows.pop(client_id, None)
        self._acc.pop(client_id, None)
        self._fill.pop(client_id, None)

    def flush(self) -> Dict[int, str]:
        """Score everything submitted since the last flush; returns {ticket: reply JSON}."""
        replies: Dict[int, str] = {}
        feats = [(t, a) for t, _, k, a in self._pending if k == "features"]
        if feats:
            groups: Dict[Tuple[int, int], List[Tuple[int, np.ndarray]]] = {}
            for t, a in feats:
                groups.setdefault(a.shape[1:], []).append((t, a))
            for _, items in groups.items():
                x = np.ascontiguousarray(np.concatenate([a for _, a in items], axis=0))
                probs = np.asarray(self.feature_session.run(None, {"input": x})[0]).reshape(-1)
                off = 0
                for t, a in items:
                    # the reference replies float(out[0].item()) for its batch-1 requests (:425-426); for b > 1 we reply the first row
                    replies[t] = encode_reply(probs[off])
                    off += a.shape[0]
        audio = [(t, c, a) for t, c, k, a in self._pending if k == "audio"]
        if audio:
            ready: List[Tuple[int, np.ndarray]] = []
            for t, c, a in audio:                      # requests of one client are applied in arrival order
                w = self._windows.setdefault(c, np.zeros(self.clip_samples, np.int16))
                n = len(a)
                if n >= self.clip_samples:
                    w[:] = a[n - self.clip_samples:]
                elif n:
                    w[:-n] = w[n:]
                    w[-n:] = a
                self._fill[c] = min(self._fill.get(c, 0) + n, self.clip_samples)
                self._acc[c] = self._acc.get(c, 0) + n
                if self._acc[c] >= self._fill[c]:          # remote_verifier.py:447-450
                    self._acc[c] = 0
                    if self._fill[c] == self.clip_samples:
                        ready.append((t, w.copy()))
                    else:
                        replies[t] = encode_reply(0.0)     # partial clip (see class docstring)
                else:
                    replies[t] = encode_reply(0.0)
            if ready:
                _, probs = self.audio_backend.forward_pcm(np.stack([w for _, w in ready]))
                for (t, _), p in zip(ready, probs):
                    replies[t] = encode_reply(p)
        self._pending.clear()
        return replies
