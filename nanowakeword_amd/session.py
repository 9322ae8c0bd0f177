"""Host-side mirror of the reference's session boundary, on top of the libnwwhip C-ABI.

``HipModel``   : create / load_state_dict / finalize / forward - one handle on one GPU.
``HipSession`` : duck-types ``onnxruntime.InferenceSession`` exactly as the reference uses it
                 (``get_inputs()[0].name/.shape`` and ``run(None, {"input": x}) -> [probs (B,1,1)]``;
                 reference: nanowakeword/interpreter/nanointerpreter.py:165-167,177-178,677,681,783;
                 in-tree precedent of a substitute backend: remote_verifier.py:490-648 ``_RemoteSession``).

Error behaviour follows the reference's conventions: ValueError for bad inputs,
KeyError/ValueError for state_dict problems, RuntimeError for device failures.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np

from . import _lib
from .config import FrontendConfig, HeadConfig, param_spec


class NwwError(RuntimeError):
    pass


_EXC = {1: ValueError, 2: KeyError, 3: ValueError, 4: NwwError, 5: NwwError, 6: NotImplementedError}


def _as_f32(a) -> np.ndarray:
    if hasattr(a, "detach"):           # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def torchaudio_tables(fe: FrontendConfig):
    """Hann window and mel filterbank exactly as torchaudio builds them (float32 torch ops):
    torch.hann_window(win_length) and torchaudio.functional.melscale_fbanks(norm=None,
    mel_scale="htk") - the tables T.MelSpectrogram carries at architectures.py:830-836.
    The C library's built-in tables evaluate the same formulas in double precision, which
    differs from this float32 evaluation by up to 1e-5 per coefficient (up to ~4e-3 dB on narrow low-frequency filters); use these
    (or the exported model's own buffers) when bit-level agreement with the reference matters."""
    import math
    import torch
    window = torch.hann_window(fe.win_length)
    n_freqs = fe.n_fft // 2 + 1
    all_freqs = torch.linspace(0, fe.sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (fe.f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (fe.f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, fe.n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    return window.numpy().astype(np.float32), fb.numpy().astype(np.float32)


class HipModel:
    """One finalized head (+ frontend) on one GPU behind the C-ABI."""

    def __init__(self, head: HeadConfig, frontend: Optional[FrontendConfig] = None, device: int = 0,
                 state_dict: Optional[Mapping] = None, window=None, mel_fb=None, tables: str = "torchaudio",
                 conv_arith: Optional[str] = None, act_dtype: Optional[str] = None):
        """act_dtype="bf16": the BcResNet head stores the activations between its kernels as bf16 (BASELINE config 3 as written;
        float32 products and accumulation; logits then agree with the float32 reference to ~1e-2 instead of 1e-4 - 0.2 on all-zero
        PCM).  act_dtype="f16": the same tensors as scaled binary16 (same bytes, 11 significant bits: ~5e-3 on every test clip)."""
        self.lib = _lib.load_library()      # ImportError if the HIP extension is missing - no fallback
        self.head = head
        self.fe = frontend or FrontendConfig()
        self.device = device
        self.act_dtype = act_dtype
        self._h = C.c_void_p()
        cfg = _lib.make_config(head, self.fe, device, conv_arith=conv_arith, act_dtype=act_dtype)      # None: library defaults
        rc = self.lib.nww_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self.lib.nww_last_error(None).decode()
            self._h = C.c_void_p()
            raise _EXC.get(rc, NwwError)(msg)
        self.finalized = False
        if window is None and mel_fb is None and tables == "torchaudio":
            try:
                window, mel_fb = torchaudio_tables(self.fe)
            except ImportError:       # no torch on this host: the C library's double-precision tables (<= 1e-5 per coefficient away)
                window = mel_fb = None
        if window is not None:
            self._load("frontend.window", _as_f32(window))
        if mel_fb is not None:
            self._load("frontend.mel_fb", _as_f32(mel_fb))
        if state_dict is not None:
            self.load_state_dict(state_dict)
            self.finalize()

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise _EXC.get(rc, NwwError)(self.lib.nww_last_error(self._h).decode())

    def _load(self, key: str, arr: np.ndarray):
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        self._check(self.lib.nww_load_tensor(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim, 0))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.nww_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def required_tensors(self):
        out = {}
        key, nd = C.c_char_p(), C.c_int32()
        shape = (C.c_int64 * 4)()
        for i in range(self.lib.nww_num_tensors(self._h)):
            self._check(self.lib.nww_tensor_info(self._h, i, C.byref(key), shape, C.byref(nd)))
            out[key.value.decode()] = tuple(int(shape[d]) for d in range(nd.value))
        return out

    def load_state_dict(self, sd: Mapping, strict: bool = True):
        """Accepts Model.state_dict() of the reference (torch tensors or numpy arrays)."""
        spec = param_spec(self.head)
        for k, v in sd.items():
            if k.endswith("num_batches_tracked") or k.startswith("model.mel_spec."):
                continue
            if k not in spec:
                if strict:
                    raise KeyError(f"Unexpected key(s) in state_dict: '{k}'")
                continue
            self._load(k, _as_f32(v))
        # exported E2E models carry their own frontend tables (ONNXSafeMelSpectrogram buffers)
        if "model.mel_spec.mel_fb" in sd:
            self._load("frontend.mel_fb", _as_f32(sd["model.mel_spec.mel_fb"]))
        if "model.mel_spec.real_basis" in sd:
            rb = _as_f32(sd["model.mel_spec.real_basis"])          # [201,1,400]; row k=0 is the padded window
            n_fft, wl = self.fe.n_fft, self.fe.win_length
            pad = (n_fft - wl) // 2
            self._load("frontend.window", np.ascontiguousarray(rb.reshape(rb.shape[0], -1)[0, pad:pad + wl]))
        return self

    def finalize(self):
        self._check(self.lib.nww_finalize(self._h))
        self.finalized = True
        return self

    # ------------------------------------------------------------------ host-array entry points
    def num_frames(self, n_samples: int) -> int:
        return int(self.lib.nww_num_frames(self._h, int(n_samples)))

    @staticmethod
    def _pcm(pcm) -> np.ndarray:
        if not isinstance(pcm, np.ndarray):
            raise ValueError("Input audio `x` must be a Numpy array.")     # nanointerpreter.py:628-629
        if pcm.dtype != np.int16:
            raise ValueError("PCM must be int16")
        if pcm.ndim == 1:
            pcm = pcm[None]
        if pcm.ndim != 2:
            raise ValueError("PCM must have shape (B, N) or (N,)")
        return np.ascontiguousarray(pcm)

    def frontend(self, pcm, return_power: bool = False):
        """int16 [B,N] -> log-mel dB float32 [B, n_mels, frames] (+ mel power)."""
        pcm = self._pcm(pcm)
        B, N = pcm.shape
        T = self.num_frames(N)
        if T <= 0:
            raise ValueError(f"Input clip of {N} samples is too short (n_fft={self.fe.n_fft}, center={self.fe.center})")
        out = np.empty((B, self.fe.n_mels, T), np.float32)
        pw = np.empty_like(out) if return_power else None
        fr = C.c_int32()
        self._check(self.lib.nww_frontend_ex(self._h, pcm.ctypes.data_as(C.c_void_p), B, N, out.ctypes.data_as(C.c_void_p),
                                             pw.ctypes.data_as(C.c_void_p) if return_power else None, C.byref(fr)))
        assert fr.value == T
        return (out, pw) if return_power else out

    def forward_pcm(self, pcm):
        """int16 [B,N] -> (logits [B], probs [B])."""
        pcm = self._pcm(pcm)
        B, N = pcm.shape
        logits, probs = np.empty(B, np.float32), np.empty(B, np.float32)
        self._check(self.lib.nww_forward_pcm(self._h, pcm.ctypes.data_as(C.c_void_p), B, N,
                                             logits.ctypes.data_as(C.c_void_p), probs.ctypes.data_as(C.c_void_p)))
        return logits, probs

    @property
    def feature_clamp(self) -> float:
        """+-bound the plan assumes on (and clamps) the head's input features; 0.0 when nothing is clamped (nww_feature_clamp)"""
        return float(self.lib.nww_feature_clamp(self._h))

    def forward_features(self, feats, return_embedding: bool = False):
        """float32 [B,T,F] -> (logits [B], probs [B][, embedding [B,E]])."""
        feats = np.ascontiguousarray(np.asarray(feats), dtype=np.float32)
        if feats.ndim == 2:
            feats = feats[None]
        if feats.ndim != 3 or tuple(feats.shape[1:]) != tuple(self.head.input_shape):
            raise ValueError(f"features must have shape (B, {self.head.input_shape[0]}, {self.head.input_shape[1]}), got {feats.shape}")
        B = feats.shape[0]
        clamp = self.feature_clamp
        if clamp > 0.0 and feats.size and float(np.abs(feats).max()) > clamp:
            import warnings
            why = "16-bit activation storage (act_dtype)" if self.act_dtype not in (None, "f32") else "the default arithmetic (conv_arith='f16x3')"
            warnings.warn(f"features reach {float(np.abs(feats).max()):.4g}: {why} clamps the head input to "
                          f"+-{clamp:g} (log-mel dB never gets there; the reference does not clamp) - pass conv_arith='bf16x6' and act_dtype=None for unbounded features",
                          RuntimeWarning, stacklevel=2)
        logits, probs = np.empty(B, np.float32), np.empty(B, np.float32)
        emb = np.empty((B, self.head.embedding_dim), np.float32) if return_embedding else None
        self._check(self.lib.nww_forward_features_ex(self._h, feats.ctypes.data_as(C.c_void_p), B,
                                                     logits.ctypes.data_as(C.c_void_p), probs.ctypes.data_as(C.c_void_p),
                                                     emb.ctypes.data_as(C.c_void_p) if return_embedding else None))
        return (logits, probs, emb) if return_embedding else (logits, probs)

    # ------------------------------------------------------------------ device-pointer entry points (torch tensors as plumbing)
    def reserve(self, B: int, N: int = 0):
        self._check(self.lib.nww_reserve(self._h, int(B), int(N)))

    def forward_pcm_dev(self, pcm_ptr: int, B: int, N: int, logits_ptr: int, probs_ptr: int = 0, stream: int = 0):
        """Raw device pointers (e.g. torch.Tensor.data_ptr()); enqueues on `stream`, does not synchronise."""
        self._check(self.lib.nww_forward_pcm_dev(self._h, C.c_void_p(pcm_ptr), B, N, C.c_void_p(logits_ptr),
                                                 C.c_void_p(probs_ptr) if probs_ptr else None,
                                                 C.c_void_p(stream) if stream else None))

    def frontend_dev(self, pcm_ptr: int, B: int, N: int, out_ptr: int, frames_major: bool = False, stream: int = 0):
        self._check(self.lib.nww_frontend_dev(self._h, C.c_void_p(pcm_ptr), B, N, C.c_void_p(out_ptr), int(frames_major),
                                              C.c_void_p(stream) if stream else None))

    def forward_features_dev(self, feats_ptr: int, B: int, logits_ptr: int, probs_ptr: int = 0, stream: int = 0):
        self._check(self.lib.nww_forward_features_dev(self._h, C.c_void_p(feats_ptr), B, C.c_void_p(logits_ptr),
                                                      C.c_void_p(probs_ptr) if probs_ptr else None,
                                                      C.c_void_p(stream) if stream else None))

    # ------------------------------------------------------------------ batched streaming (rings on the device)
    def stream_open(self, n_streams: int, window_samples: int = 16000, hop_samples: int = 1280):
        self._check(self.lib.nww_stream_open(self._h, int(n_streams), int(window_samples), int(hop_samples)))
        self._stream = (int(n_streams), int(window_samples), int(hop_samples))

    def stream_push(self, chunk):
        """chunk int16 [S, hop] -> (logits [S], probs [S]); zeros until every stream has a full window."""
        S, W, hop = self._stream
        chunk = self._pcm(chunk)
        if chunk.shape != (S, hop):
            raise ValueError(f"chunk must have shape ({S}, {hop}), got {chunk.shape}")
        logits, probs = np.empty(S, np.float32), np.empty(S, np.float32)
        self._check(self.lib.nww_stream_push(self._h, chunk.ctypes.data_as(C.c_void_p), logits.ctypes.data_as(C.c_void_p),
                                             probs.ctypes.data_as(C.c_void_p)))
        return logits, probs

    def stream_push_dev(self, chunk_ptr: int, logits_ptr: int, probs_ptr: int = 0, stream: int = 0):
        self._check(self.lib.nww_stream_push_dev(self._h, C.c_void_p(chunk_ptr), C.c_void_p(logits_ptr),
                                                 C.c_void_p(probs_ptr) if probs_ptr else None,
                                                 C.c_void_p(stream) if stream else None))

    def stream_reset(self):
        self._check(self.lib.nww_stream_reset(self._h))

    def stream_filled(self) -> int:
        return int(self.lib.nww_stream_filled(self._h))

    def stream_close(self):
        self._check(self.lib.nww_stream_close(self._h))

    # ------------------------------------------------------------------ embedding-mode preprocessor state (nww_emb_*)
    def emb_open(self, n_streams: int = 1, mel_bins: int = 32, emb_dim: int = 96, mel_cap: int = 970, feat_cap: int = 120):
        self._check(self.lib.nww_emb_open(self._h, int(n_streams), int(mel_bins), int(emb_dim), int(mel_cap), int(feat_cap)))
        self._emb = (int(n_streams), int(mel_bins), int(emb_dim))

    def emb_reset(self):
        self._check(self.lib.nww_emb_reset(self._h))

    def emb_close(self):
        self._check(self.lib.nww_emb_close(self._h))

    def emb_state(self):
        a, b = C.c_int32(), C.c_int32()
        self._check(self.lib.nww_emb_state(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def emb_push_mel(self, mel, raw: bool = True):
        """mel float32 [S, n_frames, bins]: the mel model's output for the newest audio; raw applies x/10 + 2."""
        S, bins, _ = self._emb
        mel = _as_f32(mel)
        if mel.ndim != 3 or mel.shape[0] != S or mel.shape[2] != bins:
            raise ValueError(f"mel must have shape ({S}, n_frames, {bins}), got {mel.shape}")
        self._check(self.lib.nww_emb_push_mel(self._h, mel.ctypes.data_as(C.c_void_p), mel.shape[1], 0, int(bool(raw))))

    def emb_windows(self, n_chunks: int) -> np.ndarray:
        S, bins, _ = self._emb
        out = np.empty((S, int(n_chunks), 76, bins), np.float32)
        nv = C.c_int32()
        self._check(self.lib.nww_emb_windows(self._h, int(n_chunks), out.ctypes.data_as(C.c_void_p), 0, C.byref(nv)))
        return np.ascontiguousarray(out.reshape(-1)[:S * nv.value * 76 * bins].reshape(S, nv.value, 76, bins))

    def emb_push_features(self, emb):
        S, _, D = self._emb
        emb = _as_f32(emb)
        if emb.ndim != 3 or emb.shape[0] != S or emb.shape[2] != D:
            raise ValueError(f"embeddings must have shape ({S}, k, {D}), got {emb.shape}")
        if emb.shape[1]:
            self._check(self.lib.nww_emb_push_features(self._h, emb.ctypes.data_as(C.c_void_p), emb.shape[1], 0))

    def emb_get_features(self, n_frames: int) -> np.ndarray:
        S, _, D = self._emb
        out = np.empty((S, int(n_frames), D), np.float32)
        n = C.c_int32()
        self._check(self.lib.nww_emb_get_features(self._h, int(n_frames), out.ctypes.data_as(C.c_void_p), 0, C.byref(n)))
        return np.ascontiguousarray(out.reshape(-1)[:S * n.value * D].reshape(S, n.value, D))

    def emb_forward(self):
        """(logits [S], probs [S]) of the head on get_features(T) of every stream; features stay on the device."""
        S = self._emb[0]
        logits, probs = np.empty(S, np.float32), np.empty(S, np.float32)
        self._check(self.lib.nww_emb_forward(self._h, logits.ctypes.data_as(C.c_void_p), probs.ctypes.data_as(C.c_void_p)))
        return logits, probs

    def emb_window_batch(self, mel) -> np.ndarray:
        """mel float32 [B, F, bins] -> windows [B, (F-76)//8+1, 76, bins] (AudioFeatures._get_embeddings_batch windowing)."""
        mel = _as_f32(mel)
        B, F, bins = mel.shape
        if F < 76:
            raise ValueError("Embedding model requires the input melspectrograms to have at least 76 frames")
        W = (F - 76) // 8 + 1
        out = np.empty((B, W, 76, bins), np.float32)
        nw = C.c_int32()
        self._check(self.lib.nww_emb_window_batch(self._h, mel.ctypes.data_as(C.c_void_p), B, F, bins, out.ctypes.data_as(C.c_void_p), 0, C.byref(nw)))
        assert nw.value == W
        return out

    def emb_pad_batch(self, specs, pad: float = -80.0, raw: bool = False) -> np.ndarray:
        """list of float32 [frames_i, bins] -> [B, max frames, bins] padded with `pad` (AudioFeatures.py:213-225)."""
        specs = [_as_f32(s) for s in specs]
        bins = specs[0].shape[1]
        frames = np.array([s.shape[0] for s in specs], np.int32)
        Fmax = int(frames.max())
        packed = np.ascontiguousarray(np.concatenate(specs, axis=0))
        out = np.empty((len(specs), Fmax, bins), np.float32)
        self._check(self.lib.nww_emb_pad_batch(self._h, packed.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.POINTER(C.c_int32)),
                                               len(specs), bins, Fmax, float(pad), int(bool(raw)), out.ctypes.data_as(C.c_void_p)))
        return out

    # ------------------------------------------------------------------ multi-GPU gather through the C-ABI (RCCL)
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id; rank 0 creates it and hands it to the other ranks (any transport)."""
        lib = _lib.load_library()
        buf = C.create_string_buffer(128)
        rc = lib.nww_comm_unique_id(buf)
        if rc != 0:
            raise NwwError(lib.nww_last_error(None).decode())
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._check(self.lib.nww_comm_init(self._h, int(rank), int(world), C.create_string_buffer(unique_id, 128)))
        self._comm = (int(rank), int(world))

    def comm_destroy(self):
        self._check(self.lib.nww_comm_destroy(self._h))

    def all_gather_logits_dev(self, send_ptr: int, recv_ptr: int, count: int, stream: int = 0):
        self._check(self.lib.nww_all_gather_logits(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count),
                                                   C.c_void_p(stream) if stream else None))

    def forward_pcm_gather_dev(self, pcm_ptr: int, B: int, N: int, all_logits_ptr: int, stream: int = 0):
        """this rank's B clips -> all ranks' logits [world, B] at all_logits_ptr; kernels + RCCL all-gather on one stream"""
        self._check(self.lib.nww_forward_pcm_gather_dev(self._h, C.c_void_p(pcm_ptr), int(B), int(N), C.c_void_p(all_logits_ptr),
                                                        C.c_void_p(stream) if stream else None))

    def forward_pcm_gather_async_dev(self, pcm_ptr: int, B: int, N: int, all_logits_ptr: int, stream: int = 0):
        """the same step with the all-gather on the handle's own stream behind an event (the next step's kernels do not wait for it);
        alternate two all_logits buffers, gather_fence(stream) before reading them"""
        self._check(self.lib.nww_forward_pcm_gather_async_dev(self._h, C.c_void_p(pcm_ptr), int(B), int(N), C.c_void_p(all_logits_ptr),
                                                              C.c_void_p(stream) if stream else None))

    def gather_fence(self, stream: int = 0):
        self._check(self.lib.nww_gather_fence(self._h, C.c_void_p(stream) if stream else None))

    def gather_overlap_ms(self) -> float:
        """ms from the latest asynchronous step's start to the end of the previous step's gather (> 0: they overlapped)"""
        ms = C.c_float(0.0)
        self._check(self.lib.nww_gather_overlap_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def set_profiling(self, enable=True):
        """True/1: HIP events around every launch of every forward; n > 1: of every n-th forward only; False: off."""
        self._check(self.lib.nww_set_profiling(self._h, int(enable)))

    def get_profile(self):
        """[(launch name, total ms, launches)] accumulated since set_profiling(True); HIP events on the launch stream."""
        names = self.describe_plan().strip().split("\n")
        n = C.c_int32(len(names) + 8)
        ms = (C.c_float * n.value)()
        cnt = (C.c_int32 * n.value)()
        self._check(self.lib.nww_get_profile(self._h, ms, cnt, C.byref(n)))
        return [(names[i] if i < len(names) else f"#{i}", float(ms[i]), int(cnt[i])) for i in range(n.value)]

    def describe_plan(self) -> str:
        buf = C.create_string_buffer(16384)
        self._check(self.lib.nww_describe_plan(self._h, buf, len(buf)))
        return buf.value.decode()


class _Input:
    """Stand-in for onnxruntime NodeArg (cf. _FakeInput, remote_verifier.py:595-603)."""

    def __init__(self, name, shape):
        self.name, self.shape, self.type = name, shape, "tensor(float)"


class HipSession:
    """Drop-in for the ONNX session the reference interpreter holds per model.

    mode="e2e":       input (B,1,N) or (B,N) float32 PCM/32768 (nanointerpreter.py:750,771-775) or int16.
    mode="features":  input (B,T,F) float32.
    run() returns [probabilities (B,1,1) float32] like the exported InferenceWrapper
    (_export/onnx.py:164-172); run_logits() returns the pre-sigmoid logits.
    """

    def __init__(self, model: HipModel, mode: str = "e2e", clip_samples: int = 16000, input_ndim: int = 3,
                 name: str = "hip_model"):
        if mode not in ("e2e", "features"):
            raise ValueError("mode must be 'e2e' or 'features'")
        if not model.finalized:
            raise NwwError("HipSession needs a finalized HipModel")
        self.model, self.mode, self.clip_samples, self.input_ndim = model, mode, int(clip_samples), int(input_ndim)
        self._model_filename = name + ".hip"
        self.name = name
        self.metadata = {"mode": mode}
        self.accepts_int16 = True       # HipInterpreter then skips the float32 round trip of nanointerpreter.py:750
        if mode == "e2e":
            T = model.num_frames(self.clip_samples)
            rows, cols = (model.fe.n_mels, T) if model.head.model_type == "e2e_dnn" else (T, model.fe.n_mels)
            if (rows, cols) != tuple(model.head.input_shape):
                raise ValueError(f"clip_samples={clip_samples} gives ({rows},{cols}) features, head expects {model.head.input_shape}")

    def get_inputs(self):
        if self.mode == "e2e":
            shape = [None, 1, self.clip_samples] if self.input_ndim == 3 else [None, self.clip_samples]
        else:
            shape = [None, self.model.head.input_shape[0], self.model.head.input_shape[1]]
        return [_Input("input", shape)]

    def get_outputs(self):
        return [_Input("output", [None, 1, 1])]

    def _to_pcm(self, x: np.ndarray) -> np.ndarray:
        if x.ndim == 3:
            if x.shape[1] != 1:
                raise ValueError("e2e input must have shape (B,1,N) or (B,N)")
            x = x[:, 0, :]
        if x.ndim != 2:
            raise ValueError("e2e input must have shape (B,1,N) or (B,N)")
        if x.dtype == np.int16:
            return np.ascontiguousarray(x)
        xf = np.asarray(x, dtype=np.float32) * np.float32(32768.0)
        xi = np.rint(xf)
        if not np.array_equal(xi, xf) or xi.min(initial=0) < -32768 or xi.max(initial=0) > 32767:
            raise ValueError("float PCM must be int16/32768.0 exactly (as NanoInterpreter builds it); "
                             "pass the int16 samples for anything else")
        return np.ascontiguousarray(xi.astype(np.int16))

    def _forward(self, input_feed):
        if not isinstance(input_feed, dict) or "input" not in input_feed:
            raise ValueError("input_feed must be {'input': ndarray}")
        x = input_feed["input"]
        if not isinstance(x, np.ndarray):
            raise ValueError("Input must be a Numpy array.")
        if self.mode == "e2e":
            return self.model.forward_pcm(self._to_pcm(x))
        return self.model.forward_features(x)

    def run(self, output_names, input_feed):
        _, probs = self._forward(input_feed)
        return [probs.reshape(-1, 1, 1)]

    def run_logits(self, input_feed):
        logits, _ = self._forward(input_feed)
        return logits.reshape(-1, 1)
