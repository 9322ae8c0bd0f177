"""ctypes binding of libnwwhip.so (include/nww.h).  No CPU fallback: if the HIP library is absent
or cannot be loaded, importing callers get an ImportError that says how to build it."""
from __future__ import annotations

import ctypes as C
import os

from .config import FrontendConfig, HeadConfig, HEAD_CODE, ACT_CODE

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NWW_LIB_PATH") or os.path.join(_PKG, "libnwwhip.so")   # override: A/B builds only

NWW_OK = 0
ERR_NAMES = {1: "INVALID", 2: "MISSING", 3: "SHAPE", 4: "HIP", 5: "STATE", 6: "UNSUPPORTED"}


class NwwConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop_length", C.c_int32),
        ("n_mels", C.c_int32), ("center", C.c_int32),
        ("f_min", C.c_float), ("f_max", C.c_float), ("amin", C.c_float), ("db_multiplier", C.c_float),
        ("head_type", C.c_int32), ("in_rows", C.c_int32), ("in_cols", C.c_int32),
        ("layer_dim", C.c_int32), ("n_blocks", C.c_int32), ("embedding_dim", C.c_int32), ("activation", C.c_int32),
        ("n_crnn_channels", C.c_int32), ("crnn_channels", C.c_int32 * 4),
        ("conformer_d_model", C.c_int32), ("conformer_n_head", C.c_int32),
        ("mel_major_features", C.c_int32),
        ("conv_arith", C.c_int32),
        ("crnn_rnn_lstm", C.c_int32),
        ("act_dtype", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


_lib = None

# every symbol include/nww.h declares (tests check the library exports all of them)
SYMBOLS = [
    "nww_default_config", "nww_create", "nww_destroy", "nww_last_error", "nww_load_tensor", "nww_num_tensors",
    "nww_tensor_info", "nww_finalize", "nww_num_frames", "nww_frontend", "nww_frontend_ex", "nww_forward_pcm",
    "nww_forward_features", "nww_forward_features_ex", "nww_frontend_dev", "nww_forward_pcm_dev",
    "nww_forward_features_dev", "nww_reserve", "nww_describe_plan", "nww_set_profiling", "nww_get_profile",
    "nww_stream_open", "nww_stream_push", "nww_stream_push_dev", "nww_stream_reset", "nww_stream_close",
    "nww_stream_filled", "nww_version",
    "nww_emb_open", "nww_emb_reset", "nww_emb_close", "nww_emb_state", "nww_emb_push_mel", "nww_emb_windows",
    "nww_emb_push_features", "nww_emb_get_features", "nww_emb_forward", "nww_emb_window_batch", "nww_emb_pad_batch",
    "nww_comm_unique_id", "nww_comm_init", "nww_comm_destroy", "nww_all_gather_logits", "nww_forward_pcm_gather_dev",
    "nww_forward_pcm_gather_async_dev", "nww_gather_fence", "nww_gather_overlap_ms", "nww_feature_clamp",
]


HIP_RUNTIME = "system"      # which libamdhip64 this process runs on (set by load_library; for logs and bug reports)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 under the system library's soname, and whichever
    copy is loaded first serves every later user of that soname: loaded after this library, torch would run on the system runtime it was not
    built against (seen: `RuntimeError: No HIP GPUs are available` from torch.cuda on a ROCm 7.2 image with a rocm7.0 wheel), while this library
    runs on either.  So when a torch installation is present, its copy is loaded first - located without importing torch."""
    global HIP_RUNTIME
    if os.environ.get("NWW_SHARE_TORCH_HIP", "1") == "0":      # opt-out: a process that never imports torch may want the system runtime
        HIP_RUNTIME = "system (NWW_SHARE_TORCH_HIP=0)"
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for loc in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                HIP_RUNTIME = cand
            except OSError as e:
                HIP_RUNTIME = f"system (loading {cand} failed: {e})"
            return


def load_library():
    """Load libnwwhip.so once. Raises ImportError (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m nanowakeword_amd.build` "
            "(or __graft_entry__.build()). nanowakeword_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, f32p, i16p = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int16)
    lib.nww_default_config.argtypes = [C.POINTER(NwwConfig)]; lib.nww_default_config.restype = None
    lib.nww_create.argtypes = [C.POINTER(NwwConfig), C.POINTER(vp)]; lib.nww_create.restype = C.c_int
    lib.nww_destroy.argtypes = [vp]; lib.nww_destroy.restype = C.c_int
    lib.nww_last_error.argtypes = [vp]; lib.nww_last_error.restype = C.c_char_p
    lib.nww_load_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32, i32]; lib.nww_load_tensor.restype = C.c_int
    lib.nww_num_tensors.argtypes = [vp]; lib.nww_num_tensors.restype = C.c_int
    lib.nww_tensor_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(i32)]
    lib.nww_tensor_info.restype = C.c_int
    lib.nww_finalize.argtypes = [vp]; lib.nww_finalize.restype = C.c_int
    lib.nww_num_frames.argtypes = [vp, i32]; lib.nww_num_frames.restype = i32
    lib.nww_frontend.argtypes = [vp, vp, i32, i32, vp, C.POINTER(i32)]; lib.nww_frontend.restype = C.c_int
    lib.nww_frontend_ex.argtypes = [vp, vp, i32, i32, vp, vp, C.POINTER(i32)]; lib.nww_frontend_ex.restype = C.c_int
    lib.nww_forward_pcm.argtypes = [vp, vp, i32, i32, vp, vp]; lib.nww_forward_pcm.restype = C.c_int
    lib.nww_forward_features.argtypes = [vp, vp, i32, vp, vp]; lib.nww_forward_features.restype = C.c_int
    lib.nww_forward_features_ex.argtypes = [vp, vp, i32, vp, vp, vp]; lib.nww_forward_features_ex.restype = C.c_int
    lib.nww_frontend_dev.argtypes = [vp, vp, i32, i32, vp, i32, vp]; lib.nww_frontend_dev.restype = C.c_int
    lib.nww_forward_pcm_dev.argtypes = [vp, vp, i32, i32, vp, vp, vp]; lib.nww_forward_pcm_dev.restype = C.c_int
    lib.nww_forward_features_dev.argtypes = [vp, vp, i32, vp, vp, vp]; lib.nww_forward_features_dev.restype = C.c_int
    lib.nww_reserve.argtypes = [vp, i32, i32]; lib.nww_reserve.restype = C.c_int
    lib.nww_describe_plan.argtypes = [vp, C.c_char_p, i32]; lib.nww_describe_plan.restype = C.c_int
    lib.nww_set_profiling.argtypes = [vp, i32]; lib.nww_set_profiling.restype = C.c_int
    lib.nww_get_profile.argtypes = [vp, f32p, C.POINTER(i32), C.POINTER(i32)]; lib.nww_get_profile.restype = C.c_int
    lib.nww_stream_open.argtypes = [vp, i32, i32, i32]; lib.nww_stream_open.restype = C.c_int
    lib.nww_stream_push.argtypes = [vp, vp, vp, vp]; lib.nww_stream_push.restype = C.c_int
    lib.nww_stream_push_dev.argtypes = [vp, vp, vp, vp, vp]; lib.nww_stream_push_dev.restype = C.c_int
    lib.nww_stream_reset.argtypes = [vp]; lib.nww_stream_reset.restype = C.c_int
    lib.nww_stream_close.argtypes = [vp]; lib.nww_stream_close.restype = C.c_int
    lib.nww_stream_filled.argtypes = [vp]; lib.nww_stream_filled.restype = C.c_int64
    lib.nww_version.argtypes = []; lib.nww_version.restype = C.c_char_p
    i32p = C.POINTER(i32)
    lib.nww_emb_open.argtypes = [vp, i32, i32, i32, i32, i32]; lib.nww_emb_open.restype = C.c_int
    lib.nww_emb_reset.argtypes = [vp]; lib.nww_emb_reset.restype = C.c_int
    lib.nww_emb_close.argtypes = [vp]; lib.nww_emb_close.restype = C.c_int
    lib.nww_emb_state.argtypes = [vp, i32p, i32p]; lib.nww_emb_state.restype = C.c_int
    lib.nww_emb_push_mel.argtypes = [vp, vp, i32, i32, i32]; lib.nww_emb_push_mel.restype = C.c_int
    lib.nww_emb_windows.argtypes = [vp, i32, vp, i32, i32p]; lib.nww_emb_windows.restype = C.c_int
    lib.nww_emb_push_features.argtypes = [vp, vp, i32, i32]; lib.nww_emb_push_features.restype = C.c_int
    lib.nww_emb_get_features.argtypes = [vp, i32, vp, i32, i32p]; lib.nww_emb_get_features.restype = C.c_int
    lib.nww_emb_forward.argtypes = [vp, vp, vp]; lib.nww_emb_forward.restype = C.c_int
    lib.nww_emb_window_batch.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32p]; lib.nww_emb_window_batch.restype = C.c_int
    lib.nww_emb_pad_batch.argtypes = [vp, vp, i32p, i32, i32, i32, C.c_float, i32, vp]; lib.nww_emb_pad_batch.restype = C.c_int
    lib.nww_comm_unique_id.argtypes = [vp]; lib.nww_comm_unique_id.restype = C.c_int
    lib.nww_comm_init.argtypes = [vp, i32, i32, vp]; lib.nww_comm_init.restype = C.c_int
    lib.nww_comm_destroy.argtypes = [vp]; lib.nww_comm_destroy.restype = C.c_int
    lib.nww_all_gather_logits.argtypes = [vp, vp, vp, i32, vp]; lib.nww_all_gather_logits.restype = C.c_int
    lib.nww_forward_pcm_gather_dev.argtypes = [vp, vp, i32, i32, vp, vp]; lib.nww_forward_pcm_gather_dev.restype = C.c_int
    lib.nww_forward_pcm_gather_async_dev.argtypes = [vp, vp, i32, i32, vp, vp]; lib.nww_forward_pcm_gather_async_dev.restype = C.c_int
    lib.nww_gather_fence.argtypes = [vp, vp]; lib.nww_gather_fence.restype = C.c_int
    lib.nww_feature_clamp.argtypes = [vp]; lib.nww_feature_clamp.restype = C.c_float
    lib.nww_gather_overlap_ms.argtypes = [vp, C.POINTER(C.c_float)]; lib.nww_gather_overlap_ms.restype = C.c_int
    _lib = lib
    return lib


ARITH_CODE = {None: 0, "default": 0, "f32": 1, "f16x3": 3, "bf16x6": 6, "bf16x9": 9}
ACT_DTYPE_CODE = {None: 0, "f32": 0, "bf16": 1, "f16": 2}          # nww_config.act_dtype: storage of the activations between kernels


def make_config(head: HeadConfig, fe: FrontendConfig, device: int = 0, mel_major_features: bool | None = None,
                conv_arith: str | None = None, act_dtype: str | None = None) -> NwwConfig:
    lib = load_library()
    c = NwwConfig()
    lib.nww_default_config(C.byref(c))
    c.device = device
    c.sample_rate, c.n_fft, c.win_length, c.hop_length = fe.sample_rate, fe.n_fft, fe.win_length, fe.hop_length
    c.n_mels, c.center = fe.n_mels, int(bool(fe.center))
    c.f_min, c.f_max, c.amin, c.db_multiplier = fe.f_min, fe.f_max, fe.amin, fe.db_multiplier
    c.head_type = HEAD_CODE[head.model_type]
    c.in_rows, c.in_cols = head.input_shape
    c.layer_dim, c.n_blocks, c.embedding_dim = head.layer_dim, head.n_blocks, head.embedding_dim
    c.activation = ACT_CODE[head.activation]
    ch = list(head.crnn_cnn_channels)
    if len(ch) > 4:
        raise ValueError("crnn_cnn_channels supports at most 4 stages")
    c.n_crnn_channels = len(ch)
    for i, v in enumerate(ch):
        c.crnn_channels[i] = int(v)
    c.conformer_d_model, c.conformer_n_head = head.conformer_d_model, head.conformer_n_head
    c.crnn_rnn_lstm = int(head.model_type == "crnn" and head.crnn_rnn_type == "lstm")
    if mel_major_features is None:
        mel_major_features = head.model_type == "e2e_dnn"
    c.mel_major_features = int(bool(mel_major_features))
    if conv_arith not in ARITH_CODE:
        raise ValueError(f"conv_arith must be one of {sorted(k for k in ARITH_CODE if k)}")
    c.conv_arith = ARITH_CODE[conv_arith]
    if act_dtype not in ACT_DTYPE_CODE:
        raise ValueError("act_dtype must be 'f32', 'bf16' or 'f16'")
    c.act_dtype = ACT_DTYPE_CODE[act_dtype]
    return c
