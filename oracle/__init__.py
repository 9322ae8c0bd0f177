"""CPU oracle for the PCM -> log-mel -> head -> logit hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy restatement of the
reference's algorithm (nanowakeword @ v3.0.0) and exists to *check* the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; nothing under ``nanowakeword_amd/`` does,
and the product path raises if the HIP library is missing instead of falling
back to this code.

Parity pinning: the reference ships no tests or golden vectors (SURVEY.md §0
fact 2), so the oracle is pinned against outputs of the reference itself,
generated in the build container by ``tools/make_goldens.py`` (which imports
``/root/reference`` read-only) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every oracle function against them.

Third-party arithmetic not present under /root/reference (restated from the
published algorithm, call sites cited in each function):
  * torchaudio >=2.8,<2.9 (pyproject.toml:50): MelSpectrogram defaults, Hann
    periodic window, ``melscale_fbanks(norm=None, mel_scale="htk")``,
    ``AmplitudeToDB(stype="power", top_db=None)``.
  * onnxruntime CPU kernels: replaced by float32 numpy evaluation of the same
    graph (conv1d-DFT, matmul, conv2d, ...).
The embedding-mode ONNX models (melspectrogram.onnx / embedding_model.onnx)
are un-vendored binaries and are NOT restated: parity unpinned and out of the
parity claim for that mode.
"""
from .frontend import (default_tables, dft_bases, frame_count, frame_signal,  # noqa: F401
                       mel_power, logmel_db, frontend_logmel)
from .heads import head_forward, model_forward, sigmoid  # noqa: F401
