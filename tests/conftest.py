import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that hand torch tensors to the library (RCCL communicator, device pointers) need ONE ROCm stack in the
    # process: importing torch first makes libnwwhip.so bind to the HIP runtime PyTorch ships, like bench.py does.
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def _hip_device_available() -> bool:
    """True when libnwwhip.so loads and nww_create succeeds on device 0 (probed in a subprocess-free, cheap way)."""
    try:
        import ctypes as C
        from nanowakeword_amd import _lib
        from nanowakeword_amd.config import FrontendConfig, HeadConfig
        lib = _lib.load_library()
        h = C.c_void_p()
        cfg = _lib.make_config(HeadConfig("dnn", (16, 96)), FrontendConfig(), 0)
        if lib.nww_create(C.byref(cfg), C.byref(h)) != 0:
            return False
        lib.nww_destroy(h)
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Skip gpu-marked tests on hosts without a HIP device, so a plain `pytest tests` is green on CPU boxes.  When the
    user asks for them explicitly (-m gpu) they run and fail loudly instead: no silent skip on a GPU box."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _hip_device_available():
        return
    skip = pytest.mark.skip(reason="no HIP device (or libnwwhip.so not built): gpu tests need a real MI355X")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_frontend():
    return dict(np.load(os.path.join(GOLDEN, "frontend.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_heads():
    z = np.load(os.path.join(GOLDEN, "heads.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(str(d.pop("meta_json")))
    return d, meta


@pytest.fixture(scope="session")
def predict_trace():
    with open(os.path.join(GOLDEN, "predict_trace.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_heads_r02():
    """round-2 reference goldens (tools/make_goldens.py::make_round2_goldens): LSTM CRNN, other conv stacks, E2E at 1.5 / 2 s"""
    z = np.load(os.path.join(GOLDEN, "heads_r02.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(str(d.pop("meta_json")))
    return d, meta


@pytest.fixture(scope="session")
def golden_heads_r04():
    """round-4 reference goldens (tools/make_goldens.py::make_round4_goldens): the distilled lite gate, DNN inputs that are not a
    multiple of 4, recurrent widths > 256 / not a multiple of 4, Conformer head dims outside the compiled set"""
    z = np.load(os.path.join(GOLDEN, "heads_r04.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(str(d.pop("meta_json")))
    return d, meta


@pytest.fixture(scope="session")
def golden_heads_r06():
    """round-6 reference goldens (tools/make_goldens.py::make_round6_goldens): Conformer d_model 256 / 192 (head dims 64 / 48) and the
    default width at other clip lengths (the one-launch attention module's instances)"""
    z = np.load(os.path.join(GOLDEN, "heads_r06.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(str(d.pop("meta_json")))
    return d, meta


def head_case_names_r06():
    z = np.load(os.path.join(GOLDEN, "heads_r06.npz"), allow_pickle=False)
    return sorted(json.loads(str(z["meta_json"])).keys())


def head_case_names_r04():
    z = np.load(os.path.join(GOLDEN, "heads_r04.npz"), allow_pickle=False)
    return sorted(json.loads(str(z["meta_json"])).keys())


def head_case_names_r02():
    z = np.load(os.path.join(GOLDEN, "heads_r02.npz"), allow_pickle=False)
    return sorted(json.loads(str(z["meta_json"])).keys())


def head_case_names():
    z = np.load(os.path.join(GOLDEN, "heads.npz"), allow_pickle=False)
    return sorted(json.loads(str(z["meta_json"])).keys())
