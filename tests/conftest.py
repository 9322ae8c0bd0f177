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


def head_case_names():
    z = np.load(os.path.join(GOLDEN, "heads.npz"), allow_pickle=False)
    return sorted(json.loads(str(z["meta_json"])).keys())
