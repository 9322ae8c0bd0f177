"""Bounded, seeded runs of the fuzzers and the small-call soak (tools/fuzz_frontend.py, tools/fuzz_heads.py,
tools/stress_small_calls.py) inside `pytest -m gpu`: the tools that found real bugs (an inline-asm MFMA hazard among them) now run
wherever the GPU suite runs.  About a minute together; the tools themselves take any case count and seed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_frontend_bounded():
    import fuzz_frontend
    lines = []
    bad = fuzz_frontend.run(n_cases=14, seed=4, max_batch=40, log=lines.append)
    assert bad == 0, "\n".join(lines)


def test_fuzz_heads_bounded():
    import fuzz_heads
    lines = []
    worst, ran = fuzz_heads.run(n_cases=28, seed=4, log=lines.append)
    assert ran >= 14 and worst <= 1e-4, "\n".join(lines)


@pytest.mark.parametrize("head", ["cnn", "dnn", "crnn"])
def test_small_call_soak_bounded(head):
    """random B = 1..20 host-pointer calls (zero-copy staging, completion word) == the bulk kernels' logits, bit for bit"""
    import stress_small_calls
    lines = []
    bad = stress_small_calls.run(n_calls=2500, head=head, seed=4, log=lambda *a: lines.append(" ".join(str(x) for x in a)))
    assert bad == 0, "\n".join(lines)
