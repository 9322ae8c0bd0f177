"""HipInterpreter's predict() state machine vs traces captured from the reference NanoInterpreter
(tests/golden/predict_trace.json, tools/make_goldens.py) using the same scripted session.  CPU-only."""
import numpy as np
import pytest

from nanowakeword_amd.interpreter import DetectionResult, HipInterpreter
from nanowakeword_amd.synth import synth_pcm


class _Inp:
    def __init__(self, name, shape):
        self.name, self.shape = name, shape


class ScriptedSession:
    """Same scoring rule as the fake onnxruntime session the traces were captured with."""

    def __init__(self, clip_samples=16000, ndim=3):
        self.clip_samples, self.ndim, self.calls = clip_samples, ndim, []

    def get_inputs(self):
        return [_Inp("input", [None, 1, self.clip_samples] if self.ndim == 3 else [None, self.clip_samples])]

    def run(self, output_names, feed):
        x = feed["input"]
        assert x.dtype == np.float32 and x.shape[-1] == self.clip_samples and x.shape[0] == 1
        self.calls.append(x.copy())
        s = float(np.clip(np.abs(x).mean() * 4.0, 0.0, 1.0))
        return [np.array([[[s]]], dtype=np.float32)]


def _stream(spec):
    return np.concatenate([synth_pcm(kind, 1, n, seed=seed)[0] for kind, n, seed in spec["parts"]])


@pytest.mark.parametrize("key", ["chunk1280", "chunk4000", "chunk16000", "patience3", "debounce"])
def test_predict_trace_matches_reference(predict_trace, key):
    tr = predict_trace[key]
    stream = _stream(predict_trace["stream_spec"])
    sess = ScriptedSession()
    it = HipInterpreter({"wake": sess})
    assert it.preprocessor is None and it.is_e2e["wake"]
    chunk, kw = tr["chunk"], tr["kw"]
    rows = []
    for i in range(0, len(stream) - chunk + 1, chunk):
        r = it.predict(stream[i:i + chunk], **kw)
        assert isinstance(r, DetectionResult)
        rows.append([float(it.raw_scores["wake"]), float(r.score)])
    ref = [[a, b] for a, b, _ in tr["rows"]]
    assert len(rows) == len(ref)
    assert np.allclose(np.array(rows), np.array(ref), rtol=0, atol=1e-7), key
    assert len(sess.calls) == tr["n_calls"]
    assert abs(float(np.abs(sess.calls[0]).sum()) - tr["first_clip_abs_sum"]) <= 1e-3 * tr["first_clip_abs_sum"]


def test_predict_clip_reset_and_accessors(predict_trace):
    stream = _stream(predict_trace["stream_spec"])
    it = HipInterpreter.load_model(ScriptedSession())
    res = it.predict_clip(stream[:20000])
    assert len(res) == predict_trace["predict_clip_len"] == 1
    assert res[0].score == predict_trace["predict_clip_score"] == 0.0          # first 5 predictions are zeroed
    assert abs(it.raw_scores["model"] - predict_trace["predict_clip_raw"]) < 1e-6
    assert it.score == 0.0 and not it.is_cascade and it.gate_name is None and it.gate_score == 0.0
    assert it.info["loaded_models"] == ["model"] and not it.detected(0.5)
    it.reset()
    assert it.e2e_buffer_samples["model"] == predict_trace["after_reset_buffer"] == 0
    r = res[0]
    assert r.get("model") == r["model"] == 0.0 and "model" in r and "x" not in r and "score=" in repr(r)


def test_errors_follow_reference():
    it = HipInterpreter({"wake": ScriptedSession()})
    with pytest.raises(ValueError, match="Numpy array"):
        it.predict([1, 2, 3])
    x = synth_pcm("loud", 1, 16000)[0]
    for _ in range(6):
        it.predict(x)
    with pytest.raises(ValueError, match="threshold"):
        it.predict(x, patience={"wake": 2})
    with pytest.raises(ValueError, match="cannot be used together"):
        it.predict(x, patience={"wake": 2}, debounce_time=1.0, threshold={"wake": 0.5})
    with pytest.raises(TypeError):
        it.predict_clip(123)
    with pytest.raises(FileNotFoundError):
        HipInterpreter.load_model("/nonexistent/model.nww.npz")
    with pytest.raises(TypeError):
        HipInterpreter.load_model(3.14)
    with pytest.raises(NotImplementedError):
        HipInterpreter({"wake": ScriptedSession()}, vad_threshold=0.5)


def test_cascade_gate_blocks_verifier():
    class Const(ScriptedSession):
        def __init__(self, v):
            super().__init__(); self.v = v

        def run(self, names, feed):
            self.calls.append(1)
            return [np.array([[[self.v]]], np.float32)]
    gate, ver = Const(0.1), Const(0.9)
    gate.name, ver.name = "kw_lite", "kw"
    it = HipInterpreter.load_model(ver, gate_model=gate, gate_threshold=0.3)
    assert it.is_cascade and it.gate_name == "kw_lite" and it.model_name == "kw"
    x = synth_pcm("noise", 1, 16000)[0]
    for _ in range(7):
        r = it.predict(x)
    assert len(ver.calls) == 0 and r.score == 0.0 and it.raw_scores["kw"] == 0.0     # gate below threshold: verifier skipped
    gate.v = 0.8
    r = it.predict(x)
    assert len(ver.calls) == 1 and abs(r.score - 0.9) < 1e-6 and abs(r.gate_score - 0.8) < 1e-6


def test_feature_mode_protocol():
    class Pre:
        def __init__(self):
            self.feature_buffer = np.zeros((0, 96), np.float32); self.n = 0

        def __call__(self, x):
            self.n += len(x)
            k, self.n = divmod(self.n, 1280)
            if k:
                self.feature_buffer = np.vstack([self.feature_buffer, np.ones((k, 96), np.float32)])[-120:]
            return k * 1280

        def get_features(self, n):
            return self.feature_buffer[-n:][None]

        def reset(self):
            self.__init__()

    class Feat:
        metadata = {"mode": "features"}

        def get_inputs(self):
            return [_Inp("input", [None, 16, 96])]

        def run(self, names, feed):
            assert feed["input"].shape == (1, 16, 96)
            return [np.array([[[0.7]]], np.float32)]
    it = HipInterpreter({"m": Feat()}, preprocessor=Pre())
    x = synth_pcm("noise", 1, 1280)[0]
    outs = [it.predict(x).score for _ in range(25)]
    assert outs[:15] == [0.0] * 15                      # feature buffer warming up (needs 16 frames)
    # the 15 warm-up zeros already filled the 5-entry history, so no further zeroing (same as the reference)
    assert all(abs(v - 0.7) < 1e-6 for v in outs[15:])
    assert it.predict(x[:100]).score == pytest.approx(0.7)   # < 1280 prepared samples: last scores returned
