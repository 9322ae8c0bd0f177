"""WindowedFeatures (embedding-mode preprocessor interface/windowing) vs traces of the reference AudioFeatures
driven by the same deterministic fake models (tests/golden/audio_features_trace.json). CPU-only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from fake_models import fake_embed, fake_mel
from nanowakeword_amd.audio_features import WindowedFeatures
from nanowakeword_amd.interpreter import HipInterpreter
from nanowakeword_amd.synth import synth_pcm


@pytest.fixture(scope="module")
def trace():
    return json.load(open(os.path.join(GOLDEN, "audio_features_trace.json")))


def test_streaming_state_machine_matches_reference(trace):
    np.random.seed(trace["seed"])
    wf = WindowedFeatures(fake_mel, fake_embed)
    stream = synth_pcm("noise", 1, 16000 * 3, seed=21)[0]
    pos = 0
    for row in trace["rows"]:
        n = row["n"]
        ret = wf(stream[pos:pos + n]); pos += n
        assert int(ret) == row["ret"], row
        assert list(wf.feature_buffer.shape) == row["feat_shape"] and list(wf.melspectrogram_buffer.shape) == row["mel_shape"]
        assert wf.accumulated_samples == row["acc"] and wf.raw_data_remainder.shape[0] == row["rem"]
        assert abs(float(np.asarray(wf.feature_buffer[-1], np.float64).sum()) - row["feat_last_sum"]) <= 1e-6 * max(1, abs(row["feat_last_sum"]))
        assert abs(float(np.asarray(wf.feature_buffer, np.float64).sum()) - row["feat_sum"]) <= 1e-6 * abs(row["feat_sum"])
    gf = wf.get_features(16)
    assert list(gf.shape) == trace["get_features_shape"] and gf.dtype == np.float32
    assert abs(float(gf.astype(np.float64).sum()) - trace["get_features_sum"]) <= 1e-6 * abs(trace["get_features_sum"])


def test_embed_clips_and_shapes_match_reference(trace):
    np.random.seed(trace["seed"])
    wf = WindowedFeatures(fake_mel, fake_embed)
    clips = synth_pcm("noise", 3, 32000, seed=22)
    emb = wf.embed_clips(clips, batch_size=2, ncpu=1)
    assert list(emb.shape) == trace["embed_clips_shape"] == [3, 16, 96]      # the reference's default (16, 96) features
    assert abs(float(emb.astype(np.float64).sum()) - trace["embed_clips_sum"]) <= 1e-6 * abs(trace["embed_clips_sum"])
    assert np.allclose(emb[1, 3, :8], trace["embed_clips_row"], rtol=1e-6, atol=1e-6)
    np.random.seed(99)
    assert list(wf.get_embedding_shape(2.0)) == trace["embedding_shape_2s"]
    with pytest.raises(ValueError, match="76 frames"):
        wf.embed_clips(synth_pcm("noise", 2, 8000, seed=1))                     # < 76 mel frames
    wf.reset()
    with pytest.raises(ValueError, match="400 samples"):
        wf._streaming_mel(100)


def test_plugs_into_interpreter_like_audiofeatures():
    class Feat:
        metadata = {"mode": "features"}

        def get_inputs(self):
            return [type("I", (), {"name": "input", "shape": [None, 16, 96]})()]

        def run(self, names, feed):
            x = feed["input"]
            assert x.shape == (1, 16, 96) and x.dtype == np.float32
            return [np.array([[[float(1 / (1 + np.exp(-x.mean())))]]], np.float32)]
    np.random.seed(5)
    it = HipInterpreter({"kw": Feat()}, preprocessor=WindowedFeatures(fake_mel, fake_embed))
    stream = synth_pcm("noise", 1, 1280 * 10, seed=3)[0]
    scores = [it.predict(stream[i:i + 1280]).score for i in range(0, len(stream), 1280)]
    assert scores[:5] == [0.0] * 5 and all(0.0 < s < 1.0 for s in scores[5:])    # warm feature buffer: only the 5-prediction zeroing
    r = it.predict(stream[:500])                                                # < 1280 prepared samples: last scores
    assert r.score == scores[-1]
    it.reset()
    assert it.preprocessor.accumulated_samples == 0
