"""WindowedFeatures (embedding-mode preprocessor interface/windowing) vs traces of the reference AudioFeatures
driven by the same deterministic fake models (tests/golden/audio_features_trace.json). CPU-only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from fake_models import fake_embed, fake_mel
from oracle.audio_features import WindowedFeatures
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


# --------------------------------------------------------------------------------------------- device path (rows a5-a7)
@pytest.mark.gpu
def test_device_windowed_features_match_host_mirror_and_reference_trace(trace):
    """nww_emb_* (device mel ring / windows / feature ring) behind DeviceWindowedFeatures: the same reference trace the
    host mirror is pinned to, then bit-identical buffers against the host mirror over ragged chunk sizes, and the head's
    score straight from the device ring == the head on get_features(T) == the oracle."""
    import oracle
    from nanowakeword_amd.audio_features import DeviceWindowedFeatures
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel
    from nanowakeword_amd.synth import synth_state_dict
    cfg = HeadConfig("dnn", (16, 96))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, tables="builtin")
    np.random.seed(trace["seed"])
    dev = DeviceWindowedFeatures(m, fake_mel, fake_embed)
    stream = synth_pcm("noise", 1, 16000 * 3, seed=21)[0]
    pos = 0
    for row in trace["rows"]:                                   # the reference AudioFeatures' own trace
        n = row["n"]
        ret = dev(stream[pos:pos + n]); pos += n
        assert int(ret) == row["ret"], row
        fb = dev.feature_buffer
        assert list(fb.shape) == row["feat_shape"] and dev.melspectrogram_frames == row["mel_shape"][0]
        assert dev.accumulated_samples == row["acc"] and dev.raw_data_remainder.shape[1] == row["rem"]
        assert abs(float(fb[-1].astype(np.float64).sum()) - row["feat_last_sum"]) <= 1e-6 * max(1, abs(row["feat_last_sum"]))
        assert abs(float(fb.astype(np.float64).sum()) - row["feat_sum"]) <= 1e-6 * abs(row["feat_sum"])
    gf = dev.get_features(16)
    assert list(gf.shape) == trace["get_features_shape"] and gf.dtype == np.float32
    assert abs(float(gf.astype(np.float64).sum()) - trace["get_features_sum"]) <= 1e-6 * abs(trace["get_features_sum"])
    # long run with ragged chunks: mel ring wraps (970 frames), feature ring wraps (120 rows); host mirror in lock step
    np.random.seed(7)
    host = WindowedFeatures(fake_mel, fake_embed)
    np.random.seed(7)
    dev.reset()
    long = synth_pcm("speechlike", 1, 16000 * 14, seed=33)[0]
    rng = np.random.default_rng(3)
    pos = 0
    while pos < len(long):
        n = int(rng.choice([160, 1280, 1280, 2560, 777, 4000, 1280 * 5]))
        a, b = host(long[pos:pos + n]), dev(long[pos:pos + n])
        pos += n
        assert a == b
        assert np.array_equal(host.feature_buffer.astype(np.float32), dev.feature_buffer)
        assert host.melspectrogram_buffer.shape[0] == dev.melspectrogram_frames
    assert host.feature_buffer.shape[0] == 120 and dev.melspectrogram_frames == 970          # both caps reached
    assert np.array_equal(host.get_features(16), dev.get_features(16))
    assert np.array_equal(host.get_features(5, start_ndx=-20), dev.get_features(5, start_ndx=-20))
    logits, probs = dev.scores()
    l2, p2 = m.forward_features(dev.get_features(16))
    assert np.array_equal(logits, l2) and np.array_equal(probs, p2)
    assert np.abs(logits - oracle.model_forward(host.get_features(16), sd, cfg).ravel()).max() <= 1e-4
    # batch path: -80 padding + window gather on the device == the host mirror == the reference trace
    clips = synth_pcm("noise", 3, 32000, seed=22)
    emb = dev.embed_clips(clips, batch_size=2)
    assert list(emb.shape) == trace["embed_clips_shape"]
    assert np.array_equal(emb, host.embed_clips(clips, batch_size=2))
    assert abs(float(emb.astype(np.float64).sum()) - trace["embed_clips_sum"]) <= 1e-6 * abs(trace["embed_clips_sum"])
    ragged = [fake_mel(synth_pcm("noise", 1, n, seed=n)[0].astype(np.float32))[0, 0] for n in (20000, 32000, 16000)]
    padded = m.emb_pad_batch(ragged, pad=-80.0, raw=True)
    assert padded.shape == (3, max(r.shape[0] for r in ragged), 32)
    for i, r in enumerate(ragged):
        assert np.array_equal(padded[i, :r.shape[0]], r / 10 + 2) and np.all(padded[i, r.shape[0]:] == -80.0)
    with pytest.raises(ValueError, match="76 frames"):
        dev.embed_clips(synth_pcm("noise", 2, 8000, seed=1))
    dev.close(); m.close()


@pytest.mark.gpu
def test_device_windowed_features_many_streams_and_interpreter():
    """S = 64 lock-step streams on the device rings == 64 independent host mirrors; and one stream plugged into
    HipInterpreter as its preprocessor (the reference's NanoInterpreter(preprocessor=AudioFeatures) shape)."""
    from nanowakeword_amd.audio_features import DeviceWindowedFeatures
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel, HipSession
    from nanowakeword_amd.synth import synth_state_dict
    cfg = HeadConfig("cnn", (16, 96))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, tables="builtin")
    S = 64
    np.random.seed(11)
    dev = DeviceWindowedFeatures(m, fake_mel, fake_embed, n_streams=S)
    np.random.seed(11)
    warm = np.random.randint(-1000, 1000, (S, 16000 * 4)).astype(np.int16)      # what dev.reset() drew, stream by stream
    hosts = []
    for s in range(S):
        h = WindowedFeatures.__new__(WindowedFeatures)
        h.mel_fn, h.embed_fn, h.sr, h.raw_max, h.melspectrogram_max_len, h.feature_buffer_max_len = fake_mel, fake_embed, 16000, 160000, 970, 120
        h._raw = np.zeros(0, np.float64); h.melspectrogram_buffer = np.ones((76, 32)); h.accumulated_samples = 0
        h.raw_data_remainder = np.empty(0)
        h.feature_buffer = h._get_embeddings(warm[s])
        hosts.append(h)
    streams = np.stack([synth_pcm("speechlike" if s % 2 else "noise", 1, 1280 * 30, seed=200 + s)[0] for s in range(S)])
    for i in range(0, streams.shape[1], 1920):                                 # 1.5 chunks per call: remainder carry every call
        dev(streams[:, i:i + 1920])
        for s in range(S):
            hosts[s](streams[s, i:i + 1920])
    fb = dev.feature_buffer
    for s in range(S):
        assert np.array_equal(fb[s], hosts[s].feature_buffer.astype(np.float32)), s
    logits, probs = dev.scores()
    want, _ = m.forward_features(np.stack([h.get_features(16)[0] for h in hosts]))
    assert np.array_equal(logits, want)
    dev.close()
    # single stream as the interpreter's preprocessor
    np.random.seed(5)
    one = DeviceWindowedFeatures(m, fake_mel, fake_embed)
    it = HipInterpreter({"kw": HipSession(m, mode="features")}, preprocessor=one)
    np.random.seed(5)
    ref_it = HipInterpreter({"kw": HipSession(m, mode="features")}, preprocessor=WindowedFeatures(fake_mel, fake_embed))
    audio = synth_pcm("noise", 1, 1280 * 12, seed=3)[0]
    for i in range(0, len(audio), 1280):
        assert it.predict(audio[i:i + 1280]).score == ref_it.predict(audio[i:i + 1280]).score
    one.close(); m.close()
