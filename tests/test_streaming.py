"""Batched streaming (SURVEY §8f row 2 / BASELINE config C4): StreamBatch == S independent interpreters."""
import numpy as np
import pytest

from nanowakeword_amd.interpreter import HipInterpreter, StreamBatch
from nanowakeword_amd.synth import synth_pcm
from test_interpreter import ScriptedSession


class FakeBackend:
    """CPU stand-in for HipModel's stream_* API with the scripted scoring rule (ring logic in numpy)."""

    def stream_open(self, S, W, hop):
        self.S, self.W, self.hop = S, W, hop
        self.stream_reset()

    def stream_reset(self):
        self.buf = np.zeros((self.S, self.W), np.int16); self.filled = 0

    def stream_push(self, chunk):
        self.buf = np.concatenate([self.buf, chunk], axis=1)[:, -self.W:]
        self.filled += self.hop
        if self.filled < self.W:
            z = np.zeros(self.S, np.float32)
            return z, z
        x = self.buf.astype(np.float32) / np.float32(32768.0)
        p = np.clip(np.abs(x).mean(axis=1) * 4.0, 0.0, 1.0).astype(np.float32)
        return p, p

    def stream_close(self):
        pass


@pytest.mark.parametrize("kw,ikw", [({}, {}),
                                    ({"patience": 3, "threshold": 0.5}, {"patience": {"wake": 3}, "threshold": {"wake": 0.5}}),
                                    ({"debounce_time": 0.5, "threshold": 0.5}, {"debounce_time": 0.5, "threshold": {"wake": 0.5}})])
def test_streambatch_equals_independent_interpreters_cpu(kw, ikw):
    S, hop = 5, 1280
    streams = np.stack([np.concatenate([synth_pcm("noise", 1, 32000, seed=s)[0] * (1 + s % 3), synth_pcm("loud", 1, 16000, seed=40 + s)[0]])
                        for s in range(S)]).astype(np.int16)
    sb = StreamBatch(FakeBackend(), S, 16000, hop)
    its = [HipInterpreter({"wake": ScriptedSession()}) for _ in range(S)]
    for i in range(0, streams.shape[1] - hop + 1, hop):
        got = sb.push(np.ascontiguousarray(streams[:, i:i + hop]), **kw)
        want = np.array([it.predict(streams[s, i:i + hop], **ikw).score for s, it in enumerate(its)], np.float32)
        raw = np.array([it.raw_scores["wake"] for it in its], np.float32)
        assert np.allclose(got, want, atol=1e-6), (i, got, want)
        assert np.allclose(sb.raw_scores, raw, atol=1e-6)
    sb.reset()
    assert sb.history.shape[0] == 0 and not sb.post_processed_scores.any()
    with pytest.raises(ValueError):
        sb.push(np.zeros((S, hop), np.int16), patience=2)


@pytest.mark.gpu
def test_streambatch_on_device_rings(golden_frontend):
    """Device rings + fused forward vs S independent single-stream interpreters on the same HIP model:
    bit-identical scores (batch invariance), over 10 s of 80 ms hops (BASELINE config C4 shape, CRNN-GRU head)."""
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel, HipSession
    from nanowakeword_amd.synth import synth_state_dict
    g = golden_frontend
    cfg = HeadConfig("crnn", (101, 64))
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g["fb64"])
    S, hop, n_hops = 6, 1280, 40
    streams = np.stack([synth_pcm("speechlike", 1, hop * n_hops, seed=s)[0] for s in range(S)])
    sb = StreamBatch(m, S, 16000, hop)
    batched = np.stack([sb.push(np.ascontiguousarray(streams[:, i * hop:(i + 1) * hop])) for i in range(n_hops)])
    assert m.stream_filled() == hop * n_hops
    m2 = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g["fb64"])
    sess = HipSession(m2, mode="e2e", clip_samples=16000)
    for s in range(S):
        it = HipInterpreter({"model": sess})
        single = np.array([it.predict(streams[s, i * hop:(i + 1) * hop]).score for i in range(n_hops)], np.float32)
        assert np.array_equal(single, batched[:, s]), s
    assert not batched[:12].any() and batched[17:].any()          # 12.5 hops to fill the window, then 5 zeroed predictions
    with pytest.raises(ValueError):
        sb.push(np.zeros((S, hop + 8), np.int16))
    sb.reset()
    assert m.stream_filled() == 0
    sb.close(); m.close(); m2.close()
