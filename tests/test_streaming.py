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


# (head kwargs, frontend (n_mels, center), window samples, hop samples): what a hop keeps differs per row -
#   hop 1280 = 8 frames: log-mel ring + (CRNN) pooled conv-row rings; 640 = 4 frames: one pooled row per hop; 2560: two hops of rows;
#   800 = 5 frames: frontend ring only (conv rows do not line up); 1000: not a whole number of frames -> every hop re-scores the window
_INC_CASES = [
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="crnn", input_shape=(101, 64), crnn_rnn_type="lstm"), (64, True), 16000, 640),
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 2560),
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 800),
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 1000),
    (dict(model_type="crnn", input_shape=(98, 40)), (40, False), 16000, 1280),
    (dict(model_type="crnn", input_shape=(151, 64), layer_dim=64), (64, True), 24000, 1280),
    (dict(model_type="cnn", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="dnn", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="dnn", input_shape=(98, 40)), (40, False), 16000, 1280),
    (dict(model_type="bcresnet", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="conformer", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="gru", input_shape=(101, 64)), (64, True), 16000, 1280),
    (dict(model_type="e2e_dnn", input_shape=(64, 101)), (64, True), 16000, 1280),
    # large hops: the rows a hop recomputes are one tall strip - cut to fit in LDS, or the conv rows are re-scored
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 13440),
    (dict(model_type="crnn", input_shape=(101, 64)), (64, True), 16000, 10880),
    (dict(model_type="crnn", input_shape=(151, 64), layer_dim=64), (64, True), 24000, 16640),
    # four conv stages (the third reads the trunk's rings, the fourth runs as k-split passes); a stack the fused trunk does not take; a padded recurrent width
    (dict(model_type="crnn", input_shape=(101, 64), crnn_cnn_channels=[16, 32, 64, 64]), (64, True), 16000, 1280),
    (dict(model_type="crnn", input_shape=(101, 64), crnn_cnn_channels=[32, 64], layer_dim=96), (64, True), 16000, 1280),
    (dict(model_type="gru", input_shape=(101, 64), layer_dim=100), (64, True), 16000, 640),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(_INC_CASES)))
def test_incremental_hops_equal_window_rescoring(golden_frontend, case):
    """The streaming path keeps per-stream rings of log-mel frames (and, for the CRNN stem, pooled conv rows) and computes per hop only
    what the hop invalidates.  Contract: every hop's logits are BIT-IDENTICAL to scoring that stream's last window from scratch
    (`forward_pcm`) - through ring wrap-arounds, after a reset, and for hops where only part (or none) of the state can be kept."""
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel
    from nanowakeword_amd.synth import synth_state_dict
    g = golden_frontend
    kw, (n_mels, center), W, hop = _INC_CASES[case]
    cfg = HeadConfig(**kw)
    fe = FrontendConfig(n_mels=n_mels, center=center)
    m = HipModel(cfg, fe, state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g["fb64"] if n_mels == 64 else g["fb40"])
    S = 5
    n_hops = (W + hop - 1) // hop + (34 if hop <= 4000 else 7)   # more than two trips round the log-mel ring (13 hops of 8 frames)
    streams = np.stack([synth_pcm("speechlike" if s % 2 else "noise", 1, hop * n_hops, seed=300 + 7 * s + case)[0] for s in range(S)])
    streams[4, hop * 20:hop * 30] = 0                         # a stretch of digital silence (-100 dB floor rows travel through the rings)
    m.stream_open(S, W, hop)
    for rnd in range(2):                                      # second round: after nww_stream_reset the state is rebuilt from the first full window
        hist = np.zeros((S, 0), np.int16)
        for i in range(n_hops if rnd == 0 else (W + hop - 1) // hop + 3):
            chunk = np.ascontiguousarray(streams[:, i * hop:(i + 1) * hop] if rnd == 0 else streams[:, ::-1][:, i * hop:(i + 1) * hop])
            hist = np.concatenate([hist, chunk], axis=1)[:, -W:]
            lg, pr = m.stream_push(chunk)
            if hist.shape[1] < W or m.stream_filled() < W:
                assert not lg.any() and not pr.any()
                continue
            want, wantp = m.forward_pcm(np.ascontiguousarray(hist))
            assert np.array_equal(lg, want), (case, rnd, i, np.abs(lg - want).max())
            assert np.array_equal(pr, wantp)
        m.stream_reset()
    m.stream_close(); m.close()
