"""Wire format byte-compatibility with the reference's encoders + micro-batching semantics. CPU-only (+1 GPU test)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from nanowakeword_amd import wire
from nanowakeword_amd.synth import synth_features, synth_pcm


@pytest.fixture(scope="module")
def ref_msgs():
    return json.load(open(os.path.join(GOLDEN, "wire_messages.json")))


def test_bytes_match_reference_encoders(ref_msgs):
    f = synth_features(2, (16, 96), seed=4)
    a = synth_pcm("noise", 1, 1280, seed=4)[0]
    assert wire.encode_features(f).hex() == ref_msgs["features_hex"]
    assert wire.encode_audio(a).hex() == ref_msgs["audio_hex"]
    assert (wire.TAG_FEATURES, wire.TAG_MEL, wire.TAG_AUDIO) == tuple(ref_msgs["tags"][k] for k in ("features", "mel", "audio"))
    kind, arr = wire.decode_message(bytes.fromhex(ref_msgs["features_hex"]))
    assert kind == "features" and np.array_equal(arr, f)
    kind, arr = wire.decode_message(bytes.fromhex(ref_msgs["audio_hex"]))
    assert kind == "audio" and np.array_equal(arr, a)
    assert wire.encode_reply(0.75) == ref_msgs["reply"] and wire.decode_reply(ref_msgs["reply"]) == 0.75
    for bad in (b"", b"\x09abc", bytes.fromhex(ref_msgs["features_hex"])[:40], bytes.fromhex(ref_msgs["audio_hex"])[:3]):
        with pytest.raises(ValueError):
            wire.decode_message(bad)


class MeanSession:
    calls = 0

    def run(self, names, feed):
        MeanSession.calls += 1
        x = feed["input"]
        return [(1.0 / (1.0 + np.exp(-x.mean(axis=(1, 2))))).astype(np.float32).reshape(-1, 1, 1)]


class MeanAudio:
    calls = 0

    def forward_pcm(self, pcm):
        MeanAudio.calls += 1
        p = np.clip(np.abs(pcm.astype(np.float32) / 32768.0).mean(axis=1) * 4, 0, 1).astype(np.float32)
        return p, p


def test_microbatcher_batches_clients():
    mb = wire.MicroBatcher(feature_session=MeanSession(), audio_backend=MeanAudio(), clip_samples=4000)
    feats = [synth_features(1, (16, 96), seed=s) / 30.0 for s in range(5)]
    tickets = [mb.submit(f"c{i}", wire.encode_features(f)) for i, f in enumerate(feats)]
    other = mb.submit("c9", wire.encode_features(synth_features(1, (8, 32), seed=1)))       # different (T,F): own group
    audio = synth_pcm("noise", 3, 6000, seed=2)
    at = [mb.submit(f"a{i}", wire.encode_audio(audio[i, :1280])) for i in range(3)]
    MeanSession.calls = MeanAudio.calls = 0
    rep = mb.flush()
    assert MeanSession.calls == 2 and MeanAudio.calls == 0                                  # 5 requests -> 1 call (+1 for the odd shape)
    for t, f in zip(tickets, feats):
        assert abs(wire.decode_reply(rep[t]) - float(1 / (1 + np.exp(-f.mean())))) < 1e-6
    assert other in rep and all(wire.decode_reply(rep[t]) == 0.0 for t in at)               # windows not full yet
    # the reference's per-connection state machine (remote_verifier.py:377,445-450), restated: a deque of
    # clip_samples + an `accumulated` counter that is reset after every scoring
    from collections import deque
    buf, acc, scored_at = deque(maxlen=4000), 0, []
    for k in range(0, 9):
        buf.extend(range(1280)); acc += 1280
        if acc >= len(buf):
            acc = 0
            if len(buf) == 4000:
                scored_at.append(k)
    assert scored_at == [4, 8]                       # M + n samples for the first full window, then once per M new samples
    for k in range(1, 9):
        at = [mb.submit(f"a{i}", wire.encode_audio(audio[i % 3, (1280 * k) % 4720:(1280 * k) % 4720 + 1280])) for i in range(3)]
        calls0 = MeanAudio.calls
        rep = mb.flush()
        if k in scored_at:
            assert MeanAudio.calls == calls0 + 1                                            # 3 clients, one batched call
            for i, t in enumerate(at):
                w = np.concatenate([audio[i, (1280 * j) % 4720:(1280 * j) % 4720 + 1280] for j in range(k - 3, k + 1)])[-4000:]
                want = float(np.clip(np.abs(w.astype(np.float32) / 32768.0).mean() * 4, 0, 1))
                assert abs(wire.decode_reply(rep[t]) - want) < 1e-6
        else:
            assert MeanAudio.calls == calls0 and all(wire.decode_reply(rep[t]) == 0.0 for t in at)
    mb.drop_client("a0")
    t = mb.submit("a0", wire.encode_audio(audio[0, :1280]))
    assert wire.decode_reply(mb.flush()[t]) == 0.0
    with pytest.raises(NotImplementedError):
        mb.submit("m", bytes([wire.TAG_MEL]) + wire.encode_features(feats[0])[1:])


@pytest.mark.gpu
def test_microbatcher_on_hip_model(golden_frontend):
    import oracle
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel, HipSession
    from nanowakeword_amd.synth import synth_state_dict
    g = golden_frontend
    cfg = HeadConfig("dnn", (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    mb = wire.MicroBatcher(feature_session=HipSession(m, mode="features"), audio_backend=m, clip_samples=16000)
    feats = synth_features(7, (101, 64), seed=5)
    tk = [mb.submit(f"c{i}", wire.encode_features(feats[i:i + 1])) for i in range(7)]
    pcm = g["pcm"][:3]
    ta = [mb.submit(f"a{i}", wire.encode_audio(pcm[i])) for i in range(3)]
    rep = mb.flush()
    want = oracle.sigmoid(oracle.model_forward(feats, sd, cfg)).ravel()
    assert np.abs(np.array([wire.decode_reply(rep[t]) for t in tk]) - want).max() <= 1e-5
    lm = oracle.frontend_logmel(pcm, g["window"], g["fb64"]).transpose(0, 2, 1)
    wa = oracle.sigmoid(oracle.model_forward(np.ascontiguousarray(lm), sd, cfg)).ravel()
    assert np.abs(np.array([wire.decode_reply(rep[t]) for t in ta]) - wa).max() <= 1e-5
    m.close()
