"""(round 6) Formerly tests/test_gpu_frontend3.py: parity checks of the matrix-pipe frontend experiment.  The kernel no longer ships in
libnwwhip.so (it is slower than the FFT kernel: DESIGN 4.1); these checks ran green against the round-5 library with NWW_FE3=1 and are
kept for the record - they need a library build that links tools/ubench/fe3/frontend3.hip and honours NWW_FE3.
"""
"""The matrix-pipe frontend (frontend3.hip; opt-in NWW_FE3 = 1): the 400-point DFT as a prime-factor 25 x 16 pair of dense products
on v_mfma_f32_16x16x32_f16 in two binary16 terms.  Same parity contract as the FFT kernel: frame law bit-exact, criteria A / B / C of
tests/parity.py against the reference goldens, the oracle and float64; results independent of batch, frame grouping and ring placement."""
import os

import numpy as np
import pytest

import oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
from parity import assert_frontend_amplitude, assert_frontend_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fe3_env():
    old = os.environ.get("NWW_FE3")
    os.environ["NWW_FE3"] = "1"            # read at every nww_finalize
    yield
    if old is None:
        os.environ.pop("NWW_FE3", None)
    else:
        os.environ["NWW_FE3"] = old


def _model(g, head, shape, n_mels, center):
    from nanowakeword_amd.session import HipModel
    cfg = HeadConfig(head, shape)
    return HipModel(cfg, FrontendConfig(n_mels=n_mels, center=center), state_dict=synth_state_dict(cfg), window=g["window"],
                    mel_fb=g["fb64"] if n_mels == 64 else g["fb40"])


@pytest.mark.parametrize("variant", ["64c", "40n"])
def test_matrix_pipe_frontend_vs_reference_golden(fe3_env, golden_frontend, variant):
    g = golden_frontend
    n_mels, center = (64, True) if variant == "64c" else (40, False)
    m = _model(g, "dnn", (101, 64) if center else (98, 40), n_mels, center)
    assert "frontend3" in m.describe_plan()
    fb = g["fb64"] if center else g["fb40"]
    db, mel = m.frontend(g["pcm"], return_power=True)
    mel_ref, db_ref = (g["mel64"], g["db64"]) if center else (g["mel40"], g["db40"])
    assert db.shape == db_ref.shape                      # frame count bit-exact
    e_db, e_mel, frac = assert_frontend_close(mel, db, mel_ref, db_ref, variant)
    mo = oracle.mel_power(g["pcm"], g["window"], fb, center=center)
    assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), variant + "/oracle")
    exact = oracle.mel_power(g["pcm"], g["window"], fb, center=center, dtype=np.float64)
    k_all, k_exc = assert_frontend_amplitude(mel, exact, variant)
    print(f"frontend3 {variant}: max dB err {e_db:.2e} ({frac:.0%} of bins), mel err {e_mel:.2e} x frame peak, amplitude {k_all:.2f} / {k_exc:.2f} x 2^-24")
    # the fast (frames-major, dB only) path returns the same bits as the general one
    assert np.array_equal(m.frontend(g["pcm"]), db)
    m.close()


def test_matrix_pipe_frontend_edges_and_invariance(fe3_env, golden_frontend):
    g = golden_frontend
    mc = _model(g, "dnn", (101, 64), 64, True)
    mn = _model(g, "dnn", (98, 40), 40, False)
    for n, fc, fn in zip(g["edge_n"], g["edge_frames_center"], g["edge_frames_nocenter"]):
        x = synth_pcm("noise", 2, int(n), seed=77)
        for m, frames, nm, center, fb in ((mc, fc, 64, True, g["fb64"]), (mn, fn, 40, False, g["fb40"])):
            db, mel = m.frontend(x, return_power=True)
            assert db.shape == (2, nm, int(frames))
            mo = oracle.mel_power(x, g["window"], fb, center=center)
            assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), f"N={n}")
    assert np.abs(mc.frontend(g["short_pcm"]) - g["short_db64"]).max() <= 1e-4
    # digital silence sits exactly on the clamp floor; full-scale square wave and int16 extremes stay finite and close
    assert np.all(mc.frontend(np.zeros((2, 16000), np.int16)) == -100.0)
    for kind in ("loud", "square"):
        x = synth_pcm(kind, 2, 16000)
        db, mel = mc.frontend(x, return_power=True)
        mo = oracle.mel_power(x, g["window"], g["fb64"])
        assert_frontend_close(mel, db, mo, oracle.logmel_db(mo), kind)
    # a clip's log-mel does not depend on the batch around it, nor on its position
    x = synth_pcm("speechlike", 70, 16000, seed=5)
    full = mc.frontend(x)
    assert np.array_equal(mc.frontend(x[13:14]), full[13:14])
    assert np.array_equal(mc.frontend(x[::-1].copy())[::-1], full)
    # odd-aligned rows (2-byte staging path): same values as the aligned path
    odd = np.zeros((3, 16001), np.int16)
    odd[:, :16000] = x[:3]
    assert np.array_equal(mc.frontend(odd)[:, :, :99], full[:3, :, :99])        # frames that do not touch the (different) clip end
    mc.close(); mn.close()


def test_matrix_pipe_frontend_logits_and_streaming(fe3_env, golden_frontend):
    """PCM -> logit through frontend3 + CNN head within 1e-4 of the oracle on the broadband / speech / silent golden clips, and the
    streaming hops (frame subsets into the log-mel ring) bit-identical to re-scoring the window."""
    from nanowakeword_amd.session import HipModel
    g = golden_frontend
    cfg = HeadConfig("cnn", (101, 64))
    sd = synth_state_dict(cfg)
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, window=g["window"], mel_fb=g["fb64"])
    lg, _ = m.forward_pcm(g["pcm"])
    lm = oracle.frontend_logmel(g["pcm"], g["window"], g["fb64"]).transpose(0, 2, 1)
    ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
    broad = np.array([not str(n).startswith(("sine", "chirp")) for n in g["names"]])
    assert np.abs(lg - ref)[broad].max() <= 1e-4, np.abs(lg - ref)
    cfg2 = HeadConfig("crnn", (101, 64))
    m2 = HipModel(cfg2, FrontendConfig(), state_dict=synth_state_dict(cfg2), window=g["window"], mel_fb=g["fb64"])
    S, W, hop = 4, 16000, 1280
    streams = np.stack([synth_pcm("speechlike" if s % 2 else "noise", 1, hop * 30, seed=40 + s)[0] for s in range(S)])
    m2.stream_open(S, W, hop)
    hist = np.zeros((S, 0), np.int16)
    for i in range(30):
        chunk = np.ascontiguousarray(streams[:, i * hop:(i + 1) * hop])
        hist = np.concatenate([hist, chunk], axis=1)[:, -W:]
        lgs, _ = m2.stream_push(chunk)
        if hist.shape[1] < W:
            continue
        want, _ = m2.forward_pcm(np.ascontiguousarray(hist))
        assert np.array_equal(lgs, want), (i, np.abs(lgs - want).max())
    m2.stream_close(); m2.close(); m.close()
