#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (read-only import).

Runs only in the build container, where /root/reference exists.  Nothing here
travels to the GPU box except the emitted fixtures (data: inputs + expected
outputs).  The reference's absent third-party deps are stubbed:
  * torchaudio.transforms.MelSpectrogram / AmplitudeToDB -> attribute-carrying
    shims restating torchaudio's published defaults (window = torch.hann_window,
    fb = melscale_fbanks formula in torch float32 like torchaudio evaluates it);
    the mel module is then swapped for the reference's own ONNXSafeMelSpectrogram
    by the reference's own replace_mel_spectrogram(), which is the form its CPU
    interpreter executes (nanowakeword/_export/onnx.py:27-93).
  * torchinfo -> empty stub (only used by Model.summary()).
  * onnxruntime -> scripted fake session, used only to capture predict()
    state-machine traces of NanoInterpreter (nanointerpreter.py:606-833).

usage: python tools/make_goldens.py [--out tests/golden]
"""
import argparse
import json
import math
import os
import sys
import types
import wave

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


# ------------------------------------------------------------------ third-party shims
class _Spec(torch.nn.Module):
    def __init__(self, n_fft, win_length, hop_length):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.center, self.pad_mode, self.power = True, "reflect", 2.0
        self.register_buffer("window", torch.hann_window(win_length))   # periodic=True default


class _MelScale(torch.nn.Module):
    def __init__(self, n_mels, sample_rate, n_stft, f_min=0.0, f_max=None):
        super().__init__()
        f_max = float(sample_rate // 2) if f_max is None else f_max
        # torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"), torch float32 ops
        all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
        m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
        m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
        m_pts = torch.linspace(m_min, m_max, n_mels + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
        up = slopes[:, 2:] / f_diff[1:]
        self.register_buffer("fb", torch.max(torch.zeros(1), torch.min(down, up)))


class MelSpectrogramShim(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, n_mels=128,
                 f_min=0.0, f_max=None, center=True):
        super().__init__()
        win_length = win_length or n_fft
        hop_length = hop_length or win_length // 2
        self.spectrogram = _Spec(n_fft, win_length, hop_length)
        self.spectrogram.center = center
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1, f_min, f_max)

    def forward(self, x):
        raise RuntimeError("shim: must be replaced by ONNXSafeMelSpectrogram")


class AmplitudeToDBShim(torch.nn.Module):
    """torchaudio.transforms.AmplitudeToDB(stype='power', top_db=None): 10*log10(clamp(x,1e-10)) - 10*log10(max(1e-10,1))."""
    def forward(self, x):
        x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
        return x_db - 10.0 * math.log10(max(1e-10, 1.0))


def install_stubs(scripted_session_factory=None):
    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")
    tat.MelSpectrogram = MelSpectrogramShim
    tat.AmplitudeToDB = AmplitudeToDBShim
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: None
    sys.modules["torchinfo"] = ti
    if REF not in sys.path:
        sys.path.insert(0, REF)


def ref_frontend(n_mels, center):
    from nanowakeword._export.onnx import ONNXSafeMelSpectrogram
    shim = MelSpectrogramShim(16000, n_fft=400, win_length=400, hop_length=160, n_mels=n_mels, center=center)
    return ONNXSafeMelSpectrogram(shim).eval(), shim


def load_wavs():
    out = {}
    base = os.path.join(REF, "examples", "training_data")
    for sub in sorted(os.listdir(base)):
        d = os.path.join(base, sub)
        if not os.path.isdir(d):
            continue
        for fn in sorted(os.listdir(d)):
            if fn.endswith(".wav"):
                with wave.open(os.path.join(d, fn), "rb") as f:
                    assert f.getframerate() == 16000 and f.getsampwidth() == 2 and f.getnchannels() == 1
                    out[f"{sub}/{fn}"] = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    install_stubs()
    torch.set_num_threads(1)

    from nanowakeword.modules.model import Model                      # reseeds torch/np (SEED=10)
    from nanowakeword._export import onnx as ref_onnx
    from nanowakeword_amd.config import HeadConfig, param_spec
    from nanowakeword_amd.synth import synth_pcm, synth_features, synth_state_dict, state_dict_checksum

    # ---------------------------------------------------------------- frontend goldens
    wavs = load_wavs()
    wav_names = sorted(wavs)
    wav_pcm = np.stack([wavs[k][:16000] for k in wav_names])
    assert wav_pcm.shape == (len(wav_names), 16000), wav_pcm.shape
    pcm_set = {
        "noise": synth_pcm("noise", 4), "loud": synth_pcm("loud", 1), "zeros": synth_pcm("zeros", 1),
        "sine": synth_pcm("sine", 2), "chirp": synth_pcm("chirp", 1), "square": synth_pcm("square", 1),
        "speechlike": synth_pcm("speechlike", 2), "wav": wav_pcm,
    }
    names, clips = [], []
    for k, v in pcm_set.items():
        for i in range(v.shape[0]):
            names.append(f"{k}{i}")
            clips.append(v[i])
    pcm = np.stack(clips)                                            # [16, 16000] int16
    fe64, shim64 = ref_frontend(64, True)
    fe40, shim40 = ref_frontend(40, False)
    with torch.no_grad():
        x = torch.from_numpy(pcm.astype(np.float32) / 32768.0)       # nanointerpreter.py:750
        mel64 = fe64(x)                                              # [B,64,101]
        db64 = AmplitudeToDBShim()(mel64)
        mel40 = fe40(x)                                              # [B,40,98]
        db40 = AmplitudeToDBShim()(mel40)
    assert mel64.shape[1:] == (64, 101) and mel40.shape[1:] == (40, 98)
    fr = dict(pcm=pcm, names=np.array(names), wav_names=np.array(wav_names),
              window=shim64.spectrogram.window.numpy(), fb64=shim64.mel_scale.fb.numpy(),
              fb40=shim40.mel_scale.fb.numpy(),
              real_basis_row1=fe64.real_basis[1, 0].numpy(), imag_basis_row1=fe64.imag_basis[1, 0].numpy(),
              mel64=mel64.numpy(), db64=db64.numpy(), mel40=mel40.numpy(), db40=db40.numpy())
    # frame-law edge cases (bit-exact frame counts)
    edge = {}
    for n in (400, 401, 559, 560, 15999, 16000, 16001, 32000):
        xe = torch.from_numpy(synth_pcm("noise", 1, n, seed=77).astype(np.float32) / 32768.0)
        with torch.no_grad():
            edge[n] = (int(fe64(xe).shape[-1]), int(fe40(xe).shape[-1]))
    fr["edge_n"] = np.array(sorted(edge))
    fr["edge_frames_center"] = np.array([edge[n][0] for n in sorted(edge)])
    fr["edge_frames_nocenter"] = np.array([edge[n][1] for n in sorted(edge)])
    # one short ragged clip with full outputs (N=1000) to pin reflect padding on tiny inputs
    xs = synth_pcm("noise", 2, 1000, seed=78)
    with torch.no_grad():
        fr["short_pcm"] = xs
        fr["short_db64"] = AmplitudeToDBShim()(fe64(torch.from_numpy(xs.astype(np.float32) / 32768.0))).numpy()
    np.savez_compressed(os.path.join(args.out, "frontend.npz"), **fr)
    print("frontend.npz:", {k: getattr(v, "shape", None) for k, v in fr.items()})

    # ---------------------------------------------------------------- head goldens
    def ref_model(cfg: HeadConfig, sd_np):
        conf = {"activation_function": cfg.activation, "embedding_dim": cfg.embedding_dim,
                "crnn_cnn_channels": list(cfg.crnn_cnn_channels), "crnn_rnn_type": cfg.crnn_rnn_type,
                "conformer_d_model": cfg.conformer_d_model, "conformer_n_head": cfg.conformer_n_head}
        if cfg.model_type == "e2e_dnn":
            m = Model(conf, "g", input_shape=(16000,), model_type="e2e_dnn", mode="e2e")
        else:
            m = Model(conf, "g", input_shape=cfg.input_shape, model_type=cfg.model_type,
                      layer_dim=cfg.layer_dim, n_blocks=cfg.n_blocks)
        ref_sd = m.state_dict()
        ref_keys = {k: tuple(v.shape) for k, v in ref_sd.items()
                    if not k.endswith("num_batches_tracked") and not k.startswith("model.mel_spec")}
        spec = dict(param_spec(cfg))
        assert ref_keys == spec, (set(ref_keys) ^ set(spec),
                                  {k: (ref_keys.get(k), spec.get(k)) for k in set(ref_keys) & set(spec)
                                   if ref_keys[k] != spec[k]})
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
        assert not unexpected, unexpected
        assert all(k.endswith("num_batches_tracked") or k.startswith("model.mel_spec") for k in missing), missing
        return m.eval()

    head_cases = [
        ("dnn_16x96", HeadConfig("dnn", (16, 96))),
        ("dnn_98x40", HeadConfig("dnn", (98, 40))),
        ("dnn_101x64", HeadConfig("dnn", (101, 64))),
        ("dnn_16x96_b2_gelu", HeadConfig("dnn", (16, 96), layer_dim=64, n_blocks=2, activation="gelu")),
        ("cnn_16x96", HeadConfig("cnn", (16, 96))),
        ("cnn_98x40", HeadConfig("cnn", (98, 40))),
        ("cnn_101x64", HeadConfig("cnn", (101, 64))),
        ("cnn_16x96_silu", HeadConfig("cnn", (16, 96), activation="silu")),
        ("crnn_16x96", HeadConfig("crnn", (16, 96))),
        ("crnn_101x64", HeadConfig("crnn", (101, 64))),
        ("crnn_98x40", HeadConfig("crnn", (98, 40))),
        ("crnn_16x96_b2_silu", HeadConfig("crnn", (16, 96), layer_dim=32, n_blocks=2, activation="silu")),
        ("gru_16x96", HeadConfig("gru", (16, 96))),
        ("gru_101x64", HeadConfig("gru", (101, 64))),
        ("gru_16x96_b2", HeadConfig("gru", (16, 96), layer_dim=48, n_blocks=2)),
        ("bcresnet_16x96", HeadConfig("bcresnet", (16, 96))),
        ("bcresnet_98x40", HeadConfig("bcresnet", (98, 40))),
        ("bcresnet_101x64", HeadConfig("bcresnet", (101, 64))),
        ("bcresnet_16x96_gelu", HeadConfig("bcresnet", (16, 96), activation="gelu")),
        ("conformer_16x96", HeadConfig("conformer", (16, 96))),
        ("conformer_101x64", HeadConfig("conformer", (101, 64))),
        ("conformer_16x96_b2", HeadConfig("conformer", (16, 96), n_blocks=2, embedding_dim=32)),
        ("e2e_dnn_64x101", HeadConfig("e2e_dnn", (64, 101))),
    ]
    db64_np = db64.numpy()
    db40_np = db40.numpy()
    heads = {}
    meta = {}
    for name, cfg in head_cases:
        sd = synth_state_dict(cfg)
        m = ref_model(cfg, sd)
        T, F = cfg.input_shape
        out = {}
        # (i) synthetic N(0,1) features
        feats = synth_features(4, (T, F))
        body = m.model if cfg.model_type != "e2e_dnn" else None
        with torch.no_grad():
            if cfg.model_type == "e2e_dnn":
                # run the conv body + classifier on given log-mel: emulate forward after the dB stage
                def body_fwd(lm):
                    mm = m.model
                    h = mm.conv_block(lm.unsqueeze(1))
                    h = mm.flatten(h)
                    h = mm.act1(mm.bn1(mm.fc1(h)))
                    return m.classifier(mm.out(h))
                out["logits_feat"] = body_fwd(torch.from_numpy(feats)).numpy()
            else:
                out["logits_feat"] = m(torch.from_numpy(feats)).numpy()
                out["emb_feat"] = body(torch.from_numpy(feats)).numpy()
            # (ii) composite: reference frontend log-mel -> head
            if (T, F) == (101, 64):
                lm = np.ascontiguousarray(db64_np.transpose(0, 2, 1))
                out["logits_pcm"] = m(torch.from_numpy(lm)).numpy()
            elif (T, F) == (98, 40):
                lm = np.ascontiguousarray(db40_np.transpose(0, 2, 1))
                out["logits_pcm"] = m(torch.from_numpy(lm)).numpy()
            elif cfg.model_type == "e2e_dnn":
                # the native composite: the reference's own E2E model, export-patched, fed float PCM
                ref_onnx.replace_mel_spectrogram(m)
                xin = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
                logits_torchpool = m(xin).numpy()

                class W(torch.nn.Module):
                    def __init__(s, mm):
                        super().__init__(); s.trained_model = mm
                    def forward(s, x):
                        return torch.sigmoid(s.trained_model(x)).view(-1, 1, 1)
                w = W(m).eval()
                ref_onnx.make_onnx_safe_adaptive_pool(w, xin[:1].unsqueeze(1))
                out["probs_pcm_export"] = w(xin.unsqueeze(1)).numpy()        # (B,1,1) as the ONNX output
                out["logits_pcm"] = m(xin).numpy()                            # after pool patch
                out["logits_pcm_adaptivepool"] = logits_torchpool
        out["sd_checksum"] = np.array(state_dict_checksum(sd))
        meta[name] = cfg.to_dict()
        for k, v in out.items():
            heads[f"{name}/{k}"] = v
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()},
              "logit range", float(out["logits_feat"].min()), float(out["logits_feat"].max()))
    heads["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(args.out, "heads.npz"), **heads)

    # a small full state_dict, to detect RNG drift of synth_state_dict independently of checksums
    cfg_small = HeadConfig("dnn", (16, 96), layer_dim=16, embedding_dim=8)
    np.savez_compressed(os.path.join(args.out, "sd_dnn_small.npz"), **synth_state_dict(cfg_small))

    # ---------------------------------------------------------------- predict() state-machine traces
    make_predict_traces(os.path.join(args.out, "predict_trace.json"))
    make_wire_fixtures(os.path.join(args.out, "wire_messages.json"))
    make_audio_features_traces(os.path.join(args.out, "audio_features_trace.json"))
    make_onnx_fixtures(os.path.join(args.out, "onnx"))
    make_round2_goldens(args.out)
    make_round4_goldens(args.out)
    make_round6_goldens(args.out)
    print("done ->", args.out)


def make_predict_traces(path):
    """Capture NanoInterpreter e2e-mode predict()/predict_clip() behaviour with a scripted session."""
    ort = types.ModuleType("onnxruntime")

    class SessionOptions:
        inter_op_num_threads = 0
        intra_op_num_threads = 0

    class _Inp:
        def __init__(self, name, shape):
            self.name, self.shape = name, shape

    class InferenceSession:
        def __init__(self, path, sess_options=None, providers=None):
            self._model_filename = path
            self.clip_samples = 16000
            self.calls = []

        def get_inputs(self):
            return [_Inp("input", [None, 1, self.clip_samples])]

        def run(self, output_names, feed):
            x = feed["input"]
            assert x.shape == (1, 1, self.clip_samples) and x.dtype == np.float32
            self.calls.append(x.copy())
            s = float(np.clip(np.abs(x).mean() * 4.0, 0.0, 1.0))
            return [np.array([[[s]]], dtype=np.float32)]

    ort.SessionOptions = SessionOptions
    ort.InferenceSession = InferenceSession
    sys.modules["onnxruntime"] = ort
    import tempfile
    from nanowakeword.interpreter.nanointerpreter import NanoInterpreter
    from nanowakeword_amd.synth import synth_pcm
    tmp = tempfile.mkdtemp()
    mp = os.path.join(tmp, "wake.onnx")
    open(mp, "wb").close()
    traces = {}
    stream = np.concatenate([synth_pcm("noise", 1, 16000 * 3, seed=5)[0],
                             synth_pcm("loud", 1, 16000, seed=6)[0]])

    def run_trace(chunk, **kw):
        it = NanoInterpreter.load_model(mp)
        assert it.preprocessor is None
        rows = []
        for i in range(0, len(stream) - chunk + 1, chunk):
            r = it.predict(stream[i:i + chunk], **kw)
            rows.append([float(it.raw_scores["wake"]), float(r.score), bool(r.detected) if kw.get("threshold") else False])
        sess = it.models["wake"]
        first_clip_sum = float(np.abs(sess.calls[0]).sum()) if sess.calls else 0.0
        return {"chunk": chunk, "kw": {k: v for k, v in kw.items()}, "rows": rows, "n_calls": len(sess.calls),
                "first_clip_abs_sum": first_clip_sum}

    traces["chunk1280"] = run_trace(1280)
    traces["chunk4000"] = run_trace(4000)
    traces["chunk16000"] = run_trace(16000)
    traces["patience3"] = run_trace(1280, patience={"wake": 3}, threshold={"wake": 0.5})
    traces["debounce"] = run_trace(1280, debounce_time=0.5, threshold={"wake": 0.5})
    it = NanoInterpreter.load_model(mp)
    res = it.predict_clip(stream[:20000])
    traces["predict_clip_len"] = len(res)
    traces["predict_clip_score"] = float(res[0].score)
    traces["predict_clip_raw"] = float(it.raw_scores["wake"])
    it.reset()
    traces["after_reset_buffer"] = int(it.e2e_buffer_samples["wake"])
    traces["stream_spec"] = {"parts": [["noise", 48000, 5], ["loud", 16000, 6]]}
    with open(path, "w") as f:
        json.dump(traces, f, indent=1)
    print("predict_trace.json rows:", {k: (len(v["rows"]) if isinstance(v, dict) and "rows" in v else v)
                                       for k, v in traces.items()})



def make_round2_goldens(out_dir):
    """Round-2 additions (separate file so the round-1 fixtures stay byte-identical): CRNN with the reference's DEFAULT
    recurrent backend (nn.LSTM, model.py:214), CRNN conv stacks other than [16, 32, 32], recurrent widths that take the
    generic kernels, and the native E2E composite at clip lengths where the export-form average pool
    (make_onnx_safe_adaptive_pool, _export/onnx.py:96-154) is NOT the adaptive pool."""
    install_stubs()
    torch.set_num_threads(1)
    from nanowakeword.modules.model import Model
    from nanowakeword._export import onnx as ref_onnx
    from nanowakeword_amd.config import HeadConfig, param_spec
    from nanowakeword_amd.synth import synth_pcm, synth_features, synth_state_dict, state_dict_checksum
    fr = dict(np.load(os.path.join(out_dir, "frontend.npz"), allow_pickle=False))
    db64 = fr["db64"]

    def ref_model(cfg, sd_np, n_samples=16000):
        conf = {"activation_function": cfg.activation, "embedding_dim": cfg.embedding_dim,
                "crnn_cnn_channels": list(cfg.crnn_cnn_channels), "crnn_rnn_type": cfg.crnn_rnn_type,
                "conformer_d_model": cfg.conformer_d_model, "conformer_n_head": cfg.conformer_n_head}
        if cfg.model_type == "e2e_dnn":
            m = Model(conf, "g", input_shape=(n_samples,), model_type="e2e_dnn", mode="e2e")
        else:
            m = Model(conf, "g", input_shape=cfg.input_shape, model_type=cfg.model_type, layer_dim=cfg.layer_dim, n_blocks=cfg.n_blocks)
        ref_keys = {k: tuple(v.shape) for k, v in m.state_dict().items()
                    if not k.endswith("num_batches_tracked") and not k.startswith("model.mel_spec")}
        assert ref_keys == dict(param_spec(cfg)), set(ref_keys) ^ set(param_spec(cfg))
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
        assert not unexpected, unexpected
        return m.eval()

    cases = [
        ("crnn_lstm_16x96", HeadConfig("crnn", (16, 96), crnn_rnn_type="lstm")),                      # the reference's default CRNN
        ("crnn_lstm_101x64", HeadConfig("crnn", (101, 64), crnn_rnn_type="lstm")),
        ("crnn_lstm_16x96_b2_h64_silu", HeadConfig("crnn", (16, 96), layer_dim=64, n_blocks=2, activation="silu", crnn_rnn_type="lstm")),
        ("crnn_lstm_98x40_h48", HeadConfig("crnn", (98, 40), layer_dim=48, crnn_rnn_type="lstm")),    # generic LSTM kernel
        ("crnn_gru_101x64_ch8_16", HeadConfig("crnn", (101, 64), crnn_cnn_channels=[8, 16])),
        ("crnn_gru_16x96_ch16_32_64_64", HeadConfig("crnn", (16, 96), crnn_cnn_channels=[16, 32, 64, 64])),
        ("crnn_gru_64x64_h48", HeadConfig("crnn", (64, 64), layer_dim=48)),                           # generic GRU kernel
    ]
    heads, meta = {}, {}
    for name, cfg in cases:
        sd = synth_state_dict(cfg)
        m = ref_model(cfg, sd)
        feats = synth_features(4, cfg.input_shape)
        with torch.no_grad():
            out = {"logits_feat": m(torch.from_numpy(feats)).numpy(), "emb_feat": m.model(torch.from_numpy(feats)).numpy()}
            if cfg.input_shape == (101, 64):
                out["logits_pcm"] = m(torch.from_numpy(np.ascontiguousarray(db64.transpose(0, 2, 1)))).numpy()
        out["sd_checksum"] = np.array(state_dict_checksum(sd))
        meta[name] = cfg.to_dict()
        for k, v in out.items():
            heads[f"{name}/{k}"] = v
        print("r02", name, {k: getattr(v, "shape", v) for k, v in out.items()})
    # ---- E2E composite at 1.5 s and 2 s clips: frames 151 / 201 -> conv3 output (64,16,37) / (64,16,50): the export
    # pool windows are (16,10)/(16,9) and (16,14)/(16,12) while AdaptiveAvgPool2d((1,4)) uses ragged windows
    for n_samples in (24000, 32000):
        frames = 1 + n_samples // 160
        name = f"e2e_dnn_64x{frames}"
        cfg = HeadConfig("e2e_dnn", (64, frames))
        sd = synth_state_dict(cfg)
        m = ref_model(cfg, sd, n_samples)
        pcm = np.concatenate([synth_pcm("noise", 2, n_samples, seed=41), synth_pcm("speechlike", 2, n_samples, seed=42),
                              synth_pcm("loud", 1, n_samples, seed=43), synth_pcm("zeros", 1, n_samples)])
        ref_onnx.replace_mel_spectrogram(m)
        xin = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
        with torch.no_grad():
            logits_adaptive = m(xin).numpy()

            class W(torch.nn.Module):
                def __init__(s, mm):
                    super().__init__(); s.trained_model = mm
                def forward(s, x):
                    return torch.sigmoid(s.trained_model(x)).view(-1, 1, 1)
            w = W(m).eval()
            ref_onnx.make_onnx_safe_adaptive_pool(w, xin[:1].unsqueeze(1))
            probs_export = w(xin.unsqueeze(1)).numpy()
            logits_export = m(xin).numpy()
        heads[f"{name}/pcm"] = pcm
        heads[f"{name}/logits_pcm"] = logits_export
        heads[f"{name}/probs_pcm_export"] = probs_export
        heads[f"{name}/logits_pcm_adaptivepool"] = logits_adaptive
        heads[f"{name}/sd_checksum"] = np.array(state_dict_checksum(sd))
        meta[name] = cfg.to_dict()
        print("r02", name, "export-vs-adaptive max |dlogit|", float(np.abs(logits_export - logits_adaptive).max()))
    heads["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out_dir, "heads_r02.npz"), **heads)


def make_round4_goldens(out_dir):
    """Round-4 additions (own file: earlier fixtures stay byte-identical):
      * the reference's distilled "lite" gate - the student `_build_student` builds (nanowakeword/train/distill.py:45-76: DNN,
        layer_size 8, n_blocks 1, embedding_dim 8 -> classifier 8 -> 4 -> 1), the model `load_model(cascade=True)` looks for
        (nanointerpreter.py:310-325); built by calling the reference function itself with a DNN teacher;
      * a DNN whose flattened input is >= 2048 and NOT a multiple of 4 (the split-K / VALU-fallback seam);
      * recurrent widths outside the register-resident kernels (> 256, not a multiple of 4) and Conformer head dims outside the
        compiled set: shapes nn.GRU / nn.LSTM / nn.MultiheadAttention accept (architectures.py:132-145,238-254,499)."""
    install_stubs()
    torch.set_num_threads(1)
    from nanowakeword.modules.model import Model
    from nanowakeword_amd.config import HeadConfig, param_spec
    from nanowakeword_amd.synth import synth_features, synth_state_dict, state_dict_checksum
    fr = dict(np.load(os.path.join(out_dir, "frontend.npz"), allow_pickle=False))
    db64, db40 = fr["db64"], fr["db40"]
    for name in ("tqdm",):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod = types.ModuleType(name); mod.tqdm = lambda it, *a, **k: it; sys.modules[name] = mod
    from nanowakeword.train import distill as ref_distill

    def load(m, cfg, sd_np):
        ref_keys = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
        assert ref_keys == dict(param_spec(cfg)), set(ref_keys) ^ set(param_spec(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        return m.eval()

    def ref_model(cfg, sd_np):
        conf = {"activation_function": cfg.activation, "embedding_dim": cfg.embedding_dim,
                "crnn_cnn_channels": list(cfg.crnn_cnn_channels), "crnn_rnn_type": cfg.crnn_rnn_type,
                "conformer_d_model": cfg.conformer_d_model, "conformer_n_head": cfg.conformer_n_head}
        m = Model(conf, "g", input_shape=cfg.input_shape, model_type=cfg.model_type, layer_dim=cfg.layer_dim, n_blocks=cfg.n_blocks)
        return load(m, cfg, sd_np)

    def lite_model(cfg, sd_np):
        teacher = Model({"activation_function": "relu", "embedding_dim": 64}, "g", input_shape=cfg.input_shape, model_type="dnn")
        student = ref_distill._build_student(teacher, cfg.input_shape, {})
        assert student.model_name == "g_lite"
        return load(student, cfg, sd_np)

    lite = dict(layer_dim=8, n_blocks=1, embedding_dim=8)
    cases = [
        ("lite_dnn_16x96", HeadConfig("dnn", (16, 96), **lite), lite_model),
        ("lite_dnn_101x64", HeadConfig("dnn", (101, 64), **lite), lite_model),
        ("lite_dnn_98x40", HeadConfig("dnn", (98, 40), **lite), lite_model),
        ("dnn_101x41", HeadConfig("dnn", (101, 41)), ref_model),                       # K = 4141: >= 2048 and K % 4 == 1
        ("dnn_33x63_l20", HeadConfig("dnn", (33, 63), layer_dim=20, embedding_dim=10), ref_model),   # K = 2079, N = 20: nothing a multiple of 4
        ("gru_16x96_h130", HeadConfig("gru", (16, 96), layer_dim=130), ref_model),     # not a multiple of 4
        ("gru_16x96_h320_b2", HeadConfig("gru", (16, 96), layer_dim=320, n_blocks=2), ref_model),   # > 256
        ("crnn_lstm_16x96_h512", HeadConfig("crnn", (16, 96), layer_dim=512, crnn_rnn_type="lstm"), ref_model),
        ("crnn_gru_101x64_h300", HeadConfig("crnn", (101, 64), layer_dim=300), ref_model),
        ("crnn_lstm_98x40_h21", HeadConfig("crnn", (98, 40), layer_dim=21, crnn_rnn_type="lstm"), ref_model),
        ("conformer_16x96_d100_h4", HeadConfig("conformer", (16, 96), conformer_d_model=100, conformer_n_head=4), ref_model),   # head dim 25
        ("conformer_101x64_d160_h2", HeadConfig("conformer", (101, 64), conformer_d_model=160, conformer_n_head=2), ref_model),  # head dim 80
        ("conformer_16x96_d66_h6", HeadConfig("conformer", (16, 96), conformer_d_model=66, conformer_n_head=6), ref_model),      # head dim 11
    ]
    heads, meta = {}, {}
    for name, cfg, build in cases:
        sd = synth_state_dict(cfg)
        m = build(cfg, sd)
        feats = synth_features(4, cfg.input_shape)
        with torch.no_grad():
            out = {"logits_feat": m(torch.from_numpy(feats)).numpy(), "emb_feat": m.model(torch.from_numpy(feats)).numpy()}
            if cfg.input_shape == (101, 64):
                out["logits_pcm"] = m(torch.from_numpy(np.ascontiguousarray(db64.transpose(0, 2, 1)))).numpy()
            if cfg.input_shape == (98, 40):
                out["logits_pcm"] = m(torch.from_numpy(np.ascontiguousarray(db40.transpose(0, 2, 1)))).numpy()
        out["sd_checksum"] = np.array(state_dict_checksum(sd))
        meta[name] = cfg.to_dict()
        for k, v in out.items():
            heads[f"{name}/{k}"] = v
        print("r04", name, {k: getattr(v, "shape", v) for k, v in out.items()})
    heads["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out_dir, "heads_r04.npz"), **heads)


def make_round6_goldens(out_dir):
    """Round-6 additions (own file: earlier fixtures stay byte-identical): Conformer shapes the round-6 kernels take - the textbook
    d_model 256 / 4 heads (head dim 64) and 192 / 4 (head dim 48) on the fused feed-forward / short-K / two-term attention kernels, and the
    default width at clip lengths other than 101 frames through the one-launch attention module (attn_x3.hip: 5 .. 8 blocks of 16 keys,
    two blocks) - logits and embeddings from the reference's own Model (architectures.py:441-543)."""
    install_stubs()
    torch.set_num_threads(1)
    from nanowakeword.modules.model import Model
    from nanowakeword_amd.config import HeadConfig, param_spec
    from nanowakeword_amd.synth import synth_features, synth_state_dict, state_dict_checksum
    fr = dict(np.load(os.path.join(out_dir, "frontend.npz"), allow_pickle=False))
    db64 = fr["db64"]
    cases = [
        ("conformer_101x64_d256_h4", HeadConfig("conformer", (101, 64), conformer_d_model=256, conformer_n_head=4)),
        ("conformer_101x64_d192_h4", HeadConfig("conformer", (101, 64), conformer_d_model=192, conformer_n_head=4)),
        ("conformer_40x64_d256_h8", HeadConfig("conformer", (40, 64), conformer_d_model=256, conformer_n_head=8, embedding_dim=32)),
        ("conformer_70x64_d144_h4", HeadConfig("conformer", (70, 64), embedding_dim=32)),
        ("conformer_128x40_d144_h4_b2", HeadConfig("conformer", (128, 40), n_blocks=2)),
    ]
    heads, meta = {}, {}
    for name, cfg in cases:
        sd = synth_state_dict(cfg)
        conf = {"activation_function": cfg.activation, "embedding_dim": cfg.embedding_dim,
                "crnn_cnn_channels": list(cfg.crnn_cnn_channels), "crnn_rnn_type": cfg.crnn_rnn_type,
                "conformer_d_model": cfg.conformer_d_model, "conformer_n_head": cfg.conformer_n_head}
        m = Model(conf, "g", input_shape=cfg.input_shape, model_type=cfg.model_type, layer_dim=cfg.layer_dim, n_blocks=cfg.n_blocks)
        ref_keys = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
        assert ref_keys == dict(param_spec(cfg)), set(ref_keys) ^ set(param_spec(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.eval()
        feats = synth_features(4, cfg.input_shape)
        with torch.no_grad():
            out = {"logits_feat": m(torch.from_numpy(feats)).numpy(), "emb_feat": m.model(torch.from_numpy(feats)).numpy()}
            if cfg.input_shape == (101, 64):
                out["logits_pcm"] = m(torch.from_numpy(np.ascontiguousarray(db64.transpose(0, 2, 1)))).numpy()
        out["sd_checksum"] = np.array(state_dict_checksum(sd))
        meta[name] = cfg.to_dict()
        for k, v in out.items():
            heads[f"{name}/{k}"] = v
        print("r06", name, {k: getattr(v, "shape", v) for k, v in out.items()})
    heads["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out_dir, "heads_r06.npz"), **heads)


def make_wire_fixtures(path):
    """Messages produced by the reference's own encoders (remote_verifier.py:147-158) for tests/test_wire.py."""
    from nanowakeword.interpreter import remote_verifier as rv
    from nanowakeword_amd.synth import synth_features, synth_pcm
    f = synth_features(2, (16, 96), seed=4)
    a = synth_pcm("noise", 1, 1280, seed=4)[0]
    out = {"features_shape": [2, 16, 96], "features_seed": 4, "features_hex": rv.encode_features(f).hex(),
           "audio_n": 1280, "audio_seed": 4, "audio_hex": rv.encode_audio(a).hex(),
           "tags": {"features": rv._TAG_FEATURES, "mel": rv._TAG_MEL, "audio": rv._TAG_AUDIO},
           "reply": json.dumps({"score": 0.75})}
    json.dump(out, open(path, "w"))


def make_audio_features_traces(path):
    """Drive the reference AudioFeatures (nanowakeword/data/AudioFeatures.py) with deterministic fake ORT sessions
    and record its streaming / batch behaviour for tests/test_audio_features.py."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from fake_models import fake_embed, fake_mel
    ort = types.ModuleType("onnxruntime")

    class SessionOptions:
        inter_op_num_threads = 0
        intra_op_num_threads = 0

    class InferenceSession:
        def __init__(self, path, sess_options=None, providers=None):
            self.kind = "mel" if "mel" in os.path.basename(str(path)) else "emb"
            self.providers = providers or ["CPUExecutionProvider"]

        def get_providers(self):
            return self.providers

        def run(self, names, feed):
            if self.kind == "mel":
                return [fake_mel(feed["input"])]
            return [fake_embed(feed["input_1"])]

    ort.SessionOptions, ort.InferenceSession = SessionOptions, InferenceSession
    sys.modules["onnxruntime"] = ort
    import nanowakeword.interpreter.models as ref_models_pkg
    import nanowakeword.data.AudioFeatures as AF

    class _Models:
        melspectrogram_onnx = "melspectrogram.onnx"
        embedding_model_onnx = "embedding_model.onnx"
    AF.models = _Models()
    from nanowakeword_amd.synth import synth_pcm
    np.random.seed(1234)
    af = AF.AudioFeatures()
    stream = synth_pcm("noise", 1, 16000 * 3, seed=21)[0]
    sizes = [1280, 400, 880, 1280, 2560, 300, 1280, 3000, 1280, 1280, 640, 640, 5000, 1280]
    rows, pos = [], 0
    for n in sizes:
        r = af(stream[pos:pos + n]); pos += n
        rows.append({"n": n, "ret": int(r), "feat_shape": list(af.feature_buffer.shape),
                     "mel_shape": list(af.melspectrogram_buffer.shape),
                     "feat_last_sum": float(np.asarray(af.feature_buffer[-1], np.float64).sum()),
                     "feat_sum": float(np.asarray(af.feature_buffer, np.float64).sum()),
                     "acc": int(af.accumulated_samples), "rem": int(af.raw_data_remainder.shape[0])})
    gf = af.get_features(16)
    clips = synth_pcm("noise", 3, 32000, seed=22)
    emb = af.embed_clips(clips, batch_size=2, ncpu=1)
    np.random.seed(99)
    shape2 = list(af.get_embedding_shape(2.0))
    out = {"seed": 1234, "sizes": sizes, "rows": rows, "get_features_shape": list(gf.shape),
           "get_features_sum": float(gf.astype(np.float64).sum()),
           "embed_clips_shape": list(emb.shape), "embed_clips_sum": float(emb.astype(np.float64).sum()),
           "embed_clips_row": [float(v) for v in emb[1, 3, :8]], "embedding_shape_2s": shape2}
    json.dump(out, open(path, "w"), indent=1)
    print("audio_features_trace.json:", out["embed_clips_shape"], shape2, rows[-1])


def make_onnx_fixtures(outdir):
    """Real reference exports: every in-scope head goes through the reference's own ``export_onnx_model``
    (nanowakeword/_export/onnx.py:157-229) with small shapes, and the exported wrapper's probabilities on fixed
    inputs are stored next to the files.  The reference pins torch 2.8, whose ``torch.onnx.export`` is the
    TorchScript exporter; torch 2.10 here needs ``dynamo=False`` for the same exporter, and its last step (attaching
    onnxscript functions - there are none) imports the absent ``onnx`` package, so that step is bypassed.  The bytes
    written are those of torch's C++ ONNX serializer."""
    os.makedirs(outdir, exist_ok=True)
    install_stubs()
    torch.set_num_threads(1)
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    orig_export = torch.onnx.export

    def export_torchscript(*a, **k):
        k.setdefault("dynamo", False)
        return orig_export(*a, **k)
    torch.onnx.export = export_torchscript

    from nanowakeword.modules.model import Model
    from nanowakeword._export import onnx as ref_onnx
    from nanowakeword_amd.config import HeadConfig
    from nanowakeword_amd.synth import synth_pcm, synth_features, synth_state_dict

    cases = [
        ("dnn", HeadConfig("dnn", (4, 8), layer_dim=16, n_blocks=2, embedding_dim=8, activation="gelu")),
        ("cnn", HeadConfig("cnn", (8, 16), embedding_dim=16)),
        ("crnn", HeadConfig("crnn", (8, 16), layer_dim=16, n_blocks=2, embedding_dim=16, activation="silu")),
        ("gru", HeadConfig("gru", (6, 12), layer_dim=16, n_blocks=2, embedding_dim=8)),
        ("bcresnet", HeadConfig("bcresnet", (16, 24), embedding_dim=16)),
        ("conformer", HeadConfig("conformer", (8, 12), n_blocks=2, embedding_dim=16, conformer_d_model=32, conformer_n_head=8)),
        ("e2e_dnn", HeadConfig("e2e_dnn", (64, 101))),
        ("crnn_lstm", HeadConfig("crnn", (8, 16), layer_dim=16, n_blocks=2, embedding_dim=16, crnn_rnn_type="lstm")),   # reference default backend
    ]
    meta = {}
    arrays = {}
    for name, cfg in cases:
        sd = synth_state_dict(cfg)
        conf = {"activation_function": cfg.activation, "embedding_dim": cfg.embedding_dim,
                "crnn_cnn_channels": list(cfg.crnn_cnn_channels), "crnn_rnn_type": cfg.crnn_rnn_type,
                "conformer_d_model": cfg.conformer_d_model, "conformer_n_head": cfg.conformer_n_head}
        if cfg.model_type == "e2e_dnn":
            m = Model(conf, "g", input_shape=(16000,), model_type="e2e_dnn", mode="e2e")
        else:
            m = Model(conf, "g", input_shape=cfg.input_shape, model_type=cfg.model_type,
                      layer_dim=cfg.layer_dim, n_blocks=cfg.n_blocks)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        m.eval()
        shape = (16000,) if cfg.model_type == "e2e_dnn" else cfg.input_shape
        ref_onnx.export_onnx_model(m, shape, {}, name, outdir)         # patches m in place (mel, pooling)
        path = os.path.join(outdir, name + ".onnx")
        assert os.path.exists(path), f"export of {name} failed"
        if cfg.model_type == "e2e_dnn":
            pcm = np.concatenate([synth_pcm("noise", 2, 16000, seed=31), synth_pcm("speechlike", 2, 16000, seed=32)])
            x = torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(1)
            arrays[name + "/pcm"] = pcm
        else:
            feats = synth_features(4, cfg.input_shape)
            x = torch.from_numpy(feats)
        with torch.no_grad():
            logits = m(x).numpy()
        arrays[name + "/logits"] = logits.reshape(-1).astype(np.float32)
        arrays[name + "/probs"] = (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).reshape(-1).astype(np.float32)
        meta[name] = cfg.to_dict()
        print("onnx fixture", name, os.path.getsize(path), "bytes; logits", logits.reshape(-1))
    arrays["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(outdir, "expected.npz"), **arrays)
    torch.onnx.export = orig_export


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--onnx-only":
        make_onnx_fixtures(os.path.join(REPO, "tests", "golden", "onnx"))
    elif len(sys.argv) > 1 and sys.argv[1] == "--r04-only":
        make_round4_goldens(os.path.join(REPO, "tests", "golden"))
    elif len(sys.argv) > 1 and sys.argv[1] == "--r06-only":
        make_round6_goldens(os.path.join(REPO, "tests", "golden"))
    elif len(sys.argv) > 1 and sys.argv[1] == "--r02-only":
        make_round2_goldens(os.path.join(REPO, "tests", "golden"))
    else:
        main()
