"""How close is each conv arithmetic to exact arithmetic?  CNN head (101,64), synthetic weights, log-mel-like input.
Reference: the same network evaluated in float64 (numpy).  Prints max |error| of the trunk output (via the embedding
and the logit) for f32 MFMA, bf16x9 and bf16x6, plus their mutual differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict


def conv3x3_f64(x, w, b):                     # x [B,Cin,H,W], w [Cout,Cin,3,3]
    B, Cin, H, W = x.shape
    xp = np.zeros((B, Cin, H + 2, W + 2)); xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, w.shape[0], H, W))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bchw,oc->bohw", xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx])
    return out + b[None, :, None, None]


def pool2(x):
    B, C, H, W = x.shape
    x = x[:, :, :H // 2 * 2, :W // 2 * 2]
    return x.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))


def cnn_f64(feats, sd):
    d = {k: np.asarray(v, np.float64) for k, v in sd.items()}
    x = np.asarray(feats, np.float64)[:, None]
    x = pool2(np.maximum(conv3x3_f64(x, d["model.conv1.weight"], d["model.conv1.bias"]), 0))
    x = pool2(np.maximum(conv3x3_f64(x, d["model.conv2.weight"], d["model.conv2.bias"]), 0))
    x = x.reshape(x.shape[0], -1)
    x = np.maximum(x @ d["model.fc1.weight"].T + d["model.fc1.bias"], 0)
    emb = x @ d["model.fc2.weight"].T + d["model.fc2.bias"]
    h = np.maximum(emb @ d["classifier.0.weight"].T + d["classifier.0.bias"], 0)
    return emb, (h @ d["classifier.3.weight"].T + d["classifier.3.bias"]).ravel()


cfg = HeadConfig("cnn", (101, 64))
sd = synth_state_dict(cfg)
feats = synth_features(int(os.environ.get("N_CLIPS", "64")), (101, 64), seed=5)
emb64, logit64 = cnn_f64(feats, sd)
res = {}
for mode in ("f32", "bf16x9", "bf16x6", "f16x3"):
    m = HipModel(cfg, FrontendConfig(), state_dict=sd, conv_arith=mode)
    lg, _, emb = m.forward_features(feats, return_embedding=True)
    res[mode] = (emb.astype(np.float64), lg.astype(np.float64))
    print(f"{mode:7s} vs float64: max|d emb| {np.abs(res[mode][0] - emb64).max():.3e} (|emb| max {np.abs(emb64).max():.2f})"
          f"   max|d logit| {np.abs(res[mode][1] - logit64).max():.3e}   rms d logit {np.sqrt(np.mean((res[mode][1] - logit64) ** 2)):.3e}")
    m.close()
for a, b in (("bf16x9", "f32"), ("bf16x6", "f32"), ("bf16x6", "bf16x9"), ("f16x3", "f32"), ("f16x3", "bf16x9")):
    print(f"{a} vs {b}: max|d logit| {np.abs(res[a][1] - res[b][1]).max():.3e}  max|d emb| {np.abs(res[a][0] - res[b][0]).max():.3e}")
