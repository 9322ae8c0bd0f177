#!/usr/bin/env python3
"""Performance sweep over NON-default head hyper-parameters (the reference's factory takes them all: model.py:81-90,212-261): head-only
forward at B = 2048 from device-resident features, ms per step and float32-equivalent TFLOP/s from head_macs().  Looks for cliffs - a shape or
option that drops onto a general fallback kernel - the way the GELU / SiLU rows of tools/bench_configs.py found one.  Prints one line per config.
usage (GPU box): python tools/perf_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig, head_macs
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict

CASES = [
    ("cnn default", dict(model_type="cnn", input_shape=(101, 64))),
    ("cnn emb 128", dict(model_type="cnn", input_shape=(101, 64), embedding_dim=128)),
    ("cnn (98,40)", dict(model_type="cnn", input_shape=(98, 40))),
    ("cnn (16,96)", dict(model_type="cnn", input_shape=(16, 96))),
    ("dnn default", dict(model_type="dnn", input_shape=(101, 64))),
    ("dnn 256 x 3 blocks gelu", dict(model_type="dnn", input_shape=(101, 64), layer_dim=256, n_blocks=3, activation="gelu")),
    ("dnn lite 8/1/8", dict(model_type="dnn", input_shape=(16, 96), layer_dim=8, n_blocks=1, embedding_dim=8)),
    ("crnn gru default", dict(model_type="crnn", input_shape=(101, 64))),
    ("crnn lstm", dict(model_type="crnn", input_shape=(101, 64), crnn_rnn_type="lstm")),
    ("crnn gru L=64 silu", dict(model_type="crnn", input_shape=(101, 64), layer_dim=64, activation="silu")),
    ("crnn 4 stages", dict(model_type="crnn", input_shape=(96, 64), crnn_cnn_channels=[16, 32, 64, 64])),
    ("crnn (16,96)", dict(model_type="crnn", input_shape=(16, 96))),
    ("gru default", dict(model_type="gru", input_shape=(101, 64))),
    ("gru L=64 2 layers", dict(model_type="gru", input_shape=(101, 64), layer_dim=64, n_blocks=2)),
    ("gru (16,96)", dict(model_type="gru", input_shape=(16, 96))),
    ("gru L=200", dict(model_type="gru", input_shape=(101, 64), layer_dim=200)),
    ("gru L=96", dict(model_type="gru", input_shape=(101, 64), layer_dim=96)),
    ("gru L=256", dict(model_type="gru", input_shape=(101, 64), layer_dim=256)),
    ("gru L=160 2 layers", dict(model_type="gru", input_shape=(101, 64), layer_dim=160, n_blocks=2)),
    ("crnn lstm L=256", dict(model_type="crnn", input_shape=(101, 64), crnn_rnn_type="lstm", layer_dim=256)),
    ("crnn lstm L=100", dict(model_type="crnn", input_shape=(101, 64), crnn_rnn_type="lstm", layer_dim=100)),
    ("crnn [32,64,64]", dict(model_type="crnn", input_shape=(101, 64), crnn_cnn_channels=[32, 64, 64])),
    ("bcresnet default", dict(model_type="bcresnet", input_shape=(101, 64))),
    ("bcresnet (98,40)", dict(model_type="bcresnet", input_shape=(98, 40))),
    ("bcresnet (16,96)", dict(model_type="bcresnet", input_shape=(16, 96))),
    ("conformer default", dict(model_type="conformer", input_shape=(101, 64))),
    ("conformer 2 blocks", dict(model_type="conformer", input_shape=(101, 64), n_blocks=2)),
    ("conformer d 256 / 4 heads", dict(model_type="conformer", input_shape=(101, 64), conformer_d_model=256)),
    ("conformer d 96 / 2 heads", dict(model_type="conformer", input_shape=(101, 64), conformer_d_model=96, conformer_n_head=2)),
    ("conformer (16,96)", dict(model_type="conformer", input_shape=(16, 96))),
    ("e2e default", dict(model_type="e2e_dnn", input_shape=(64, 101))),
    ("e2e gelu", dict(model_type="e2e_dnn", input_shape=(64, 101), activation="gelu")),
]


def main():
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("B", 2048))
    only = sys.argv[1:]
    for name, kw in CASES:
        if only and not any(o in name for o in only):
            continue
        cfg = HeadConfig(**kw)
        try:
            m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[0] if cfg.model_type == "e2e_dnn" else cfg.input_shape[1]), state_dict=synth_state_dict(cfg))
        except Exception as e:
            print(f"{name:28s} refused: {e}")
            continue
        x = torch.from_numpy(synth_features(B, cfg.input_shape, seed=1)).to(dev)
        out = torch.empty(B, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        m.reserve(B, 0)
        for _ in range(3):
            m.forward_features_dev(x.data_ptr(), B, out.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        m.set_profiling(True)
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            m.forward_features_dev(x.data_ptr(), B, out.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        prof = sorted(((ms / max(c, 1), nm) for nm, ms, c in m.get_profile() if c > 0), reverse=True)[:3]
        tf = 2 * head_macs(cfg) * B / dt / 1e12
        print(f"{name:28s} {dt * 1e3:8.3f} ms  {head_macs(cfg) / 1e6:7.2f} MMAC/clip {tf:7.1f} TFLOP/s   top: " + "; ".join(f"{ms:.3f} {nm[:46]}" for ms, nm in prof), flush=True)
        m.close()


if __name__ == "__main__":
    main()
