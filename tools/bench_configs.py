#!/usr/bin/env python3
"""Measure every BASELINE.json config on ONE MI355X (the multi-GPU ones at their per-GPU batch):
clips/s PCM->logit with PCM resident in HBM, per-launch times, max |dlogit| vs the oracle on a few clips.
Prints one JSON line per config.  (bench.py remains the contract benchmark for configs[1].)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import oracle
    from nanowakeword_amd.config import FrontendConfig, HeadConfig, head_macs
    from nanowakeword_amd.session import HipModel, torchaudio_tables
    from nanowakeword_amd.synth import synth_pcm, synth_state_dict
    dev = torch.device("cuda", 0)
    only = sys.argv[1:]
    configs = [
        ("C1 dnn 40-mel no-center B=32", HeadConfig("dnn", (98, 40)), FrontendConfig(n_mels=40, center=False), 32, 200),
        ("C1' dnn 40-mel no-center B=4096", HeadConfig("dnn", (98, 40)), FrontendConfig(n_mels=40, center=False), 4096, 30),
        ("C2 cnn 64-mel B=4096", HeadConfig("cnn", (101, 64)), FrontendConfig(), 4096, 30),
        ("C3 bcresnet 64-mel B=8192/GPU (fp32)", HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, 10),
        ("C3b bcresnet 64-mel B=8192/GPU (bf16 activations)", HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, 10),
        ("C3c bcresnet 64-mel B=8192/GPU (f16 activations)", HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, 10),
        ("C4 crnn-gru 64-mel streaming S=1024 x 125 hops", HeadConfig("crnn", (101, 64)), FrontendConfig(), 1024, 0),
        ("C5 conformer 64-mel B=2048/GPU", HeadConfig("conformer", (101, 64)), FrontendConfig(), 2048, 10),
        ("e2e_dnn 64-mel B=4096", HeadConfig("e2e_dnn", (64, 101)), FrontendConfig(), 4096, 20),
        ("gru 64-mel B=4096", HeadConfig("gru", (101, 64)), FrontendConfig(), 4096, 10),
        # the reference's other activations (model.py:81-87) at speed: each within 1.15 x of its ReLU row
        ("C2-gelu cnn 64-mel B=4096 activation=gelu", HeadConfig("cnn", (101, 64), activation="gelu"), FrontendConfig(), 4096, 30),
        ("C2-silu cnn 64-mel B=4096 activation=silu", HeadConfig("cnn", (101, 64), activation="silu"), FrontendConfig(), 4096, 30),
        ("C3-gelu bcresnet 64-mel B=8192/GPU activation=gelu", HeadConfig("bcresnet", (101, 64), activation="gelu"), FrontendConfig(), 8192, 10),
        ("C3-silu bcresnet 64-mel B=8192/GPU activation=silu", HeadConfig("bcresnet", (101, 64), activation="silu"), FrontendConfig(), 8192, 10),
    ]
    for name, cfg, fe, B, steps in configs:
        if only and not any(o in name for o in only):
            continue
        sd = synth_state_dict(cfg)
        window, fb = torchaudio_tables(fe)
        m = HipModel(cfg, fe, device=0, state_dict=sd, window=window, mel_fb=fb, act_dtype="bf16" if "bf16 activations" in name else "f16" if "f16 activations" in name else None)
        stream = torch.cuda.current_stream(dev).cuda_stream
        out = {"config": name, "head": cfg.model_type, "batch": B, "mmac_per_clip": round(head_macs(cfg) / 1e6, 2)}
        if "streaming" in name:
            S, hop, n_hops = B, 1280, 125
            m.stream_open(S, 16000, hop)
            chunks = torch.from_numpy(synth_pcm("noise", S, hop * 4, seed=1)).to(dev)
            logits = torch.empty(S, dtype=torch.float32, device=dev)
            for i in range(15):                                   # fill the windows (12.5 hops) + warm up
                m.stream_push_dev(chunks[:, (i % 4) * hop:(i % 4 + 1) * hop].contiguous().data_ptr(), logits.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            parts = [chunks[:, k * hop:(k + 1) * hop].contiguous() for k in range(4)]
            t0 = time.perf_counter()
            for i in range(n_hops):
                m.stream_push_dev(parts[i % 4].data_ptr(), logits.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out.update({"hops_per_s_per_stream": round(n_hops / dt, 1), "window_scores_per_s": round(S * n_hops / dt, 1),
                        "realtime_streams_supported": int(S * n_hops / dt / 12.5), "ms_per_hop": round(dt / n_hops * 1e3, 3)})
        else:
            pcm_h = synth_pcm("noise", B, 16000, seed=10)
            pcm = torch.from_numpy(pcm_h).to(dev)
            logits = torch.empty(B, dtype=torch.float32, device=dev)
            m.reserve(B, 16000)
            for _ in range(3):
                m.forward_pcm_dev(pcm.data_ptr(), B, 16000, logits.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            m.set_profiling(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                m.forward_pcm_dev(pcm.data_ptr(), B, 16000, logits.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            prof = m.get_profile()
            m.set_profiling(False)
            out.update({"clips_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
                        "head_tflops": round(2 * head_macs(cfg) * B * steps / dt / 1e12, 2),
                        "kernel_ms": {n: round(ms / max(c, 1), 4) for n, ms, c in prof if c > 0}})
            n_chk = 8
            lm = oracle.frontend_logmel(pcm_h[:n_chk], window, fb, n_mels=fe.n_mels, center=fe.center)
            if cfg.model_type != "e2e_dnn":
                lm = lm.transpose(0, 2, 1)
            ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
            out["max_abs_dlogit_vs_oracle"] = float(np.abs(logits[:n_chk].cpu().numpy() - ref).max())
            t0 = time.perf_counter(); k = 0
            while time.perf_counter() - t0 < 3.0:
                lm = oracle.frontend_logmel(pcm_h[:32], window, fb, n_mels=fe.n_mels, center=fe.center)
                lm = lm if cfg.model_type == "e2e_dnn" else lm.transpose(0, 2, 1)
                oracle.model_forward(np.ascontiguousarray(lm), sd, cfg); k += 32
            out["cpu_oracle_clips_per_s"] = round(k / (time.perf_counter() - t0), 1)
        print(json.dumps(out), flush=True)
        m.close()


if __name__ == "__main__":
    main()
