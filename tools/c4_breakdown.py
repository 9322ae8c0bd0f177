#!/usr/bin/env python3
"""Per-launch times of one C4 hop (CRNN-GRU head, S lock-step streams, 80 ms hop): HIP events on every hop."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel, torchaudio_tables
    from nanowakeword_amd.synth import synth_pcm, synth_state_dict
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rnn = sys.argv[2] if len(sys.argv) > 2 else "gru"
    dev = torch.device("cuda", 0)
    cfg, fe = HeadConfig("crnn", (101, 64), crnn_rnn_type=rnn), FrontendConfig()
    sd = synth_state_dict(cfg)
    window, fb = torchaudio_tables(fe)
    m = HipModel(cfg, fe, device=0, state_dict=sd, window=window, mel_fb=fb)
    stream = torch.cuda.current_stream(dev).cuda_stream
    hop = 1280
    m.stream_open(S, 16000, hop)
    chunks = torch.from_numpy(synth_pcm("noise", S, hop * 4, seed=1)).to(dev)
    parts = [chunks[:, k * hop:(k + 1) * hop].contiguous() for k in range(4)]
    logits = torch.empty(S, dtype=torch.float32, device=dev)
    for i in range(40):
        m.stream_push_dev(parts[i % 4].data_ptr(), logits.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for i in range(n):
        m.stream_push_dev(parts[i % 4].data_ptr(), logits.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.set_profiling(True)
    for i in range(50):
        m.stream_push_dev(parts[i % 4].data_ptr(), logits.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    prof = m.get_profile()
    print(json.dumps({"S": S, "rnn": rnn, "ms_per_hop": round(dt / n * 1e3, 4),
                      "kernel_ms": {k: round(ms / max(c, 1), 4) for k, ms, c in prof if c > 0}}, indent=1))
    m.close()


if __name__ == "__main__":
    main()
