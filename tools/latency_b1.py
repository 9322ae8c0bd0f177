"""Single-clip latency through the host entry points (what the interpreter path pays per 80 ms hop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

for name, cfg in (("cnn", HeadConfig("cnn", (101, 64))), ("e2e_dnn", HeadConfig("e2e_dnn", (64, 101))), ("cnn", HeadConfig("cnn", (101, 64))),
                  ("dnn", HeadConfig("dnn", (101, 64))), ("crnn", HeadConfig("crnn", (101, 64)))):
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
    for B in (1, 8):
        pcm = synth_pcm("noise", B, 16000, seed=3)
        m.reserve(B, 16000)
        for _ in range(20):
            m.forward_pcm(pcm)
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            m.forward_pcm(pcm)
        dt = (time.perf_counter() - t0) / n
        print(f"{name:8s} B={B}: {dt*1e6:7.1f} us per forward_pcm (host in, host out), {len(m.describe_plan().splitlines())} launches")
    m.close()
