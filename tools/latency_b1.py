"""Single-clip latency through the host entry points (what the interpreter path pays per 80 ms hop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

# The first ~0.1-0.3 s of a process have a slow stretch (a block of 300 calls reads 200-300 us where the same block later reads
# 68: runtime / power-state warm-up, not the model: "1, 8, 1, 8" and "8, 8, 1" orders both put it on their second block) - run
# half a second of calls before anything is timed.
_w = HipModel(HeadConfig("cnn", (101, 64)), FrontendConfig(), state_dict=synth_state_dict(HeadConfig("cnn", (101, 64))))
_t = time.perf_counter()
while time.perf_counter() - _t < 0.5:
    _w.forward_pcm(synth_pcm("noise", 8, 16000, seed=3))
_w.close()
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
