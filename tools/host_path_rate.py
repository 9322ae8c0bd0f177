"""PCIe-inclusive rate of the host entry point nww_forward_pcm (pageable numpy input, host logits out), B = 4096."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
cfg = HeadConfig("cnn", (101, 64))
m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
for B in (256, 4096):
    pcm = synth_pcm("noise", B, 16000, seed=1)
    m.reserve(B, 16000)
    for _ in range(3): m.forward_pcm(pcm)
    t0 = time.perf_counter(); n = 10
    for _ in range(n): m.forward_pcm(pcm)
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt*1e3:.3f} ms per call host->host = {B/dt:,.0f} clips/s ({B*32000/dt/1e9:.1f} GB/s of PCM)")
