"""Per-launch times (HIP events on the launch stream) of one head at one batch size: python tools/profile_head.py crnn 1024"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_state_dict

head, B = sys.argv[1], int(sys.argv[2])
shape = (64, 101) if head == "e2e_dnn" else (101, 64)
cfg = HeadConfig(head, shape)
m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
m.reserve(B, 16000)
pcm = torch.randint(-8192, 8192, (B, 16000), dtype=torch.int16, device="cuda:0")
lg = torch.empty(B, dtype=torch.float32, device="cuda:0")
s = torch.cuda.Stream()
for _ in range(3):
    m.forward_pcm_dev(pcm.data_ptr(), B, 16000, lg.data_ptr(), 0, s.cuda_stream)
torch.cuda.synchronize()
m.set_profiling(True)
for _ in range(10):
    m.forward_pcm_dev(pcm.data_ptr(), B, 16000, lg.data_ptr(), 0, s.cuda_stream)
torch.cuda.synchronize()
prof = m.get_profile()
tot = 0.0
for name, ms, cnt in prof:
    if cnt:
        print(f"{name:60s} {ms / cnt:8.4f} ms")
        tot += ms / cnt
print(f"{'sum':60s} {tot:8.4f} ms  -> {B / tot * 1e3:,.0f} clips/s")
