import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
cfg = HeadConfig("crnn", (101, 64)); m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
dev = torch.device("cuda", 0); S, hop = 1024, 1280
m.stream_open(S, 16000, hop)
chunks = torch.from_numpy(synth_pcm("noise", S, hop * 4, seed=1)).to(dev)
parts = [chunks[:, k * hop:(k + 1) * hop].contiguous() for k in range(4)]
lg = torch.empty(S, dtype=torch.float32, device=dev); s = torch.cuda.current_stream(dev).cuda_stream
for i in range(20): m.stream_push_dev(parts[i % 4].data_ptr(), lg.data_ptr(), 0, s)
torch.cuda.synchronize(); m.set_profiling(True)
for i in range(40): m.stream_push_dev(parts[i % 4].data_ptr(), lg.data_ptr(), 0, s)
torch.cuda.synchronize()
for n, ms, c in m.get_profile():
    if c: print(f"{n:60s} {ms / c * 1e3:8.1f} us")
