#!/usr/bin/env python3
"""Does overlapping the step's kernels across two batches help?  Two handles on two streams, each running the headline
step (B = 4096 per call) back to back, against one handle doing the same number of steps alone.  The frontend draws
1150 W (issue-bound), the head sits at the 1400 W cap: if the runtime interleaves them the package could stay at the cap
all the time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel, torchaudio_tables
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

dev = torch.device("cuda", 0)
cfg, fe = HeadConfig("cnn", (101, 64)), FrontendConfig()
sd = synth_state_dict(cfg)
window, fb = torchaudio_tables(fe)
B, N, steps = int(os.environ.get("B", 4096)), 16000, 400
ms = [HipModel(cfg, fe, device=0, state_dict=sd, window=window, mel_fb=fb) for _ in range(2)]
pcm = [torch.from_numpy(synth_pcm("noise", B, N, seed=10 + i)).to(dev) for i in range(2)]
lg = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(2)]
st = [torch.cuda.Stream(dev) for _ in range(2)]
for m in ms:
    m.reserve(B, N)


def run(n_streams, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        for s in range(n_streams):
            ms[s].forward_pcm_dev(pcm[s].data_ptr(), B, N, lg[s].data_ptr(), 0, st[s].cuda_stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for n in (1, 2):
    run(n, 200)
t_end = time.time() + 2.0
while time.time() < t_end:
    run(2, 100)
for rep in range(3):
    d1 = run(1, steps)
    d2 = run(2, steps // 2)
    print(f"B={B}: one stream {B * steps / d1 / 1e6:.3f} M clips/s ({d1 / steps * 1e3:.4f} ms/step) | two streams "
          f"{B * steps / d2 / 1e6:.3f} M clips/s ({d2 / steps * 1e3:.4f} ms/step)")
ref = lg[0].cpu().numpy().copy()
run(1, 1)
assert np.array_equal(ref, lg[0].cpu().numpy())
