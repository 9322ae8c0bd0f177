#!/usr/bin/env python3
"""Does it matter where the PCM batch comes from?  The headline step (CNN head, 4096 clips) over 1 / 2 / 3 / 6 rotating copies of the batch
(131 MB each; the Infinity Cache holds 256 MiB): step time and the per-kernel times of the library's own event profile.
usage: python tools/pcm_rotation_probe.py [B=4096]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = HeadConfig(model_type="cnn", input_shape=(101, 64))
m = HipModel(cfg, FrontendConfig(n_mels=64), state_dict=synth_state_dict(cfg))
dev = torch.device("cuda", 0)
pcm0 = torch.from_numpy(synth_pcm("noise", B, 16000, seed=10)).to(dev)
lg = torch.empty(B, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
m.reserve(B, 16000)
for nbuf in (1, 2, 3, 6, 1):
    ring = [pcm0] + [pcm0.clone() for _ in range(nbuf - 1)]
    for k in range(12):
        m.forward_pcm_dev(ring[k % nbuf].data_ptr(), B, 16000, lg.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 120
    for k in range(K):
        m.forward_pcm_dev(ring[k % nbuf].data_ptr(), B, 16000, lg.data_ptr(), 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    m.set_profiling(True)
    for k in range(30):
        m.forward_pcm_dev(ring[k % nbuf].data_ptr(), B, 16000, lg.data_ptr(), 0, st)
    torch.cuda.synchronize()
    prof = {n.split(":")[0]: round(ms / max(c, 1), 4) for n, ms, c in m.get_profile() if c > 0}
    m.set_profiling(False)
    print(f"{nbuf} buffer(s): {dt * 1e3:.4f} ms per step = {B / dt / 1e6:.2f} M clips/s; kernels {prof}", flush=True)
    del ring
