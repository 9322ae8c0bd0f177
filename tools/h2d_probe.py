#!/usr/bin/env python3
"""Host-to-device upload rate of one batch of PCM (4096 x 16000 int16 = 131 MB, pinned) split over 1 / 2 / 4 / 8 copy streams,
on an idle GPU and beside the headline step running back to back on another stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel, torchaudio_tables
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

dev = torch.device("cuda", 0)
B, N = 4096, 16000
cfg, fe = HeadConfig("cnn", (101, 64)), FrontendConfig()
window, fb = torchaudio_tables(fe)
m = HipModel(cfg, fe, device=0, state_dict=synth_state_dict(cfg), window=window, mel_fb=fb)
m.reserve(B, N)
host = torch.from_numpy(synth_pcm("noise", B, N, seed=1)).pin_memory()
dbuf = torch.empty((B, N), dtype=torch.int16, device=dev)
pcm = host.to(dev)
lg = torch.empty(B, dtype=torch.float32, device=dev)
comp = torch.cuda.Stream(dev)


def upload(ncopy, reps):
    streams = [torch.cuda.Stream(dev) for _ in range(ncopy)]
    rows = (B + ncopy - 1) // ncopy
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c, s in enumerate(streams):
            with torch.cuda.stream(s):
                dbuf[c * rows:(c + 1) * rows].copy_(host[c * rows:(c + 1) * rows], non_blocking=True)
    for s in streams:
        s.synchronize()
    return B * N * 2 * reps / (time.perf_counter() - t0) / 1e9


for busy in (False, True):
    for ncopy in (1, 2, 4, 8):
        upload(ncopy, 3)
        if busy:
            with torch.cuda.stream(comp):
                for _ in range(400):                       # ~0.26 s of kernels queued: the uploads below run beside them
                    m.forward_pcm_dev(pcm.data_ptr(), B, N, lg.data_ptr(), 0, comp.cuda_stream)
        gbs = upload(ncopy, 40)
        torch.cuda.synchronize()
        print(f"{'beside the step' if busy else 'idle GPU      '} {ncopy} copy stream(s): {gbs:.1f} GB/s = {gbs * 1e9 / (N * 2) / 1e6:.2f} M clips/s")
