#!/usr/bin/env python3
"""The PCIe-inclusive leg of bench.py (pinned PCM -> copy streams -> device buffers -> step -> logits on the host) with its
knobs exposed: copy streams per batch, device buffers, logits copied back or not."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel, torchaudio_tables
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

dev = torch.device("cuda", 0)
B, N = 4096, 16000
if os.environ.get("BLAS_FIRST"):          # what bench.py has done before its PCIe leg: a multi-threaded numpy / oracle pass
    a_ = np.random.rand(3000, 3000).astype(np.float32)
    for _ in range(5):
        a_ = a_ @ a_ * 1e-3
    print("numpy BLAS pass done", float(a_[0, 0]))
if os.environ.get("SHOW_AFFINITY"):
    print("affinity:", len(os.sched_getaffinity(0)), "cpus; running on", open("/proc/self/stat").read().split()[38])
cfg, fe = HeadConfig("cnn", (101, 64)), FrontendConfig()
window, fb = torchaudio_tables(fe)
m = HipModel(cfg, fe, device=0, state_dict=synth_state_dict(cfg), window=window, mel_fb=fb)
m.reserve(B, N)
pcm_host = synth_pcm("noise", B, N, seed=1)


def leg(ncopy, nbuf, d2h, k=120):
    host = [torch.from_numpy(np.roll(pcm_host, i, axis=0).copy()).pin_memory() for i in range(2)]
    dbuf = [torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(nbuf)]
    lbuf = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(nbuf)]
    hlog = [torch.empty(B, dtype=torch.float32).pin_memory() for _ in range(nbuf)]
    copy_s, comp_s = [torch.cuda.Stream(dev) for _ in range(ncopy)], torch.cuda.Stream(dev)
    up = [[torch.cuda.Event() for _ in range(ncopy)] for _ in range(nbuf)]
    done = [torch.cuda.Event() for _ in range(nbuf)]
    rows = (B + ncopy - 1) // ncopy
    for e in done:
        e.record(comp_s)

    def run(n):
        for i in range(n):
            j = i % nbuf
            for c, cs in enumerate(copy_s):
                with torch.cuda.stream(cs):
                    cs.wait_event(done[j])
                    dbuf[j][c * rows:(c + 1) * rows].copy_(host[i & 1][c * rows:(c + 1) * rows], non_blocking=True)
                    up[j][c].record(cs)
            with torch.cuda.stream(comp_s):
                for c in range(ncopy):
                    comp_s.wait_event(up[j][c])
                m.forward_pcm_dev(dbuf[j].data_ptr(), B, N, lbuf[j].data_ptr(), 0, comp_s.cuda_stream)
                if d2h:
                    hlog[j].copy_(lbuf[j], non_blocking=True)
                done[j].record(comp_s)
        torch.cuda.synchronize(dev)
    run(6)
    t0 = time.perf_counter()
    run(k)
    dt = time.perf_counter() - t0
    return B * k / dt / 1e6, B * k * N * 2 / dt / 1e9


for ncopy, nbuf, d2h in ((4, 2, True), (4, 2, False), (2, 2, True), (1, 2, True), (2, 3, True), (1, 3, True), (2, 4, True)):
    v, g = leg(ncopy, nbuf, d2h)
    print(f"copy streams {ncopy}, device buffers {nbuf}, logits back {d2h}: {v:.3f} M clips/s ({g:.1f} GB/s)")
