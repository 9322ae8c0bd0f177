"""Probe: do the VALU-bound frontend and the MFMA-bound conv trunk overlap when enqueued on two streams?
Two handles of the same CNN model: A runs the head on a resident log-mel batch, B runs the frontend on resident PCM.
Prints ms per (frontend + head) pair when serialised on one stream and when issued on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_state_dict

B = int(os.environ.get("B", 4096)); N = 16000; K = 20
cfg = HeadConfig("cnn", (101, 64)); sd = synth_state_dict(cfg)
A = HipModel(cfg, FrontendConfig(), state_dict=sd); Bm = HipModel(cfg, FrontendConfig(), state_dict=sd)
A.reserve(B, N); Bm.reserve(B, N)
dev = torch.device("cuda:0")
pcm = torch.randint(-8192, 8192, (B, N), dtype=torch.int16, device=dev)
lm = torch.empty((B, 101, 64), dtype=torch.float32, device=dev)
lm2 = torch.empty_like(lm)
logits = torch.empty(B, dtype=torch.float32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
Bm.frontend_dev(pcm.data_ptr(), B, N, lm.data_ptr(), True, s1.cuda_stream); torch.cuda.synchronize()

def run(two):
    for it in range(K + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        Bm.frontend_dev(pcm.data_ptr(), B, N, lm2.data_ptr(), True, s1.cuda_stream)
        A.forward_features_dev(lm.data_ptr(), B, logits.data_ptr(), 0, (s2 if two else s1).cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

def solo(which):
    for it in range(K + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if which == "fe":
            Bm.frontend_dev(pcm.data_ptr(), B, N, lm2.data_ptr(), True, s1.cuda_stream)
        else:
            A.forward_features_dev(lm.data_ptr(), B, logits.data_ptr(), 0, s2.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

print("env", {k: v for k, v in os.environ.items() if k.startswith("NWW_")})
print("frontend alone %.3f ms | head alone %.3f ms" % (solo("fe"), solo("head")))
print("one stream %.3f ms | two streams %.3f ms" % (run(False), run(True)))
