#!/usr/bin/env python3
"""Where a B = 1 forward_pcm spends its time: host-to-host wall time, per-launch device time (HIP events), and the
same call with device-resident input/output (no copies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict

dev = torch.device("cuda", 0)
for name, cfg in (("cnn (first in process)", HeadConfig("cnn", (101, 64))), ("cnn", HeadConfig("cnn", (101, 64))), ("e2e_dnn", HeadConfig("e2e_dnn", (64, 101))), ("dnn", HeadConfig("dnn", (101, 64))), ("crnn", HeadConfig("crnn", (101, 64))), ("gru", HeadConfig("gru", (101, 64)))):
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
    for B in (1, 16):
        pcm = synth_pcm("noise", B, 16000, seed=3)
        m.reserve(B, 16000)
        for _ in range(50):
            m.forward_pcm(pcm)
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            m.forward_pcm(pcm)
        host = (time.perf_counter() - t0) / n * 1e6
        d_pcm = torch.from_numpy(pcm).to(dev)
        d_log = torch.empty(B, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(50):
            m.forward_pcm_dev(d_pcm.data_ptr(), B, 16000, d_log.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m.forward_pcm_dev(d_pcm.data_ptr(), B, 16000, d_log.data_ptr(), 0, stream)
            torch.cuda.synchronize()
        devcall = (time.perf_counter() - t0) / n * 1e6
        t0 = time.perf_counter()
        for _ in range(n):
            m.forward_pcm_dev(d_pcm.data_ptr(), B, 16000, d_log.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - t0) / n * 1e6
        graph_us = float("nan")
        try:                                                   # the same launches replayed from a captured graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cs = torch.cuda.current_stream(dev).cuda_stream
                m.forward_pcm_dev(d_pcm.data_ptr(), B, 16000, d_log.data_ptr(), 0, cs)
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
                torch.cuda.synchronize()
            graph_us = (time.perf_counter() - t0) / n * 1e6
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            graph_pipe = (time.perf_counter() - t0) / n * 1e6
            print(f"   graph replay + sync {graph_us:.0f} us | back-to-back {graph_pipe:.0f} us")
        except Exception as e:
            print("   graph capture failed:", repr(e)[:300])
        m.set_profiling(True)
        for _ in range(100):
            m.forward_pcm_dev(d_pcm.data_ptr(), B, 16000, d_log.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        prof = [(k, round(ms / max(c, 1) * 1e3, 1)) for k, ms, c in m.get_profile() if c]
        m.set_profiling(False)
        print(f"{name} B={B}: host-to-host {host:.0f} us | device buffers + sync {devcall:.0f} us | back-to-back {pipelined:.0f} us/call | per launch (us): {prof}")
    m.close()
