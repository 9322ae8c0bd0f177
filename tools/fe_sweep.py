#!/usr/bin/env python3
"""Frontend-only sweep over non-default frontend parameters (n_mels, centre, clip length) at B = 4096 device-resident clips: ms per launch, clips/s and
algorithmic GB/s (2 bytes per sample in, 4 per log-mel value out).  Looks for parameter values that drop onto a slower variant of the kernel.
usage (GPU box): python tools/fe_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict


def main():
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("B", 4096))
    for n_mels in (32, 40, 64, 80, 96, 128):
        for center in (True, False):
            for N in (8000, 16000, 32000, 48000):
                fe = FrontendConfig(n_mels=n_mels, center=center)
                T = 1 + N // 160 if center else 1 + (N - 400) // 160
                cfg = HeadConfig("dnn", (T, n_mels))
                try:
                    m = HipModel(cfg, fe, state_dict=synth_state_dict(cfg))
                except Exception as e:
                    print(f"n_mels={n_mels} center={center} N={N}: refused: {str(e)[:80]}")
                    continue
                pcm = torch.from_numpy(synth_pcm("noise", 64, N, seed=1)).to(dev).repeat(B // 64, 1).contiguous()
                out = torch.empty((B, T, n_mels), dtype=torch.float32, device=dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                m.reserve(B, N)
                for _ in range(3):
                    m.frontend_dev(pcm.data_ptr(), B, N, out.data_ptr(), 1, stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); n = 20
                for _ in range(n):
                    m.frontend_dev(pcm.data_ptr(), B, N, out.data_ptr(), 1, stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                gb = B * (2 * N + 4 * T * n_mels) / dt / 1e9
                print(f"n_mels={n_mels:3d} center={int(center)} N={N:5d} (T={T:3d}): {dt * 1e3:7.3f} ms  {B / dt / 1e6:6.2f} M clips/s  {B * T / dt / 1e9:5.2f} G frames/s  {gb:6.0f} GB/s", flush=True)
                m.close()


if __name__ == "__main__":
    main()
