#!/usr/bin/env python3
"""Pinned-host -> device upload rate of one 131 MB batch (4096 x 16000 int16) split over k copy streams."""
import time, torch
dev = torch.device("cuda", 0)
B, N = 4096, 16000
host = torch.randint(-8192, 8192, (B, N), dtype=torch.int16).pin_memory()
dst = torch.empty((B, N), dtype=torch.int16, device=dev)
for k in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    rows = B // k
    def run(n):
        for _ in range(n):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dst[i * rows:(i + 1) * rows].copy_(host[i * rows:(i + 1) * rows], non_blocking=True)
        torch.cuda.synchronize()
    run(3)
    t0 = time.perf_counter(); run(20); dt = time.perf_counter() - t0
    print(f"{k} stream(s): {B * N * 2 * 20 / dt / 1e9:.1f} GB/s -> {B * 20 / dt / 1e6:.2f} M clips/s")
