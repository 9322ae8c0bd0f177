"""N>1 path on CPU: gloo, world_size 2 (the RCCL path is the same code with backend 'nccl')."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT
from nanowakeword_amd.shard import max_shard, shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(n, w) or n == 0
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_all_gather_equals_single_process_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
