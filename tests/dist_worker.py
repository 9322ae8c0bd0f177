"""Worker for tests/test_distributed.py: launched by torch.distributed.run with gloo, world_size 2 (CPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanowakeword_amd.shard import ShardedScorer, shard_bounds   # noqa: E402


def fake_forward(pcm):
    """Deterministic per-clip function (stands in for the HIP model on CPU): depends only on the clip."""
    x = pcm.to(torch.float32) / 32768.0
    return (x * x).mean(dim=1) * 3.0 - x[:, ::7].abs().mean(dim=1)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == int(os.environ["WORLD_SIZE"]) == 2
    for B in (8, 7, 1, 2, 33):
        g = np.random.default_rng(B)
        pcm = torch.from_numpy(g.integers(-8192, 8192, size=(B, 640)).astype(np.int16))
        calls = []

        def fwd(shard):
            calls.append(shard.shape[0])
            return fake_forward(shard)
        out = ShardedScorer(fwd).score(pcm)
        ref = fake_forward(pcm)
        assert out.shape == ref.shape, (out.shape, ref.shape)
        assert torch.equal(out, ref), f"rank {rank}: gathered logits differ from the single-process result at B={B}"
        lo, hi = shard_bounds(B, world, rank)
        assert calls == ([hi - lo] if hi > lo else []), (calls, lo, hi)     # only the local shard was computed
    # every rank must hold the same gathered vector
    chk = torch.tensor([float(out.sum())], dtype=torch.float64)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(torch.equal(lst[0], t) for t in lst)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
