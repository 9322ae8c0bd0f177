"""Worker for test_capi_gather_two_ranks (tests/test_gpu_variants.py): launched by torch.distributed.run, one rank per GPU.
Every rank scores its own shard through nww_forward_pcm_gather_dev (kernels + RCCL all-gather on one stream, C-ABI
communicator) and must end with the logits a single process computes for the whole batch, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanowakeword_amd.config import FrontendConfig, HeadConfig   # noqa: E402
from nanowakeword_amd.session import HipModel                     # noqa: E402
from nanowakeword_amd.synth import synth_pcm, synth_state_dict    # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    one_device = "--one-device" in sys.argv          # both ranks on cuda:0 (1-GPU box): torch.distributed over gloo carries the id
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if one_device:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = HeadConfig("cnn", (101, 64))
    m = HipModel(cfg, FrontendConfig(), device=local, state_dict=synth_state_dict(cfg))
    idt = torch.zeros(128, dtype=torch.uint8, device=torch.device("cpu") if one_device else dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(HipModel.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)                      # torch.distributed only carries the 128-byte id
    try:
        m.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
    except Exception as e:
        if one_device:                           # RCCL refuses two ranks on one device: say so and leave (the caller reports it)
            print(f"rank {rank} one-device communicator refused: {e}")
            m.close()
            dist.destroy_process_group()
            return
        raise
    B = 48                                      # clips per rank
    x_all = synth_pcm("noise", B * world, 16000, seed=5)
    pcm = torch.from_numpy(x_all[rank * B:(rank + 1) * B]).to(dev)
    out = torch.zeros(B * world, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    m.forward_pcm_gather_dev(pcm.data_ptr(), B, 16000, out.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    want, _ = m.forward_pcm(x_all)              # the whole batch on this rank's GPU
    assert np.array_equal(out.cpu().numpy(), want), f"rank {rank}: gathered logits differ from the single-process result"
    # the asynchronous form (gather on the handle's own stream, two buffers in flight): same vector, five steps
    outs = [torch.zeros(B * world, dtype=torch.float32, device=dev) for _ in range(2)]
    for k in range(5):
        m.forward_pcm_gather_async_dev(pcm.data_ptr(), B, 16000, outs[k & 1].data_ptr(), stream)
    m.gather_fence(stream)
    torch.cuda.synchronize(dev)
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want), f"rank {rank}: asynchronous gather differs"
    m.comm_destroy()
    m.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
