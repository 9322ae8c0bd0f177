#!/usr/bin/env python3
"""Soak test of the zero-copy small-call path (pinned staging buffer read by the kernels, completion word polled by the host):
N random-size calls (B = 1 .. 20, so both sides of the B <= 16 completion-word rule) whose logits must equal a table computed
once through the device-pointer entry point (B = 256: the bulk kernels), bit for bit - batch invariance of the small-batch
instances (chain GEMM, trunk strips, folded recurrent step ...) included.  GPU box.
usage: python tools/stress_small_calls.py [calls=100000] [head=cnn|dnn|crnn|gru|e2e_dnn|bcresnet|conformer]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict


def run(n_calls=100000, head="cnn", seed=0, log=print):
    """-> number of calls whose logits differ from the bulk-kernel table"""
    cfg = HeadConfig(head, (64, 101) if head == "e2e_dnn" else (101, 64))
    m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
    dev = torch.device("cuda", 0)
    pool = synth_pcm("noise", 256, 16000, seed=21)
    pd = torch.from_numpy(pool).to(dev)
    want = torch.empty(256, dtype=torch.float32, device=dev)
    m.forward_pcm_dev(pd.data_ptr(), 256, 16000, want.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    want = want.cpu().numpy()
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for i in range(n_calls):
        B = int(rng.integers(1, 21)); o = int(rng.integers(0, 256 - B))
        lg, pr = m.forward_pcm(pool[o:o + B])
        if not np.array_equal(lg, want[o:o + B]):
            bad += 1
            if bad < 5:
                log("MISMATCH at call", i, "B", B, "offset", o, lg[:4], want[o:o + 4])
    log(f"{head}: {n_calls} calls in {time.time() - t0:.1f} s, mismatches: {bad}")
    m.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100000, sys.argv[2] if len(sys.argv) > 2 else "cnn") else 0)
