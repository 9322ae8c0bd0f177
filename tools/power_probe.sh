#!/bin/bash
# Power, clocks and temperature of the GPU while the headline step runs (rocm-smi sampled once a second beside a sustained loop).
# usage (GPU box): bash tools/power_probe.sh [head=cnn]
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -v "^$" | head -40
echo "=== under load"
python - <<'PY' &
import time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict
cfg = HeadConfig("cnn", (101, 64)); m = HipModel(cfg, FrontendConfig(), state_dict=synth_state_dict(cfg))
B, N = 4096, 16000
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(synth_pcm("noise", B, N, seed=10)).to(dev); lg = torch.empty(B, dtype=torch.float32, device=dev)
m.reserve(B, N); s = torch.cuda.current_stream(dev).cuda_stream
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    for _ in range(100): m.forward_pcm_dev(pcm.data_ptr(), B, N, lg.data_ptr(), 0, s)
    torch.cuda.synchronize(); n += 100
print("sustained ms/step", (time.time() - t0) / n * 1e3)
PY
sleep 5
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 1; done
wait
rocm-smi --showmaxpower 2>/dev/null | grep -v "^$" | head
