#!/usr/bin/env python3
"""A/B of a Conformer plan knob on the GPU box: head-only forward at a few batch sizes with the knob off and on (two subprocesses, the
knobs are read once per process), logits against each other and against the oracle, per-launch times.
usage: python tools/attn_check.py [KNOB=NWW_ATTN_FUSED] [B ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r'''
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["NWW_ROOT"])
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict
kw = json.loads(sys.argv[1]); Bs = json.loads(sys.argv[2])
cfg = HeadConfig(**kw)
sd = synth_state_dict(cfg)
m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[1]), state_dict=sd)
dev = torch.device("cuda", 0)
out = {"plan": m.describe_plan() if hasattr(m, "describe_plan") else "", "runs": {}}
for B in Bs:
    xh = synth_features(B, cfg.input_shape, seed=3)
    x = torch.from_numpy(xh).to(dev)
    lg = torch.empty(B, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    m.reserve(B, 0)
    for _ in range(3):
        m.forward_features_dev(x.data_ptr(), B, lg.data_ptr(), 0, st)
    torch.cuda.synchronize()
    m.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(10):
        m.forward_features_dev(x.data_ptr(), B, lg.data_ptr(), 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    prof = {n: round(ms / max(c, 1), 4) for n, ms, c in m.get_profile() if c > 0}
    m.set_profiling(False)
    out["runs"][str(B)] = {"ms": round(dt * 1e3, 4), "kernel_ms": prof, "logits": lg.cpu().numpy().astype(float).tolist()}
print("RESULT" + json.dumps(out))
'''


def run(env, kw, Bs):
    e = dict(os.environ, NWW_ROOT=ROOT, **env)
    r = subprocess.run([sys.executable, "-c", WORKER, json.dumps(kw), json.dumps(Bs)], env=e, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT"):
            return json.loads(line[6:])
    raise RuntimeError(r.stdout[-2000:] + r.stderr[-4000:])


def main():
    import numpy as np
    import oracle
    from nanowakeword_amd.config import HeadConfig
    from nanowakeword_amd.synth import synth_features, synth_state_dict
    args = sys.argv[1:]
    knob = "NWW_ATTN_FUSED"
    if args and "=" not in args[0] and not args[0].isdigit():
        knob = args.pop(0)
    Bs = [int(a) for a in args] or [5, 37, 2048]
    for kw in (dict(model_type="conformer", input_shape=(101, 64)), dict(model_type="conformer", input_shape=(16, 96)),
               dict(model_type="conformer", input_shape=(128, 64), n_blocks=2)):
        off = run({knob: "0"}, kw, Bs)
        on = run({knob: "1"}, kw, Bs)
        cfg = HeadConfig(**kw)
        sd = synth_state_dict(cfg)
        print(f"== {kw}")
        for B in Bs:
            a, b = np.array(off["runs"][str(B)]["logits"]), np.array(on["runs"][str(B)]["logits"])
            k = min(B, 8)
            ref = oracle.model_forward(synth_features(B, cfg.input_shape, seed=3)[:k], sd, cfg).ravel()
            print(f"B={B}: {knob}=0 {off['runs'][str(B)]['ms']} ms, =1 {on['runs'][str(B)]['ms']} ms; max|on-off| {np.abs(a - b).max():.3e}; "
                  f"vs oracle off {np.abs(a[:k] - ref).max():.3e} on {np.abs(b[:k] - ref).max():.3e}; finite {np.isfinite(b).all()}")
        B = str(Bs[-1])
        print("  off:", json.dumps(off["runs"][B]["kernel_ms"]))
        print("  on :", json.dumps(on["runs"][B]["kernel_ms"]))


if __name__ == "__main__":
    main()
