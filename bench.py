#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json on MI355X: clips/s, 1 s @ 16 kHz int16 PCM -> logit.

Workload (configs[1]): CNN head (CNNModel on (101,64) log-mel), batch 4096 clips per GPU, fused
STFT+mel HIP frontend, float32.  A "step" = one pass of the hot path over one resident batch:
frontend kernel -> conv1 -> conv2 -> fc1 -> fc2 -> classifier -> logits (+ for N>1 one RCCL
all-gather of the per-clip logits, the path's only exchange).  Weak scaling: every rank holds its
own 4096 clips.  PCM is resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract fields + "roofline" for the dominant kernel, measured
with HIP events on the launch stream inside the timed region, + "cpu_baseline": the numpy oracle
of the same workload timed on the host cores; kind "port").
"""
import argparse
import glob
import json
import os
import sys
import time

# The HIP runtime maps streams onto 4 hardware queues by default; this process creates more streams than that (one per model
# handle, torch's, the PCIe leg's copy and compute streams), and copy streams that share a queue with the compute stream
# serialise behind it: the PCIe-inclusive leg read 41 GB/s where the same pipeline alone does 56 (tools/h2d_pipeline.py).
# Must be set before the runtime initialises; the headline value does not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling
PEAK_F32_TFLOPS = 157.3        # dense f32 peak, MFMA f32 == vector rate (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 16 * 157.3  # dense bf16 MFMA peak = 16 x the f32 MFMA rate (same guide; 2495 TF measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps before the W warm-up steps until this much time has passed: the GPU needs ~0.1 s of load "
                         "to reach its sustained clocks (20 cold steps run 10 %% slower than the same steps a second later)")
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU (BASELINE config: 4096)")
    ap.add_argument("--head", default="cnn", help="cnn (BASELINE config 2, the headline) | bcresnet (config 3: --global-batch 65536 over 8 ranks) | "
                                                  "conformer (config 5: --global-batch 16384 over 8) | dnn | crnn | gru | e2e_dnn")
    ap.add_argument("--act-dtype", default=None, choices=["f32", "bf16", "f16"],
                    help="storage of the activations between the BcResNet head's kernels (BASELINE config 3 'bf16 activations': f16 is the "
                         "16-bit mode that holds 1e-2 on every test clip, bf16 the lower-accuracy variant; BASELINE.md)")
    ap.add_argument("--conv-arith", default="f16x3", choices=["f32", "bf16x9", "bf16x6", "f16x3"],
                    help="arithmetic of the MFMA contractions (all float32-grade; nww_config.conv_arith; the library default is f16x3)")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="HIP events around the launches of every n-th timed step (an event per launch boundary costs the "
                         "stream ~10 us: n = 1 slows the timed loop by ~10 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the self-audit legs (other conv arithmetics, the other BASELINE configs, PCIe-inclusive rate, sustained loop); N=1 only anyway")
    ap.add_argument("--sustain-seconds", type=float, default=6.5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: split this many clips over the ranks (BASELINE config 3: 65536 over 8); default 0 = "
                         "weak scaling with --batch clips per GPU")
    ap.add_argument("--gather", default="capi", choices=["capi", "torch"],
                    help="N>1: the logits all-gather through the C-ABI (nww_forward_pcm_gather_dev: RCCL on the kernels' stream) "
                         "or through torch.distributed")
    ap.add_argument("--debug-single-gpu", action="store_true",
                    help="control-flow check of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the gather "
                         "runs over gloo on host copies (never a measurement)")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sd, window, fb, seconds):
    """The oracle (numpy/BLAS restatement of the reference path) on the host cores: PCM -> logits.
      value          all cores the way the reference's batch path uses them: a pool of 0.6 x nproc single-threaded workers over
                     clip batches (transform_clips.py:441, AudioFeatures.py:195-207) - oracle/cpu_pool.py, separate processes
      value_1thread  one process, one BLAS thread (the reference interpreter's setting, nanointerpreter.py:955-959)
    plus BASELINE config 1 exactly (DNN head, (98,40) no-centre log-mel, batches of 32) the same two ways."""
    import oracle                                  # checker/baseline only, never on the product path
    from oracle.cpu_pool import pool_throughput
    from threadpoolctl import threadpool_limits
    from nanowakeword_amd.synth import synth_pcm
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, int(0.6 * ncpu))
    chunk = 64
    pcm = synth_pcm("noise", chunk, 16000, seed=123)

    def one():
        lm = oracle.frontend_logmel(pcm, window, fb).transpose(0, 2, 1)
        return oracle.model_forward(np.ascontiguousarray(lm), sd, cfg)

    def timed(fn, per, nthreads, budget):
        with threadpool_limits(limits=nthreads):
            fn()                                   # warm-up (BLAS threads, page-in)
            n, t0 = 0, time.perf_counter()
            while True:
                fn()
                n += per
                dt = time.perf_counter() - t0
                if dt >= budget:
                    return n, dt
    n1, dt1 = timed(one, chunk, 1, seconds * 0.25)
    kw = {k: getattr(cfg, k) for k in ("layer_dim", "n_blocks", "embedding_dim", "activation") if hasattr(cfg, k)}
    # the pool at 0.6 x nproc workers (the reference's rule) and at a smaller count: container CPU quotas and the host's memory
    # bandwidth decide which is faster; batches of 8 clips per task (the reference maps single clips)
    pools = {}
    sweep = sorted({w for w in (16, 32, 64, 128, workers) if 1 <= w <= max(1, ncpu)} | {max(1, min(16, ncpu))})
    for wk in sweep:
        pools[wk] = pool_throughput(cfg.model_type, cfg.input_shape, 64, True, window, fb, wk, chunk=8, budget_s=max(2.0, seconds * 0.2), **kw)
    pool = max(pools.values(), key=lambda r: r["rate"])
    rule_pool = pools.get(workers, pool)
    # what the container lets this process use: cgroup CPU quota (v2 cpu.max / v1 cfs), the affinity mask, NUMA nodes
    quota = "none"
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = "none" if q[0] == "max" else f"{int(q[0]) / int(q[1]):.1f} cpus"
    except (OSError, ValueError, IndexError):
        try:
            cq, cp = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = "none" if cq <= 0 else f"{cq / cp:.1f} cpus"
        except (OSError, ValueError):
            quota = "unknown"
    import glob as _glob
    numa_nodes = len(_glob.glob("/sys/devices/system/node/node[0-9]*")) or 1
    # BASELINE config 1 exactly: DNN head on (98,40) no-centre log-mel, batches of 32
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import torchaudio_tables
    from nanowakeword_amd.synth import synth_state_dict
    c1_cfg, c1_fe = HeadConfig("dnn", (98, 40)), FrontendConfig(n_mels=40, center=False)
    c1_sd = synth_state_dict(c1_cfg)
    c1_w, c1_fb = torchaudio_tables(c1_fe)
    c1_pcm = synth_pcm("noise", 32, 16000, seed=123)

    def c1_one():
        lm = oracle.frontend_logmel(c1_pcm, c1_w, c1_fb, n_mels=40, center=False).transpose(0, 2, 1)
        return oracle.model_forward(np.ascontiguousarray(lm), c1_sd, c1_cfg)
    c1n, c1dt = timed(c1_one, 32, 1, seconds * 0.1)
    c1_pool = pool_throughput("dnn", (98, 40), 40, False, c1_w, c1_fb, pool["workers"], chunk=32, budget_s=max(1.5, seconds * 0.15))
    return {"value": round(pool["rate"], 1), "unit": "clips/s", "cores": int(pool["workers"]), "kind": "port",
            "value_1thread": round(n1 / dt1, 1), "host_cpus": int(ncpu), "cpu_model": _cpu_model(),
            "pool_clips_per_s_by_workers": {str(k): round(v["rate"], 1) for k, v in pools.items()},
            "rule_0.6_x_nproc": {"workers": int(workers), "clips_per_s": round(rule_pool["rate"], 1)},
            "best": {"workers": int(pool["workers"]), "clips_per_s": round(pool["rate"], 1)},
            "quota": quota, "nproc": int(os.cpu_count() or 0), "affinity_cpus": int(ncpu), "numa_nodes": int(numa_nodes),
            "scaling_note": (f"the container's cgroup CPU quota is {quota}: workers beyond it only add context switches and cache contention, which is why "
                             "the pool peaks at the quota and the reference's 0.6 x nproc rule (counted on the HOST's CPUs) is slower"
                             if quota not in ("none", "unknown") else
                             "no CPU quota: the pool's rate is bound by shared L3 / memory bandwidth once the workers' dense-DFT bases and "
                             "head weights no longer fit their L2 slices (the sweep shows where)"),
            "config1_dnn_98x40_batch32": {"clips_per_s_1_thread": round(c1n / c1dt, 1),
                                          f"clips_per_s_{c1_pool['workers']}_workers": round(c1_pool["rate"], 1)},
            "sample": f"{pool['clips']} synthetic 1 s clips in batches of 8 through oracle/ (numpy float32, dense-DFT frontend + "
                      f"{cfg.model_type} head) by {pool['workers']} single-threaded worker processes side by side for {pool['seconds']:.1f} s each "
                      f"(best of {sweep} workers; 0.6 x {ncpu} = {workers} is the reference's batch-path pool size: transform_clips.py:441); value_1thread: {n1} clips "
                      f"in {dt1:.1f} s in one process on one BLAS thread (the reference interpreter's setting)"}


def sustained_leg(a, torch, dev, model, pcm, logits, B, N):
    """The headline step repeated for >= --sustain-seconds: the LAST thing the run does, long enough for an outside GPU-busy
    sampler with a 5 s period to see it (the CPU baseline before it leaves the GPU idle for ~12 s)."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(50):
            model.forward_pcm_dev(pcm.data_ptr(), B, N, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        n += 50
        dt = time.perf_counter() - t0
        if dt >= a.sustain_seconds:
            break
    return {"value": round(B * n / dt, 1), "unit": "clips/s", "seconds": round(dt, 2), "steps": n}


def step_work(cfg, fe, name, N=16000, act_bytes=4):
    """Algorithmic work per CLIP of one plan step of a BASELINE-config head (DESIGN.md 4 "Kernels"): ("mfma", flops) for the matrix-pipe
    kernels - float32-equivalent flops, priced against the dense 16-bit MFMA peak / 3 partial products under the default arithmetic -
    or ("hbm", bytes) for the ones that only move data.  None: no figure for this step (small / latency-bound)."""
    import re
    T, F = (cfg.input_shape[1], cfg.input_shape[0]) if cfg.model_type == "e2e_dnn" else cfg.input_shape
    frames = 1 + (N - (0 if fe.center else fe.n_fft)) // fe.hop_length
    if name.startswith("frontend"):
        return "hbm", 2 * N + 4 * fe.n_mels * frames
    mt = cfg.model_type
    if mt == "conformer":
        D, NH = cfg.conformer_d_model, cfg.conformer_n_head
        if name.startswith("ffn_x3"):
            return "mfma", 2 * T * 2 * D * 4 * D                       # linear1 + linear2
        if name.startswith("attn_x3"):
            return "mfma", 2 * T * D * 3 * D + 4 * T * T * D + 2 * T * D * D      # in_proj + q k^T + p v + out_proj
        if name.startswith("mha"):
            return "mfma", 4 * T * T * D
        if "in_proj" in name:
            return "mfma", 2 * T * D * 3 * D
        if "glu" in name:
            return "mfma", 2 * T * D * 2 * D
        if "out_proj" in name or "conv2(pw)" in name:
            return "mfma", 2 * T * D * D
        if "input_proj" in name:
            return "mfma", 2 * T * F * D
        if name.startswith("dwconv1d"):
            return "hbm", 2 * T * D * 4
        if name.startswith("layernorm+mean"):
            return "hbm", T * D * 4
    if mt == "bcresnet":
        # depthwise outputs d1, d2, d3 (channels, rows, columns): with the blocks chained only these (and as many strided shortcut samples) reach HBM
        h, w, d = T // 2, F // 2, []
        for ci, sh, sw in ((32, 2, 2), (64, 2, 2), (128, 2, 1)):
            h, w = (h - 1) // sh + 1, (w - 1) // sw + 1
            d.append((ci, h, w))
        if name.startswith("conv1_dw"):
            return "mfma", 2 * 9 * 32 * T * F + 2 * 9 * 32 * d[0][1] * d[0][2]
        m = re.search(r"model\.block(\d)\.pointwise", name)
        if m:
            i = int(m.group(1)) - 1
            c, h, w = d[i]
            nbytes = 2 * c * h * w * act_bytes                           # the depthwise output + the strided shortcut samples
            if i + 1 < 3 and "->" in name:
                c2, h2, w2 = d[i + 1]
                nbytes += 2 * c2 * h2 * w2 * act_bytes
            return "hbm", nbytes
    if mt == "cnn":
        H1, W1 = T // 2, F // 2
        if name.startswith("trunk"):
            return "mfma", 2 * 9 * 16 * (2 * H1) * (2 * W1) + 2 * 9 * 16 * 32 * (2 * (H1 // 2)) * (2 * (W1 // 2))
        if name.startswith("gemm:fc1"):
            return "mfma", 2 * 32 * (T // 4) * (F // 4) * 128
    if mt == "dnn" and name.startswith("gemm:layer1"):
        return "mfma", 2 * T * F * cfg.layer_dim
    return None


def leg_roofline(cfg, fe, B, prof, act_bytes=4):
    """roofline block of a config leg from its live per-launch profile: the step with the largest share of device time"""
    per = [(n, ms / max(c, 1)) for (n, ms, c) in prof if c > 0]
    if not per:
        return None
    name, ms = max(per, key=lambda r: r[1])
    w = step_work(cfg, fe, name, act_bytes=act_bytes)
    out = {"kernel": name.split(" [")[0], "avg_launch_ms": round(ms, 4), "share_of_step": round(ms / sum(r[1] for r in per), 3)}
    if w is None:
        out["bound"] = None
        return out
    bound, amount = w
    if bound == "hbm":
        ach, peak, unit = B * amount / 1e9 / (ms * 1e-3), PEAK_HBM_GBS, "GB/s"
        out["algorithmic_gb"] = round(B * amount / 1e9, 4)
    else:
        ach, peak, unit = B * amount / 1e12 / (ms * 1e-3), round(PEAK_BF16_TFLOPS / 3, 1), "TFLOP/s"
        out["algorithmic_gflop"] = round(B * amount / 1e9, 2)
        out["peak_definition"] = "dense 16-bit MFMA peak / 3 partial products per float32 product (default arithmetic f16x3)"
    out.update({"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)})
    return out


def config_legs(torch, dev):
    """The other BASELINE.json configs on this one GPU (the multi-GPU ones at their per-GPU batch), driver-observed:
    10 timed steps each with PCM resident in HBM, and max |dlogit| of 8 clips against the oracle.  Never part of `value`."""
    import oracle                                  # checker only
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel, torchaudio_tables
    from nanowakeword_amd.synth import synth_pcm, synth_state_dict
    stream = torch.cuda.current_stream(dev).cuda_stream
    legs = {}
    for key, name, cfg, fe, B, act_dtype in (
            ("C1", "dnn head, (98,40) log-mel (40-mel, no centre), batch 32", HeadConfig("dnn", (98, 40)), FrontendConfig(n_mels=40, center=False), 32, None),
            ("C3", "bcresnet head, (101,64), batch 8192 (= 65536 / 8 GPUs), fp32", HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, None),
            ("C3_bf16", "bcresnet head, (101,64), batch 8192, bf16 activations between kernels (act_dtype=bf16: the LOWER-ACCURACY 16-bit variant - 0.08 off on silence, 3e-2 on tones; kept because BASELINE config 3 names bf16)",
             HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, "bf16"),
            ("C3_f16", "bcresnet head, (101,64), batch 8192, 16-bit activations between kernels = THE row that answers BASELINE config 3's 'bf16 activations': act_dtype=f16, binary16 x plan-time powers of two - same bytes as bf16, 11 significant bits, 1e-2 on every test clip without carve-outs",
             HeadConfig("bcresnet", (101, 64)), FrontendConfig(), 8192, "f16"),
            ("C5", "conformer head, (101,64), batch 2048 (= 16384 / 8 GPUs), MFMA attention", HeadConfig("conformer", (101, 64)), FrontendConfig(), 2048, None)):
        sd = synth_state_dict(cfg)
        window, fb = torchaudio_tables(fe)
        m = HipModel(cfg, fe, device=dev.index, state_dict=sd, window=window, mel_fb=fb, act_dtype=act_dtype)
        pcm_h = synth_pcm("noise", B, 16000, seed=10)
        pcm = torch.from_numpy(pcm_h).to(dev)
        logits = torch.empty(B, dtype=torch.float32, device=dev)
        m.reserve(B, 16000)
        for _ in range(5):
            m.forward_pcm_dev(pcm.data_ptr(), B, 16000, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            m.forward_pcm_dev(pcm.data_ptr(), B, 16000, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        lm = oracle.frontend_logmel(pcm_h[:8], window, fb, n_mels=fe.n_mels, center=fe.center).transpose(0, 2, 1)
        ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
        legs[key] = {"workload": name, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "clips_per_s": round(B * steps / dt, 1),
                     "max_abs_dlogit": float(np.abs(logits[:8].cpu().numpy() - ref).max())}
        # the leg's own roofline: a separate profiled pass (per-launch HIP events on the launch stream) behind the timed one
        m.set_profiling(True)
        for _ in range(5):
            m.forward_pcm_dev(pcm.data_ptr(), B, 16000, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        prof = m.get_profile()
        m.set_profiling(False)
        legs[key]["kernel_ms"] = {n.split(" [")[0]: round(ms / max(c, 1), 4) for (n, ms, c) in prof if c > 0}
        legs[key]["launches_per_step"] = sum(1 for (n, ms, c) in prof if c > 0)
        legs[key]["roofline"] = leg_roofline(cfg, fe, B, prof, act_bytes=2 if act_dtype else 4)
        m.close()
    # C4: CRNN-GRU head, 1024 lock-step 10 s streams, one 80 ms hop per step.  Per hop the library computes what the hop invalidates
    # (12 of 101 log-mel frames, 5 of 25 pooled conv rows: per-stream rings, nww_stream.hip) - bit-identical to re-scoring the window
    cfg, fe = HeadConfig("crnn", (101, 64), crnn_rnn_type="gru"), FrontendConfig()
    sd = synth_state_dict(cfg)
    window, fb = torchaudio_tables(fe)
    m = HipModel(cfg, fe, device=dev.index, state_dict=sd, window=window, mel_fb=fb)
    S, hop = 1024, 1280
    m.stream_open(S, 16000, hop)
    chunk_h = synth_pcm("noise", S, hop * 16, seed=1)
    chunks = torch.from_numpy(chunk_h).to(dev)
    parts = [chunks[:, k * hop:(k + 1) * hop].contiguous() for k in range(16)]
    logits = torch.empty(S, dtype=torch.float32, device=dev)
    for i in range(15):                                       # fill the windows (12.5 hops)
        m.stream_push_dev(parts[i].data_ptr(), logits.data_ptr(), 0, stream)
    torch.cuda.synchronize(dev)
    steps = 10
    t0 = time.perf_counter()
    for i in range(steps):
        m.stream_push_dev(parts[(15 + i) % 16].data_ptr(), logits.data_ptr(), 0, stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # the last window of stream s = its last 16000 pushed samples
    order = [(15 + i) % 16 for i in range(steps)]
    hist = np.concatenate([chunk_h[:8, k * hop:(k + 1) * hop] for k in (list(range(15)) + order)], axis=1)[:, -16000:]
    lm = oracle.frontend_logmel(np.ascontiguousarray(hist), window, fb).transpose(0, 2, 1)
    ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
    legs["C4"] = {"workload": "crnn (GRU) head, 1024 lock-step streams, 80 ms hop, one 1 s window score per stream and hop (incremental: per-stream log-mel / conv-row rings)", "steps": steps,
                  "ms_per_step": round(dt / steps * 1e3, 4), "window_scores_per_s": round(S * steps / dt, 1),
                  "max_abs_dlogit": float(np.abs(logits[:8].cpu().numpy() - ref).max())}
    m.close()
    return legs


def audit_legs(a, torch, dev, cfg, fe, sd, window, fb, pcm_host, B, N, ref_logits):
    """Self-audit legs of the N = 1 line (never part of `value`):
      arith      the same step in the other arithmetics: plain f32 MFMA, three bf16 terms with nine (exact products) or six
                 partial products, two binary16 terms with three
      h2d_inclusive  PCM starting in pinned host memory: double-buffered uploads on a copy stream overlapped with the
                 previous batch's kernels, logits copied back to the host (the PCIe-inclusive rate; Gen5 x16 ceiling
                 = 63 GB/s / 32 kB per clip = 1.97 M clips/s)"""
    from nanowakeword_amd.session import HipModel
    out = {}
    stream = torch.cuda.current_stream(dev).cuda_stream
    pcm = torch.from_numpy(pcm_host).to(dev)
    logits = torch.empty(B, dtype=torch.float32, device=dev)
    arith = {}
    # every arithmetic against the SAME network evaluated in float64 by the oracle (checker only) on the device's own log-mel of the
    # first 32 clips: the arithmetics are float32-grade when they sit as close to exact arithmetic as the float32 MFMA does
    import oracle
    n_acc = 32
    acc_feats = acc_ref = None
    acc = {}
    for mode in ("f32", "bf16x9", "bf16x6", "f16x3"):
        m = HipModel(cfg, fe, device=dev.index, state_dict=sd, window=window, mel_fb=fb, conv_arith=mode)
        if acc_feats is None:
            acc_feats = np.ascontiguousarray(m.frontend(pcm_host[:n_acc]).transpose(0, 2, 1))
            acc_ref = oracle.model_forward(acc_feats, sd, cfg, dtype=np.float64).ravel()
        lg_acc, _ = m.forward_features(acc_feats)
        acc[mode] = float(np.abs(lg_acc.astype(np.float64) - acc_ref).max())
        if mode == a.conv_arith:
            m.close()
            continue
        m.reserve(B, N)
        for _ in range(3):
            m.forward_pcm_dev(pcm.data_ptr(), B, N, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m.forward_pcm_dev(pcm.data_ptr(), B, N, logits.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        arith[mode] = {"value": round(B * a.steps / dt, 1), "ms_per_step": round(dt / a.steps * 1e3, 4),
                       "max_abs_dlogit_vs_headline_arith": float(np.abs(logits.cpu().numpy() - ref_logits).max())}
        m.close()
    out["arith"] = arith
    out["arith_accuracy"] = {"clips": n_acc, "max_abs_dlogit_vs_float64": {k: float(f"{v:.3e}") for k, v in acc.items()},
                             "reference": "the same network in float64 (oracle, checker only) on the device's log-mel features"}
    m = HipModel(cfg, fe, device=dev.index, state_dict=sd, window=window, mel_fb=fb, conv_arith=a.conv_arith)
    m.reserve(B, N)
    # ---- PCIe-inclusive: pinned host PCM -> (copy stream) -> device double buffer -> kernels -> host logits
    # The staging buffers should live on the GPU's NUMA node (two-socket hosts: 53.8 vs 56.9 GB/s on this pool): probe the nodes,
    # bind to the best one while the buffers are allocated and first touched, restore the affinity afterwards.
    old_aff, numa_note = os.sched_getaffinity(0), "single NUMA node"
    try:
        nodes = {}
        for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
            cpus = set()
            for part in open(os.path.join(d, "cpulist")).read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= old_aff
            if cpus:
                nodes[os.path.basename(d)] = cpus
        if len(nodes) > 1:
            rates = {}
            probe_dev = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            for name, cpus in nodes.items():
                os.sched_setaffinity(0, cpus)
                buf = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
                buf.fill_(1)
                probe_dev.copy_(buf, non_blocking=True)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(4):
                    probe_dev.copy_(buf, non_blocking=True)
                torch.cuda.synchronize(dev)
                rates[name] = 4 * (64 << 20) / (time.perf_counter() - t0) / 1e9
                del buf
            best = max(rates, key=rates.get)
            os.sched_setaffinity(0, nodes[best])
            numa_note = "staging buffers pinned on " + best + " (probe GB/s: " + ", ".join(f"{k} {v:.1f}" for k, v in rates.items()) + ")"
    except OSError:
        pass
    host = [torch.from_numpy(pcm_host).pin_memory(), torch.from_numpy(np.roll(pcm_host, 1, axis=0).copy()).pin_memory()]
    NBUF = 3        # device buffers in flight (tools/h2d_pipeline.py: 56.0 / 56.4 / 56.6 GB/s with 2 / 3 / 4)
    dbuf = [torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(NBUF)]
    lbuf = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(NBUF)]
    hlog = [torch.empty(B, dtype=torch.float32).pin_memory() for _ in range(NBUF)]
    # the upload of a batch is split over TWO copy streams: 56.7 GB/s through this pipeline against 54 with one and 49 with
    # four (tools/h2d_pipeline.py; the bare uploads reach 56-57 GB/s idle or beside the step, tools/h2d_probe.py)
    NCOPY = 2
    copy_s, comp_s = [torch.cuda.Stream(dev) for _ in range(NCOPY)], torch.cuda.Stream(dev)
    up = [[torch.cuda.Event() for _ in range(NCOPY)] for _ in range(NBUF)]
    done = [torch.cuda.Event() for _ in range(NBUF)]
    rows = (B + NCOPY - 1) // NCOPY

    def run(k):
        for i in range(k):
            j = i % NBUF
            for c, cs in enumerate(copy_s):
                with torch.cuda.stream(cs):
                    cs.wait_event(done[j])                # the kernels that read dbuf[j] NBUF batches ago are finished
                    dbuf[j][c * rows:(c + 1) * rows].copy_(host[i & 1][c * rows:(c + 1) * rows], non_blocking=True)
                    up[j][c].record(cs)
            with torch.cuda.stream(comp_s):
                for c in range(NCOPY):
                    comp_s.wait_event(up[j][c])
                m.forward_pcm_dev(dbuf[j].data_ptr(), B, N, lbuf[j].data_ptr(), 0, comp_s.cuda_stream)
                hlog[j].copy_(lbuf[j], non_blocking=True)
                done[j].record(comp_s)
        torch.cuda.synchronize(dev)
    for e in done:
        e.record(comp_s)
    run(8)
    # long enough for the steady state to be what is measured: a batch is ~2.3 ms of upload, so the pipeline's fill and drain (one upload
    # that overlaps nothing, the last step and its copy back) were 6 % of a 20-batch run (r05: 50.9 GB/s of the link's 56-57)
    k = max(150, a.steps)
    t0 = time.perf_counter()
    run(k)
    dt = time.perf_counter() - t0
    last0 = max(i for i in range(k) if (i & 1) == 0)      # a batch uploaded from host[0] (= the timed clips)
    assert np.array_equal(hlog[last0 % NBUF].numpy(), ref_logits), "PCIe-inclusive path changed the logits"
    os.sched_setaffinity(0, old_aff)
    out["h2d_inclusive"] = {"value": round(B * k / dt, 1), "unit": "clips/s", "batches": k, "numa": numa_note,
                            "pcm_gb_per_s": round(B * k * N * 2 / dt / 1e9, 2),
                            "note": "pinned host int16 PCM, uploads on two copy streams into three device buffers under the previous "
                                    "batches' kernels, logits copied back; never the headline value (the link itself: "
                                    "56-57 GB/s = 1.79 M clips/s, tools/h2d_probe.py)"}
    m.close()
    return out


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from nanowakeword_amd.config import FrontendConfig, HeadConfig, head_macs
    from nanowakeword_amd.session import HipModel, torchaudio_tables
    from nanowakeword_amd.synth import synth_pcm, synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if a.debug_single_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.debug_single_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    fe = FrontendConfig()                                  # 16 kHz, 400/400/160, 64 mel, center
    shape = (64, 101) if a.head == "e2e_dnn" else (101, 64)
    cfg = HeadConfig(a.head, shape)
    sd = synth_state_dict(cfg)
    window, fb = torchaudio_tables(fe)
    arith = a.conv_arith
    model = HipModel(cfg, fe, device=local, state_dict=sd, window=window, mel_fb=fb, conv_arith=arith,
                     **({"act_dtype": a.act_dtype} if a.act_dtype and a.act_dtype != "f32" else {}))
    B, N = a.batch, 16000
    scaling = "weak"
    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        B, scaling = a.global_batch // world, "strong"
    pcm_host = synth_pcm("noise", B, N, seed=10 + rank)    # default_rng(10).integers(-8192, 8192)
    pcm = torch.from_numpy(pcm_host).to(dev)               # resident in HBM before timing
    # The timed loop re-reads ONE 131 MB batch (as in rounds 1-5: comparable), which fits the 256 MiB Infinity Cache - said on the line;
    # the `pcm_rotation` leg times the same step walking THREE copies of the batch (same clips, distinct addresses: 393 MB), so that the
    # frontend's reads come from HBM every step
    pcm_ring = [pcm] + ([pcm.clone() for _ in range(2)] if world == 1 else [pcm, pcm])
    logits = torch.empty(B, dtype=torch.float32, device=dev)
    gdev = torch.device("cpu") if a.debug_single_gpu else dev
    gathered = torch.empty(B * world, dtype=torch.float32, device=gdev) if world > 1 else None
    model.reserve(B, N)
    stream = torch.cuda.current_stream(dev).cuda_stream
    gather_via = "none"
    if world > 1:
        gather_via = "torch.distributed"
        if a.gather == "capi" and not a.debug_single_gpu:
            # the communicator of the C-ABI: rank 0 creates the RCCL id, torch.distributed only carries its 128 bytes
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(HipModel.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                model.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
                gather_via = "capi"
            except Exception as e:                                  # stay measurable: fall back to the torch collective
                print(f"[bench] rank {rank}: C-ABI communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
            ok = torch.tensor([1 if gather_via == "capi" else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)               # all ranks must agree on the collective they run
            if int(ok.item()) == 0:
                gather_via = "torch.distributed"
    gathered2 = None
    if gather_via == "capi":
        logits = gathered[rank * B:(rank + 1) * B]                  # this rank's slot of the gathered vector
        gathered2 = [gathered, torch.empty_like(gathered)]          # two steps in flight: the gather runs on the handle's own stream
    step_no = [0, 0]

    def step(rotate=False):
        pcm = pcm_ring[step_no[1] % 3] if rotate else pcm_ring[0]
        step_no[1] += 1
        if gather_via == "capi":
            # kernels on `stream`, the RCCL all-gather of step k on the library's side stream behind an event: step k + 1's kernels do
            # not wait for it (nww_forward_pcm_gather_async_dev); the fence below closes the timed region
            model.forward_pcm_gather_async_dev(pcm.data_ptr(), B, N, gathered2[step_no[0] & 1].data_ptr(), stream)
            step_no[0] += 1
            return
        model.forward_pcm_dev(pcm.data_ptr(), B, N, logits.data_ptr(), 0, stream)
        if world > 1:
            dist.all_gather_into_tensor(gathered, logits if not a.debug_single_gpu else logits.cpu())   # RCCL over xGMI: 4 B per clip

    t_pre = time.perf_counter()
    while True:                                                 # clock ramp (untimed, not counted in W)
        # N > 1: every step holds a collective, so all ranks must run the SAME number of pre-warm steps - rank 0's clock decides
        # (ranks deciding by their own clocks can disagree by a block of steps and dead-lock in the all-gather)
        go = time.perf_counter() - t_pre < a.prewarm_seconds
        if world > 1:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=gdev)
            dist.broadcast(flag, 0)
            go = bool(int(flag.item()))
        if not go:
            break
        for _ in range(10):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    model.set_profiling(max(1, a.profile_every))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if gather_via == "capi":
        model.gather_fence(stream)                              # every gather of the timed steps is inside the timed region
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof = model.get_profile()
    model.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness guard on the timed buffers (cheap): finite, and first clips == small-batch run
    lg = logits.cpu().numpy()
    if world > 1:       # every rank's shard must sit at its slot of the gathered vector(s), and every rank must hold the same vector
        for gbuf in (gathered2 if gathered2 else [gathered]):
            gv = gbuf.cpu().numpy()
            assert np.array_equal(gv[rank * B:(rank + 1) * B], lg), "all-gather placed a shard wrongly"
            chk = torch.tensor([float(np.float64(gv.astype(np.float64).sum()))], dtype=torch.float64, device=gdev)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert float(lo.item()) == float(hi.item()), "ranks hold different gathered vectors"
    assert np.isfinite(lg).all()
    l8, _ = model.forward_pcm(pcm_host[:8])
    assert np.array_equal(l8, lg[:8]), "batch-size dependence in the timed path"
    # the headline configuration's own max |dlogit| against the oracle (checker only; 16 clips of the timed batch, CPU, outside the timed region)
    max_dlogit = None
    if rank == 0:
        import oracle
        nchk = 16
        lm = oracle.frontend_logmel(pcm_host[:nchk], window, fb)
        if cfg.model_type != "e2e_dnn":
            lm = lm.transpose(0, 2, 1)
        ref = oracle.model_forward(np.ascontiguousarray(lm), sd, cfg).ravel()
        max_dlogit = float(np.abs(lg[:nchk] - ref).max())

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = B * world * a.steps / dt
        # ---- roofline of the dominant kernel (largest share of device time in the timed region)
        per = [(n.split(" [")[0], ms / max(c, 1), c, ms) for (n, ms, c) in prof if c > 0]      # " [f16x3]" marks a step's arithmetic
        dom = max(per, key=lambda r: r[3])
        kernel_ms = {n: round(avg, 4) for (n, avg, c, tot) in per}
        T, n_mels = 101, fe.n_mels
        algo = {   # algorithmic work per LAUNCH (B clips): DESIGN.md "Kernels"
            "frontend:fe_stft_mel_db_kernel": ("hbm", B * (2 * N + 4 * n_mels * T) / 1e9, "GB/s", PEAK_HBM_GBS),
        }
        if cfg.model_type == "cnn":
            H1, W1 = T // 2, n_mels // 2
            algo["conv3x3:conv1"] = ("mfma", B * 2 * 9 * 1 * 16 * (2 * H1) * (2 * W1) / 1e12, "TFLOP/s", PEAK_F32_TFLOPS)
            algo["conv3x3:conv2"] = ("mfma", B * 2 * 9 * 16 * 32 * (2 * (H1 // 2)) * (2 * (W1 // 2)) / 1e12, "TFLOP/s", PEAK_F32_TFLOPS)
            algo["gemm:fc1"] = ("mfma", B * 2 * 32 * (T // 4) * (n_mels // 4) * 128 / 1e12, "TFLOP/s", PEAK_F32_TFLOPS)
            # fused trunk: conv1 on the 2*H1 x 2*W1 positions that survive the floor pooling + conv2 on 2*H2 x 2*W2
            algo["trunk:conv1+pool+conv2+pool"] = ("mfma", algo["conv3x3:conv1"][1] + algo["conv3x3:conv2"][1], "TFLOP/s", PEAK_F32_TFLOPS)
            # trunk_x3: BOTH convolutions run as P bf16 partial products per float32 product on v_mfma_f32_32x32x16_bf16
            # (trunk_b.hip), so the matrix-pipe speed of light for the algorithmic (float32-equivalent) flops is bf16_peak / P.
            P = {"bf16x6": 6, "bf16x9": 9, "f16x3": 3}.get(arith, 6)
            f1, f2 = algo["conv3x3:conv1"][1], algo["conv3x3:conv2"][1]
            algo["trunk_x3:conv1+pool+conv2+pool"] = ("mfma", f1 + f2, "TFLOP/s", round(PEAK_BF16_TFLOPS / P, 1))
        name = dom[0]
        if name in algo:
            bound, work, unit, peak = algo[name]
            achieved = work / (dom[1] * 1e-3)
        else:
            bound, unit, peak = "mfma", "TFLOP/s", PEAK_F32_TFLOPS
            achieved = B * 2 * head_macs(cfg) / 1e12 / (sum(r[1] for r in per if not r[0].startswith("frontend")) * 1e-3)
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                ent = tj.get(name)
                if ent and ent.get("batch") == B:
                    traffic = ent["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        roofline = {"kernel": name, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel at this batch "
                                      "(tools/collect_profiles.py), not measured in this run" if traffic is not None else None,
                    "avg_launch_ms": round(dom[1], 4), "launches": dom[2]}
        if name.startswith("trunk_x3"):
            P = {"bf16x6": 6, "bf16x9": 9, "f16x3": 3}.get(arith, 6)
            f1, f2 = algo["conv3x3:conv1"][1], algo["conv3x3:conv2"][1]
            roofline["peak_definition"] = (f"dense bf16 / f16 MFMA peak ({PEAK_BF16_TFLOPS:.0f} TFLOP/s) / {P} partial products per float32 product: "
                                           "both convolutions are issued as split-operand 16-bit MFMAs")
            # rounds 2-3 priced conv1 at the f32-MFMA rate (it ran there in round 2); kept only so that rounds compare
            peak_r2 = (f1 + f2) / (f1 / PEAK_F32_TFLOPS + f2 / (PEAK_BF16_TFLOPS / P))
            roofline["frac_round2_definition"] = round(achieved / peak_r2, 4)
            roofline["note"] = ("peaks assume the 2.4 GHz boost clock; with all 256 CUs busy this kernel runs power-limited at "
                                "1.9-2.0 GHz (tools/ubench/trunk_trace.hip, DESIGN.md 4.10)")
            # back-to-back bf16 MFMAs with toggling operands on all 256 CUs sustain 5.63e10 wave-instructions per second of the
            # nominal 7.68e10 (1.85 GHz at the package's power limit; tools/ubench/power_mix.hip mode 4)
            roofline["frac_of_sustained_matrix_rate"] = round(achieved / peak / (5.63 / 7.68), 4)
            # how busy the two pipes the kernel is limited by actually are, from the committed PMC pass of this kernel at this batch
            # (profiles/traffic.json) and the live launch time: with the two-term arithmetic the matrix pipe is no longer the only
            # limiter - the launch takes about the SUM of its matrix time and its VALU time
            try:
                ent = json.load(open(tp)).get(name) or {}
                if ent.get("batch") == B and ent.get("sq_insts_mfma"):
                    cyc = 1024 * dom[1] * 1e-3 * 2.4e9
                    roofline["mfma_busy_frac_at_2.4GHz"] = round(ent["sq_insts_mfma"] * 32 / cyc, 4)
                    roofline["valu_issue_frac_at_2.4GHz"] = round(ent["sq_insts_valu"] * 4 / cyc, 4)
                    roofline["pipe_counters_source"] = "SQ_INSTS_MFMA x 32 clocks, SQ_INSTS_VALU x 4 clocks (profiles/traffic.json) / (1024 SIMDs x avg_launch_ms x 2.4 GHz)"
            except Exception:
                pass
            roofline["vs_f32_mfma_peak"] = round(achieved / PEAK_F32_TFLOPS, 3)      # the float32 products per second against the chip's float32 matrix peak
        fe_row = [r for r in per if r[0].startswith("frontend")]
        extra = {}
        if fe_row:
            fe_ms = fe_row[0][1]
            fe_gbs = B * (2 * N + 4 * n_mels * T) / 1e9 / (fe_ms * 1e-3)
            extra["stft_stage"] = {"bound": "hbm", "achieved": round(fe_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(fe_gbs / PEAK_HBM_GBS, 4),
                                   "f32_tflops": round(B * 1.27e6 / 1e12 / (fe_ms * 1e-3), 2),
                                   "clips_per_s_stage": round(B / (fe_ms * 1e-3), 0), "avg_launch_ms": round(fe_ms, 4)}
            # The stage's BINDING roofline is VALU issue, not HBM: SQ_INSTS_VALU wave-instructions (rocprofv3 --pmc pass of this
            # kernel at this batch, profiles/traffic.json) x 4 clocks each / (1024 SIMDs x the live launch time x 2.4 GHz)
            try:
                ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("frontend:fe_stft_mel_db_kernel", {})
                if ent.get("batch") == B and ent.get("sq_insts_valu"):
                    n_simd, clk = 1024, 2.4e9
                    extra["stft_stage"]["valu_issue_frac"] = round(ent["sq_insts_valu"] * 4 / (n_simd * fe_ms * 1e-3 * clk), 4)
                    extra["stft_stage"]["valu_issue_source"] = (f"SQ_INSTS_VALU {ent['sq_insts_valu']:.4g} per launch (profiles/traffic.json, rocprofv3 --pmc) "
                                                                "x 4 clocks / (1024 SIMDs x avg_launch_ms x 2.4 GHz)")
                    extra["stft_stage"]["binding"] = "valu_issue"
                    extra["stft_stage"]["binding_frac"] = extra["stft_stage"]["valu_issue_frac"]      # the stage's fraction of its REAL roof
            except Exception:
                pass
        out = {
            "metric": "clips/sec (1 s @16 kHz PCM->logits)", "value": round(value, 1), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.model_type} head on (101,64) log-mel, batch={B}/GPU, 1 s 16 kHz mono int16 "
                                   "clips, 64-mel 25 ms/10 ms center frontend, fused STFT+mel HIP kernel, fp32",
                       "pcm_buffers": "the timed loop re-reads one 131 MB batch, which fits the 256 MiB Infinity Cache (the frontend is VALU-bound; `pcm_rotation` is the same step over three rotating copies)",
                       "clips_per_gpu": B, "n_samples": N,
                       "conv_arith": {"f32": "conv2 on v_mfma_f32_32x32x2_f32",
                                      "bf16x9": "float32 operands split exactly into 3 bf16 terms, all 9 partial products on v_mfma_f32_32x32x16_bf16, f32 accumulate",
                                      "bf16x6": "float32 operands split exactly into 3 bf16 terms, the 6 partial products >= 2^-23 of a product on v_mfma_f32_32x32x16_bf16, f32 accumulate (float32-grade: DESIGN.md 4.2)",
                                      "f16x3": "float32 operands, scaled by plan-time powers of two, split into 2 binary16 terms (22-23 of 24 significant bits), the 3 partial products >= 2^-22 of a product on v_mfma_f32_32x32x16_f16, f32 accumulate (float32-MFMA accuracy against float64: DESIGN.md 2)"}[arith],
                       "parallelism": f"batch-split x{world}" + (f" + RCCL all-gather of logits ({gather_via}" + (", on a side stream, two steps in flight)" if gather_via == "capi" else ")") if world > 1 else "")},
            "max_abs_dlogit": max_dlogit,      # the timed logits of the first 16 clips against the oracle (north_star: <= 1e-4)
            "max_abs_dlogit_note": "16 clips of the timed batch, PCM -> logit, against oracle/ (numpy float32 restatement of the reference); checker only, outside the timed region",
            "gather_via": gather_via,          # "capi" = RCCL all-gather inside the C-ABI on the kernels' stream; "none" at N = 1
            "roofline": roofline,
            "kernel_ms": kernel_ms,
        }
        if world > 1 and a.gather == "capi" and gather_via != "capi":
            out["gather_fallback"] = "the C-ABI communicator could not be created: torch.distributed all_gather_into_tensor was timed instead"
        out.update(extra)
        extras = world == 1 and not a.no_extras
        if extras:
            # the same step over three rotating copies of the batch (393 MB > the Infinity Cache): PCM really comes from HBM
            t_pre = time.perf_counter()                         # (the clocks fell while the checker ran on the CPU: the same ramp as in front of the headline loop)
            while time.perf_counter() - t_pre < max(a.prewarm_seconds, 0.5):
                for _ in range(12):
                    step(True)
                torch.cuda.synchronize(dev)
            t0r = time.perf_counter()
            for _ in range(max(a.steps, 60)):
                step(True)
            torch.cuda.synchronize(dev)
            dtr = time.perf_counter() - t0r
            out["pcm_rotation"] = {"value": round(B * max(a.steps, 60) / dtr, 1), "unit": "clips/s", "ms_per_step": round(dtr / max(a.steps, 60) * 1e3, 4),
                                   "buffers": 3, "note": "never the headline: the headline re-reads one batch, as every round did"}
            out.update(audit_legs(a, torch, dev, cfg, fe, sd, window, fb, pcm_host, B, N, lg))
            out["configs"] = config_legs(torch, dev)
        if not a.no_cpu_baseline and world == 1:           # reported at N = 1 only (rank 0, the GPU box's host cores)
            out["cpu_baseline"] = cpu_baseline(cfg, sd, window, fb, a.cpu_seconds)
        if extras:                                         # last: the GPU is busy for the final >= 6 s of the run
            out["sustained"] = sustained_leg(a, torch, dev, model, pcm, logits, B, N)
        print(json.dumps(out))
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
