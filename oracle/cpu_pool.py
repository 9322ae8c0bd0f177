"""All-core CPU baseline: a pool of worker PROCESSES, each running the oracle on batches of clips.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); used by bench.py's `cpu_baseline` leg.

This is how the reference itself uses a many-core host for its batch path: a pool over clips with
`ncpu = 0.6 x cores` workers (nanowakeword/data/transform_clips.py:441; AudioFeatures.py:195-207 maps
single-clip session calls over a ThreadPool, every session on one intra-op thread).  Workers are separate
interpreters (`python -m oracle.cpu_pool ...`: the bench process holds a HIP context, so nothing is forked),
pin BLAS to one thread and import numpy + oracle only.  They warm up, wait for a common wall-clock start
line and then run for the same budget, so every worker is timed while all the others are running.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_main(argv):
    spec = json.loads(argv[0])
    import numpy as np
    from threadpoolctl import threadpool_limits
    sys.path.insert(0, ROOT)
    import oracle
    from nanowakeword_amd.config import HeadConfig
    from nanowakeword_amd.synth import synth_pcm, synth_state_dict
    tb = np.load(spec["tables"])
    window, fb = tb["window"], tb["fb"]
    cfg = HeadConfig(spec["model_type"], tuple(spec["shape"]), **spec["kwargs"])
    sd = synth_state_dict(cfg)
    chunk = spec["chunk"]
    pcm = synth_pcm("noise", chunk, 16000, seed=spec["seed"])

    def one():
        lm = oracle.frontend_logmel(pcm, window, fb, n_mels=spec["n_mels"], center=spec["center"]).transpose(0, 2, 1)
        return oracle.model_forward(np.ascontiguousarray(lm), sd, cfg)

    with threadpool_limits(limits=1):
        one()                                       # warm-up (page-in, BLAS init)
        late = time.time() > spec["start_at"]
        while time.time() < spec["start_at"]:       # common start line
            time.sleep(0.002)
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += chunk
            dt = time.perf_counter() - t0
            if dt >= spec["budget_s"]:
                break
    print(json.dumps({"clips": n, "seconds": dt, "late": late}))


def pool_throughput(model_type, shape, n_mels, center, window, fb, workers, chunk=64, budget_s=4.0, spawn_s=None, **kwargs):
    """clips/s of `workers` single-threaded oracle processes running side by side: sum of the per-worker rates.
    Returns dict(rate, clips, seconds, workers, late)."""
    import numpy as np
    if spawn_s is None:
        spawn_s = 6.0 + 0.05 * workers             # interpreter start + imports + warm-up of every worker
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "tables.npz")
        np.savez(tp, window=np.asarray(window, np.float32), fb=np.asarray(fb, np.float32))
        start_at = time.time() + spawn_s
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        procs = []
        for i in range(workers):
            spec = {"model_type": model_type, "shape": list(shape), "n_mels": n_mels, "center": bool(center), "tables": tp,
                    "chunk": chunk, "budget_s": budget_s, "seed": 1000 + i, "start_at": start_at, "kwargs": kwargs}
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_pool", json.dumps(spec)], cwd=ROOT, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        res = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=spawn_s + budget_s + 120)
                res.append(json.loads(out.strip().splitlines()[-1]))
            except Exception:
                p.kill()
    if not res:
        raise RuntimeError("no CPU-pool worker reported")
    return {"rate": sum(r["clips"] / r["seconds"] for r in res), "clips": sum(r["clips"] for r in res),
            "seconds": max(r["seconds"] for r in res), "workers": len(res), "late": sum(1 for r in res if r["late"])}


if __name__ == "__main__":
    _worker_main(sys.argv[1:])
