"""Build libnwwhip.so (gfx950) in-tree with hipcc, and the g++ CPU emulator used by the non-GPU tests.

The .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnwwhip.so")
EMU_DIR = os.path.join(ROOT, "tests", "hostemu")
EMU_LIB = os.path.join(EMU_DIR, "libfe_emu.so")

HIP_SOURCES = ["nww_api.hip", "nww_plan.hip", "nww_stream.hip", "nww_comm.hip", "nww_emb.hip", "frontend2.hip", "layers.hip", "gemm_x3.hip", "trunk.hip", "trunk_b.hip", "conv3_x3.hip", "ffn_x3.hip", "lin_x3.hip", "dual_x3.hip", "bc_chain.hip", "rnn_x3.hip", "rnn_stream.hip", "mha_mfma.hip", "mha_h2.hip", "attn_x3.hip", "emb_stream.hip", "fe_tables.cpp"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# hipcc's SLP pass pairs adjacent scalar f32 ops into v_pk_* instructions, which issue at half rate on gfx950 (no
# gain) and cost a v_mov per operand to build the register pairs - measured on the 25-point DFT body: 532 VALU /
# 120 VGPRs with it, 432 VALU / 62 VGPRs without.  Disabled for the VALU-bound files; NWW_SLP="all" / "none"
# overrides for A/B builds (together with NWW_LIB_PATH to keep two libraries side by side).
# Round 3: a packed f32 instruction is also starved by any matrix-pipe wave on its SIMD; A/B over every file
# (NWW_SLP=none against the default, tools/bench_configs.py): BcResNet front 0.534 -> 0.515, its depthwise kernels 0.118 /
# 0.102 -> 0.114 / 0.094, dual_x3 0.172 / 0.242 -> 0.163 / 0.236, ffn_x3 -1 %; mha_mfma 0.318 -> 0.366 and conv3_x3 +1 % (they keep SLP).
NO_SLP = {"frontend2.hip", "trunk_b.hip", "dual_x3.hip", "bc_chain.hip", "layers.hip", "ffn_x3.hip"}


def build_hip(force: bool = False, verbose: bool = False, out: str = LIB) -> str:
    """Compile every source whose object is older than it (or than any header) - up to NWW_BUILD_JOBS (default 6) hipcc
    processes at a time - and link.  `force` recompiles everything."""
    from concurrent.futures import ThreadPoolExecutor
    tag = "" if out == LIB else "_" + os.path.splitext(os.path.basename(out))[0]
    odir = os.path.join(PKG, "build" + tag)
    os.makedirs(odir, exist_ok=True)
    mode = os.environ.get("NWW_SLP", "")
    extra = os.environ.get("NWW_HIPCC_FLAGS", "").split()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "nww.h")]
    jobs, objs = [], []
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and not _newer(obj, [sp] + headers):
            continue
        no_slp = mode == "none" or (mode != "all" and src in NO_SLP)
        jobs.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-fno-slp-vectorize"] if no_slp else []) + extra +
                    ["-I", CSRC, "-I", os.path.join(ROOT, "include"), "-x", "hip", "-c", sp, "-o", obj])
    if not jobs and os.path.exists(out) and not _newer(out, objs):
        return out

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("NWW_BUILD_JOBS", "6")))) as ex:
        list(ex.map(run, jobs))
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


def build_emu(force: bool = False) -> str:
    srcs = [os.path.join(EMU_DIR, "fe_emu.cpp"), os.path.join(CSRC, "fe_tables.cpp")]
    deps = srcs + [os.path.join(CSRC, "fe_steps.h"), os.path.join(CSRC, "fe_tables.h")]
    if not force and not _newer(EMU_LIB, deps):
        return EMU_LIB
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, "-o", EMU_LIB] + srcs, check=True)
    return EMU_LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_emu(force="--force" in sys.argv))
