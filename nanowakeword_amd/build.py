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

HIP_SOURCES = ["nww_api.hip", "frontend.hip", "frontend2.hip", "layers.hip", "gemm_x3.hip", "gemm_x3s.hip", "trunk.hip", "trunk_x3.hip", "conv3_x3.hip", "ffn_x3.hip", "lin_x3.hip", "dual_x3.hip", "mha_mfma.hip", "emb_stream.hip", "fe_tables.cpp"]


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


def _all_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "nww.h"))
    return deps


# hipcc's SLP pass pairs adjacent scalar f32 ops into v_pk_* instructions, which issue at half rate on gfx950 (no
# gain) and cost a v_mov per operand to build the register pairs - measured on the 25-point DFT body: 532 VALU /
# 120 VGPRs with it, 432 VALU / 62 VGPRs without.  Disabled for the VALU-bound files; NWW_SLP="all" / "none"
# overrides for A/B builds (together with NWW_LIB_PATH to keep two libraries side by side).
NO_SLP = {"frontend.hip", "frontend2.hip"}


def build_hip(force: bool = False, verbose: bool = False, out: str = LIB) -> str:
    if not force and not _newer(out, _all_deps()):
        return out
    objs = []
    tag = "" if out == LIB else "_" + os.path.splitext(os.path.basename(out))[0]
    odir = os.path.join(PKG, "build" + tag)
    os.makedirs(odir, exist_ok=True)
    mode = os.environ.get("NWW_SLP", "")
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        no_slp = mode == "none" or (mode != "all" and src in NO_SLP)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-fno-slp-vectorize"] if no_slp else []) + \
              ["-I", CSRC, "-I", os.path.join(ROOT, "include"), "-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    subprocess.run(cmd, check=True)
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
