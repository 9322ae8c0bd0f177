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

HIP_SOURCES = ["nww_api.hip", "frontend.hip", "layers.hip", "gemm_x3.hip", "trunk.hip", "trunk_x3.hip", "fe_tables.cpp"]


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


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not _newer(LIB, _all_deps()):
        return LIB
    objs = []
    odir = os.path.join(PKG, "build")
    os.makedirs(odir, exist_ok=True)
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I",
               os.path.join(ROOT, "include"), "-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    return LIB


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
