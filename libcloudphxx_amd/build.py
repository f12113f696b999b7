"""Builds the HIP library (csrc/liblcx_hip.so) in-tree with hipcc for gfx950.

-ffp-contract=off: the parity build keeps IEEE operation order (no FMA contraction), see csrc/lcx_math.hpp.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "liblcx_hip.so")
SRCS = ["lcx_core.hip"]
DEPS = ["lcx_core.hip", "lcx_kernels.hpp", "lcx_math.hpp", "lcx_multi.hpp", os.path.join("..", "..", "include", "lcx.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-result", "-o", OUT] + [os.path.join(CSRC, s) for s in SRCS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
