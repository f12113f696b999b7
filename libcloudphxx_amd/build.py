"""Builds the HIP library (csrc/liblcx_hip.so) in-tree with hipcc for gfx950.

-ffp-contract=off: the parity build keeps IEEE operation order (no FMA contraction), see csrc/lcx_math.hpp.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "liblcx_hip.so")
# (source, extra flags).  lcx_cond_wq.hip: the condensation kernel that loops over batches of droplets, built without the machine-level
# loop-invariant code motion (csrc/lcx_cond_wq.hpp says why); everything else with the compiler's defaults.
SRCS = [("lcx_core.hip", []), ("lcx_cond_wq.hip", ["-mllvm", "-disable-machine-licm"])]
DEPS = ["lcx_core.hip", "lcx_cond_wq.hip", "lcx_cond_wq.hpp", "lcx_kernels.hpp", "lcx_math.hpp", "lcx_multi.hpp", "lcx_pool.hpp",
        os.path.join("..", "..", "include", "lcx.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
    objs, procs = [], []
    for src, extra in SRCS:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = common + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for obj in objs:
        os.remove(obj)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
