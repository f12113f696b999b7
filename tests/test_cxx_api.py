"""The C++ host mirror (include/libcloudph++/lgrngn/*.hpp over the C ABI): a driver written against the
reference's C++ interface, built by examples/Makefile, run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "parcel_cxx")


def test_cxx_example_builds():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cxx_example_runs():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    out = subprocess.check_output([EXE], env=dict(os.environ, LCX_DATA_DIR=os.path.join(ROOT, "libcloudphxx_amd", "data"))).decode()
    vals = {l.split()[0]: l.split()[1:] for l in out.strip().splitlines()}
    assert vals["call_order_exception"] == ["1"]
    # the `<enum>_name` tables of the option headers (reference lgrngn/kernel.hpp:11-24 etc.), as a driver prints them
    assert vals["options"] == ["backend=HIP", "kernel=hall_pinsky_stratocumulus", "vt=beard77fast", "adve=implicit", "RH=pv_cc", "src=off", "n_kernels=12"]
    th, rv, sd = float(vals["parcel"][1]), float(vals["parcel"][3]), float(vals["parcel"][5])
    assert abs(th - 307.78) < 1e-4 * 307.78 and abs(rv - 1.7e-2) < 1e-3 * 1.7e-2 and sd == 100   # lgrngn_cond.py:52-56
    assert float(vals["box"][1]) == 300 and int(vals["box"][3]) == 300


@pytest.mark.gpu
def test_cxx_ring_example_round_trip():
    """examples/ring_cxx.cpp: factory<double>(multi_CUDA, opts_init) -- ONE object over two slabs (both on device 0 here), driven
    through the reference's C++ interface only; Courant number 1 for nx steps takes every super-droplet through both slab faces and
    back to its cell (tests/mpi/mpi_adve_test.cpp:196-255)"""
    exe = os.path.join(ROOT, "examples", "ring_cxx")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    env = dict(os.environ, LCX_DATA_DIR=os.path.join(ROOT, "libcloudphxx_amd", "data"), LCX_MULTI_DEVICE_MAP="0,0")
    out = subprocess.check_output([exe], env=env).decode().split()
    assert out[:3] == ["ring", "round_trip_identical", "1"], out
    assert float(out[4]) == 8 * 6 * 4 and int(out[6]) == 2 and int(out[8]) == 1, out
