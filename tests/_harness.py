"""Shared test plumbing.

* oracle_particles(opts_init): the CPU oracle (oracle/liblcx_oracle.so, test infrastructure) driven
  through the SAME ctypes mirror as the product (libcloudphxx_amd.lgrngn.particles_t), prefix orc_.
* hip_particles(opts_init): the product (HIP library through the C ABI).
* small numpy restatements of the helpers the reference's python tests take from its `common` module.
"""
import ctypes
import os
import subprocess

import numpy as np

from libcloudphxx_amd import lgrngn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblcx_oracle.so")
_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("lcx_oracle.c", "orc_physics.h", "orc_tables.h")]
        fm = os.path.join(ORACLE_DIR, "liblcx_oracle_fastmath.so")
        if (not os.path.exists(ORACLE_SO)) or (not os.path.exists(fm)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        os.environ.setdefault("LCX_DATA_DIR", os.path.join(ROOT, "libcloudphxx_amd", "data"))
        _oracle = ctypes.CDLL(ORACLE_SO)
    return _oracle


def oracle_particles(opts_init):
    return lgrngn.particles_t(opts_init, np.float64, lib=oracle_lib(), prefix="orc_")


_oracle_fm = None


def oracle_fastmath_particles(opts_init):
    """the oracle source compiled with -O3 -ffast-math (how the reference's Release build is compiled);
    only for pinning against the reference's committed refdata"""
    global _oracle_fm
    if _oracle_fm is None:
        oracle_lib()
        _oracle_fm = ctypes.CDLL(os.path.join(ORACLE_DIR, "liblcx_oracle_fastmath.so"))
    return lgrngn.particles_t(opts_init, np.float64, lib=_oracle_fm, prefix="orc_")


def hip_particles(opts_init, real_t=np.float64):
    return lgrngn.factory(lgrngn.backend_t.HIP, opts_init, real_t)


def oracle_rng_preview(prt, calls):
    """calls: list of (kind, length); returns list of arrays = what the oracle's engine will generate next."""
    kinds = (ctypes.c_int * len(calls))(*[k for k, _ in calls])
    lens = (ctypes.c_size_t * len(calls))(*[n for _, n in calls])
    tot = sum(n for _, n in calls)
    out = np.empty(tot, dtype=np.float64)
    f = oracle_lib().orc_rng_preview
    rc = f(prt._h, kinds, lens, ctypes.c_int(len(calls)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == 0
    res, o = [], 0
    for _, n in calls:
        res.append(out[o:o + n].copy())
        o += n
    return res


# ---- the reference's `common` python module (bindings/python/common.hpp), restated with numpy
c_pd = 1005.
M_d = 0.02897
M_v = 1e-3 + 17e-3
R_d = 8.3144621 / M_d
R_v = 8.3144621 / M_v
p_1000 = 1e5


def T_of(th, rhod):          # common::theta_dry::T, theta_dry.hpp:24-41
    return (th * (rhod * R_d / p_1000) ** (R_d / c_pd)) ** (c_pd / (c_pd - R_d))


def p_of(rhod, rv, T):       # common::theta_dry::p, theta_dry.hpp:43-55
    return rhod * (R_d + rv * R_v) * T


def th_dry2std(th, rv):      # theta_dry.hpp:101-113
    return th / (1 + rv * R_v / R_d) ** (R_d / c_pd)


def lognormal_fn(mean_r, stdev, n_tot):
    """the exact python expression the reference's tests use for n(ln r)"""
    from math import exp, log, sqrt, pi

    def f(lnr):
        return n_tot * exp(-pow((lnr - log(mean_r)), 2) / 2 / pow(log(stdev), 2)) / log(stdev) / sqrt(2 * pi)
    return f
