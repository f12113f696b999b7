"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/lcx.h declares (no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcx_[a-z0-9_A-Z]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("lcx_create", "lcx_init", "lcx_step_sync", "lcx_step_async", "lcx_outbuf", "lcx_diag_puddle", "lcx_migrate_pack"):
        assert must in syms


def test_hip_library_exports_every_declared_symbol():
    from libcloudphxx_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_oracle_exports_the_same_surface():
    from _harness import oracle_lib
    lib = oracle_lib()
    # the oracle mirrors the ABI (prefix orc_) for everything the tests drive through the shared harness
    skip = {"lcx_dev_alloc", "lcx_dev_free", "lcx_dev_copy", "lcx_dev_sync", "lcx_timings", "lcx_set_profiling",
            "lcx_rng_replay_push", "lcx_rng_replay_pending", "lcx_math_probe",
            "lcx_create_multi", "lcx_multi_dev_count", "lcx_multi_slab"}     # (the oracle's ring is LocalRing in tests/_harness.py)
    missing = [s for s in declared_symbols() if s not in skip and not hasattr(lib, "orc_" + s[4:])]
    assert not missing, missing


def test_no_gpu_means_loud_failure():
    """without a HIP device lcx_create must fail with a message, never fall back to a CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libcloudphxx_amd import lgrngn
    oi = lgrngn.opts_init_t()
    oi.dt, oi.sd_conc, oi.n_sd_max = 1, 10, 10
    with pytest.raises(RuntimeError):
        lgrngn.factory(lgrngn.backend_t.HIP, oi)


def test_multi_backend_without_gpu_fails_loudly():
    """factory(multi_CUDA) is the native multi-device object; without a device it raises like the single-device one"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libcloudphxx_amd import lgrngn
    oi = lgrngn.opts_init_t()
    oi.nx, oi.x1, oi.dt, oi.sd_conc, oi.n_sd_max = 4, 4., 1, 10, 100
    for b in (lgrngn.backend_t.multi_CUDA, lgrngn.backend_t.multi_HIP):
        with pytest.raises(RuntimeError):
            lgrngn.factory(b, oi)


def test_unavailable_backends_raise():
    from libcloudphxx_amd import lgrngn
    oi = lgrngn.opts_init_t()
    for b in (lgrngn.backend_t.serial, lgrngn.backend_t.OpenMP):
        with pytest.raises(RuntimeError):
            lgrngn.factory(b, oi)
