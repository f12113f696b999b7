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


@pytest.mark.parametrize("flavour", ["double", "float"])
def test_oracle_exports_the_same_surface(flavour):
    from _harness import oracle_lib, oracle_f32_lib
    lib = oracle_lib() if flavour == "double" else oracle_f32_lib()       # (round 5: the same source with real = float)
    # the oracle mirrors the ABI (prefix orc_) for everything the tests drive through the shared harness
    skip = {"lcx_dev_alloc", "lcx_dev_free", "lcx_dev_copy", "lcx_dev_sync", "lcx_timings", "lcx_set_profiling",
            "lcx_rng_dump", "lcx_math_probe",                      # (lcx_rng_dump: the device generator's stream; the oracle CONSUMES it, orc_rng_replay_push)
            "lcx_create_multi", "lcx_multi_dev_count", "lcx_multi_slab", "lcx_philox_probe"}     # (the oracle's ring is LocalRing in tests/_harness.py)
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


def test_api_default_arithmetic_is_the_same_in_every_mirror():
    """Round 5: what a caller that sets nothing gets is fast arithmetic with the reference's TOMS748 iterates (strict_fp = 0,
    cond_solver = 1) -- the C ABI's lcx_opts_init_default, the Python mirror's constructor and the C++ header mirror's member
    initialisers say the same (the test suite itself pins the parity mode, tests/_harness.py)"""
    import _harness as h
    from libcloudphxx_amd import lgrngn, _lib
    h.api_default_opts(lgrngn.opts_init_t())
    assert h.API_DEFAULTS == {"strict_fp": False, "cond_solver": 1}
    hdr = open(os.path.join(ROOT, "include", "libcloudph++", "lgrngn", "opts_init.hpp")).read()
    assert re.search(r"bool strict_fp = false;", hdr) and re.search(r"int cond_solver = 1;", hdr)
    c = lgrngn._opts_init_c()
    f = _lib.load().lcx_opts_init_default
    f.restype = None
    f(ctypes.byref(c))
    assert (c.strict_fp, c.cond_solver) == (0, 1)
    # (the oracle is one arithmetic whatever the option says; its default follows the header it shares)
    g = h.oracle_lib().orc_opts_init_default
    g.restype = None
    g(ctypes.byref(c))
    assert (c.strict_fp, c.cond_solver) == (0, 1)


def test_unavailable_backends_raise():
    from libcloudphxx_amd import lgrngn
    oi = lgrngn.opts_init_t()
    for b in (lgrngn.backend_t.serial, lgrngn.backend_t.OpenMP):
        with pytest.raises(RuntimeError):
            lgrngn.factory(b, oi)


# Philox4x32-10 known answers: Random123's kat_vectors (philox4x32 10 <counter x4> <key x2> -> <out x4>), re-derived for this
# test from the round function published in Salmon et al. 2011 (tests/test_abi.py::philox_reference below reproduces them)
PHILOX_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox_reference(ctr, key, rounds=10):
    """Philox4x32-R as published (multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85)"""
    c, k = list(ctr), list(key)
    for _ in range(rounds):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return tuple(c)


def philox_probe(triples, on_device):
    import numpy as np
    from libcloudphxx_amd import _lib
    lib = _lib.load()
    ics = np.ascontiguousarray(triples, dtype=np.uint64).reshape(-1)
    out = np.zeros(4 * (len(ics) // 3), dtype=np.uint32)
    rc = lib.lcx_philox_probe(ics.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(ics) // 3), out.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_int(int(on_device)))
    assert rc == 0
    return out.reshape(-1, 4)


def kat_triples():
    return [(c[0] | (c[1] << 32), c[2] | (c[3] << 32), k[0] | (k[1] << 32)) for c, k, _ in PHILOX_KAT]


def test_philox_known_answers_host():
    """the library's generator IS Philox4x32-10: Random123's known-answer vectors through the host build of the very routine the
    kernels inline (csrc/lcx_math.hpp philox::gen)"""
    for c, k, want in PHILOX_KAT:
        assert philox_reference(c, k) == want
    got = philox_probe(kat_triples(), on_device=False)
    for row, (_, _, want) in zip(got, PHILOX_KAT):
        assert tuple(int(x) for x in row) == want


@pytest.mark.gpu
def test_philox_known_answers_device():
    """the same on the GPU, plus 1e5 random (index, call, seed) triples against the reference round function"""
    import numpy as np
    got = philox_probe(kat_triples(), on_device=True)
    for row, (_, _, want) in zip(got, PHILOX_KAT):
        assert tuple(int(x) for x in row) == want
    rng = np.random.default_rng(3)
    tri = rng.integers(0, 2 ** 63, size=(100000, 3), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(100000, 3), dtype=np.uint64)
    dev = philox_probe(tri, on_device=True)
    host = philox_probe(tri, on_device=False)
    assert np.array_equal(dev, host)
    for t, row in list(zip(tri, dev))[:200]:
        i, c, s = (int(x) for x in t)
        want = philox_reference((i & 0xffffffff, i >> 32, c & 0xffffffff, c >> 32), (s & 0xffffffff, s >> 32))
        assert tuple(int(x) for x in row) == want
