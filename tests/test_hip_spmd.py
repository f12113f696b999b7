"""The one-process-per-GPU path with the HIP ENGINE: libcloudphxx_amd.multi.particles_multi_t (lcx_exch_* of the C ABI: emigrant
counts stay on the device, one message per direction whose size is agreed one step ahead, one host synchronisation per step) run by
two and three ranks under torch.distributed.  The test box has one GPU, so the ranks share device 0 and the messages are staged
through the host (gloo); on a node with one GPU per rank the same calls hand the device buffers to RCCL.

Checkers: the ring round trip of the reference's MPI test (tests/mpi/mpi_adve_test.cpp:196-255) bit-identical; the slabs of the
native multi-device object (factory(multi_CUDA), itself held to the oracle ring in tests/test_hip_multi.py) bit-identical."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def wait_all(procs, limit_s):
    """exit codes of the ranks; a rank that fails (or the time limit) ends the others at once -- they would wait for its messages for ever"""
    import time
    t0 = time.time()
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            return codes
        if any(c not in (None, 0) for c in codes) or time.time() - t0 > limit_s:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            return [p.wait() for p in procs]
        time.sleep(0.05)


def launch(mode, world, res, transport="host"):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_spmd_worker.py"), mode, str(r), str(world), str(port), res, transport])
             for r in range(world)]
    codes = wait_all(procs, 240)
    assert codes == [0] * world, codes


@pytest.mark.parametrize("world", [2, 3])
def test_spmd_hip_engine_ring_round_trip_bit_identical(world, tmp_path):
    res = str(tmp_path / "r%d.npy")
    launch("ring", world, res)
    for r in range(world):
        d = np.load(res % r)
        assert np.array_equal(d[0], d[1]) and d[0][0].sum() > 0


@pytest.mark.parametrize("mode,world", [("steps", 2), ("steps", 4), ("fsteps", 2), ("fsteps", 3), ("uneven", 3)])
def test_spmd_hip_engine_equals_the_native_multi_device_object(mode, world, tmp_path, monkeypatch):
    """three full steps (cond + coal + adve + sedi, Philox streams, production storage order) by `world` ranks = the slabs of ONE
    multi-device object over `world` slabs on the same device, bit for bit: same kernels, same message layout, same unpack order --
    only the transport differs.  "uneven": nx = 7 over 3 ranks (2 + 2 + 3 planes) at Courant number 0.95: the thin slabs send nearly a
    whole plane each step into the inbox of the thick one (one capacity for all slabs; the native object at these sizes is held to the
    oracle ring by tests/test_hip_multi.py::test_multi_device_uneven_slabs_at_courant_one)"""
    import _spmd_worker as w
    res = str(tmp_path / "s%d.npz")
    launch(mode, world, res)
    full = mode in ("steps", "fsteps")        # ("fsteps": fast arithmetic, the slabs' re-sort riding on the next condensation kernel)
    nx, ny, nz = (8, 3, 4) if full else (7, 0, 5)
    oi = w.box(nx, ny, nz, 24, 44, coal_switch=full, strict_fp=(mode != "fsteps"))
    th, rv, rhod, C = h.box_fields(oi)
    if mode == "uneven":
        C["Cx"] = 0.95 * np.ones_like(C["Cx"])
    monkeypatch.setenv("LCX_MULTI_DEVICE_MAP", ",".join(["0"] * world))
    oi.dev_count = world
    mul = lgrngn.factory(lgrngn.backend_t.multi_CUDA, oi)
    mul.init(th, rv, rhod, **C)
    opts = lgrngn.opts_t()
    opts.coal = full
    n0 = mul.n_part
    for it in range(3 if full else 5):
        mul.step_sync(opts, th, rv, rhod, **C)
        mul.step_async(opts)
    per = nx // world
    tot = 0
    for r in range(world):
        d = np.load(res % r)
        s = w.slab_state(mul.slab(r))
        tot += int(d["n_part"][0])
        for k in s:
            assert np.array_equal(d[k], s[k]), (r, k)
        b, n = r * per, (per if r < world - 1 else nx - r * per)
        assert np.array_equal(d["th"], th[b:b + n]) and np.array_equal(d["rv"], rv[b:b + n]), r
    assert tot == mul.n_part and n0 - 4 <= tot <= n0      # (no coalescence in "uneven": at most a droplet or two through the floor)


# ---- the RCCL transport itself.  RCCL refuses two ranks on one device, and the test box has one: ONE rank whose left and right neighbour
# is the rank itself (particles_multi_t(self_ring=True)) sends every message of the protocol to its own inbox through RCCL -- the engine's
# buffers handed to torch.distributed as device tensors, the operations queued on the engine's own stream (ExternalStream), the unpack
# kernels ordered behind them by that stream and nothing else.  What a node with one GPU per rank adds to this is other peers.
def test_rccl_transport_ring_of_one_round_trip_bit_identical(tmp_path):
    res = str(tmp_path / "r%d.npy")
    launch("selfring", 1, res, "rccl")
    d = np.load(res % 0)
    assert np.array_equal(d[0], d[1]) and d[0][0].sum() > 0


def test_rccl_transport_ring_of_one_three_steps_equal_the_periodic_box(tmp_path):
    """condensation + advection + sedimentation, three steps: the super-droplets of the ring of one (every droplet that leaves the slab
    comes back through RCCL) are the super-droplets of the plain periodic single-device run -- as a set: the immigrants take other
    storage slots.  (No coalescence here: its pairing follows the storage order.)"""
    import _spmd_worker as w
    res = str(tmp_path / "s%d.npz")
    launch("selfsteps", 1, res, "rccl")
    d = np.load(res % 0)
    assert d["bytes_moved"][0] > 0
    oi = w.box(8, 3, 4, 24, 44, coal_switch=True)
    th, rv, rhod, C = h.box_fields(oi)
    prt = lgrngn.particles_t(oi, np.float64)
    prt.init(th, rv, rhod, **C)
    opts = lgrngn.opts_t()
    opts.coal = False
    for it in range(3):
        prt.step_sync(opts, th, rv, rhod, **C)
        prt.step_async(opts)
    assert int(d["n_part"][0]) == prt.n_part
    key = lambda rd3, rw2, x, z: np.lexsort((z, x, rw2, rd3))
    a = {k: d[k] for k in ("rd3", "rw2", "x", "z")}
    b = {k: prt.get_attr(k) for k in ("rd3", "rw2", "x", "z")}
    ia, ib = key(**a), key(**b)
    assert np.array_equal(a["rd3"][ia], b["rd3"][ib])
    assert np.array_equal(d["n"][ia], prt.state_u64("n")[ib])
    np.testing.assert_allclose(a["rw2"][ia], b["rw2"][ib], rtol=1e-12, atol=0)
    np.testing.assert_allclose(a["z"][ia], b["z"][ib], rtol=0, atol=1e-9)
    # x: the periodic box wraps with fmod, the ring re-bases by the neighbour's edge -- the same number up to the rounding of one subtraction
    dx = np.abs(a["x"][ia] - b["x"][ib])
    assert np.all(np.minimum(dx, oi.x1 - dx) < 1e-9)
    np.testing.assert_allclose(d["th"], th, rtol=1e-12)
    np.testing.assert_allclose(d["rv"], rv, rtol=1e-12)
