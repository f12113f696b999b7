"""BASELINE.json configs[3] (C4) at FULL size through the native multi-device object: 256 x 256 x 128 cells x 64 super-droplets
(5.4e8 SDs) cut into 8 x-slabs of 32 planes.  The GPU box has one device, so all eight slabs live on it (LCX_MULTI_DEVICE_MAP) --
the object, its worker threads, the device-driven exchange and the per-slab device arrays are exactly what an 8-GPU node runs; only
the peer writes stay on the device.  Checked through size-independent properties (the oracle cannot run 5.4e8 SDs)."""
import numpy as np
import pytest

import bench
import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu
NX, NY, NZ, SD, SLABS = 256, 256, 128, 64, 8


@pytest.fixture(scope="module")
def c4():
    import os
    import torch
    old = os.environ.get("LCX_MULTI_DEVICE_MAP")
    os.environ["LCX_MULTI_DEVICE_MAP"] = ",".join(["0"] * SLABS)
    try:
        oi = bench.make_opts_init(NX, NY, NZ, SD, 40., 1, 1, 44)
        oi.strict_fp = False
        oi.cond_solver = 0             # (bench.py's headline: set here, read back from every slab in test_c4_steps)
        oi.dev_count = SLABS
        oi.n_sd_max = int(oi.n_sd_max * 1.1)
        prt = lgrngn.factory(lgrngn.backend_t.multi_HIP, oi)
        per = NX // SLABS
        dev = torch.device("cuda", 0)

        class TorchXP:
            @staticmethod
            def arange(m, dtype=None):
                return torch.arange(m, dtype=torch.float64, device=dev)
            sin, cos, exp, log = staticmethod(torch.sin), staticmethod(torch.cos), staticmethod(torch.exp), staticmethod(torch.log)
        shapes = [(per, NY, NZ)] * 3 + [(per + 1, NY, NZ), (per, NY + 1, NZ), (per, NY, NZ + 1)]
        parts = [[t.expand(sh).contiguous() for t, sh in zip(bench.make_fields(per, NY, NZ, r * per, NX, TorchXP, torch.float64), shapes)]
                 for r in range(SLABS)]
        arrays = [lgrngn.DeviceArrays([parts[r][k].data_ptr() for r in range(SLABS)], parts[0][k].shape) for k in range(6)]
        torch.cuda.synchronize()
        prt.init(arrays[0], arrays[1], arrays[2], Cx=arrays[3], Cy=arrays[4], Cz=arrays[5])
        yield prt, oi, parts, arrays
    finally:
        if old is None:
            os.environ.pop("LCX_MULTI_DEVICE_MAP", None)
        else:
            os.environ["LCX_MULTI_DEVICE_MAP"] = old


def gathered(parts, k):
    """the global field k from the slabs' device arrays"""
    import torch
    torch.cuda.synchronize()
    return np.concatenate([p[k].cpu().numpy() for p in parts])


def check_slab_sorted(s, oi):
    sid, sijk, ijk, cs = (s.state_u64(nm) for nm in ("sorted_id", "sorted_ijk", "ijk", "cell_start"))
    n = s.n_part
    assert len(sid) == n
    d = np.diff(sijk.astype(np.int64))
    assert d.min() >= 0
    assert np.array_equal(ijk[sid], sijk)
    chk = np.zeros(len(ijk), dtype=bool)
    chk[sid] = True
    assert chk.sum() == n                                     # a permutation of the live SDs
    n_cell = (NX // SLABS) * NY * NZ
    assert np.array_equal(np.diff(cs.astype(np.int64)), np.bincount(sijk.astype(np.int64), minlength=n_cell))


def test_c4_init(c4):
    prt, oi, parts, arrays = c4
    assert prt.dev_count == SLABS
    assert prt.n_part == NX * NY * NZ * SD
    prt.diag_all()
    prt.diag_sd_conc()
    out = prt.outbuf_array()
    assert out.shape[0] == NX * NY * NZ and out.min() == SD and out.max() == SD
    for r in range(SLABS):
        assert prt.slab(r).n_part == NX * NY * NZ * SD // SLABS


def test_c4_steps(c4):
    prt, oi, parts, arrays = c4
    th, rv, rhod, Cx, Cy, Cz = arrays

    def m3():
        prt.diag_all()
        prt.diag_wet_mom(3)
        return prt.outbuf_array().reshape(NX, NY, NZ)
    # --- condensation only: water is conserved cell by cell, in every slab
    rv0, before = gathered(parts, 1), m3()
    o = lgrngn.opts_t()
    o.coal = o.adve = o.sedi = False
    prt.step_sync(o, th, rv, rhod, Cx, Cy, Cz)
    for r in range(SLABS):             # (read back from every slab's object)
        h.assert_mode(prt.slab(r), False, 0, ("lean", "lean_sorted"))
    prt.step_async(o)
    rv1, after = gathered(parts, 1), m3()
    np.testing.assert_allclose(rv1 - rv0, -(after - before) * 4. / 3 * np.pi * 1e3, rtol=1e-8, atol=1e-15)
    assert np.abs(rv1 - rv0).max() > 0
    # --- full steps: super-droplets cross the slab faces; nobody is lost or duplicated
    n0 = prt.n_part
    per_slab0 = [prt.slab(r).n_part for r in range(SLABS)]
    opts = lgrngn.opts_t()
    for _ in range(3):
        prt.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
        prt.step_async(opts)
    n1 = prt.n_part
    assert n1 <= n0 and n1 > 0.99 * n0                        # coalescence and precipitation only remove a few
    prt.diag_all()
    prt.diag_sd_conc()
    assert prt.outbuf_array().sum() == n1 == sum(prt.slab(r).n_part for r in range(SLABS))
    assert any(prt.slab(r).n_part != per_slab0[r] for r in range(SLABS))
    pud = prt.diag_puddle()
    assert pud["particle_number"] >= 0
    dx = oi.dx
    for r in (0, SLABS - 1):
        s = prt.slab(r)
        check_slab_sorted(s, oi)
        x, y, z = s.get_attr("x"), s.get_attr("y"), s.get_attr("z")
        assert x.min() >= 0 and x.max() < (NX // SLABS) * dx      # slab-local coordinates
        assert y.min() >= 0 and y.max() < NY * dx and z.min() >= 0 and z.max() < NZ * dx
