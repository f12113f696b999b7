"""GPU tests of the 1-D decomposition primitives (migrant lists, pack, unpack, finish): a ring of slabs driven
in one process on one GPU, compared slab by slab with the oracle running the identical protocol."""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu


def ring_pair(oi, size, fields):
    th, rv, rhod, C = fields
    rings = []
    for make, alloc in ((h.oracle_particles, h.host_alloc), (h.hip_particles, h.dev_alloc)):
        rings.append(h.LocalRing(oi, size, make, alloc))
    orc, hip = rings
    # same random stream for both: replay the oracle's init draws slab by slab
    for po, ph in zip(orc.prts, hip.prts):
        for arr in h.oracle_rng_preview(po, h.init_replay_calls(po.opts_init)):
            ph.rng_replay_push(0, arr)
    for ring in rings:
        ring.init(th.copy(), rv.copy(), rhod.copy(), **C)
    for po, ph in zip(orc.prts, hip.prts):
        h.copy_state(po, ph)
    return orc, hip


@pytest.mark.parametrize("dims,size", [((6, 0, 5), 2), ((7, 3, 4), 3)])
def test_migration_matches_oracle(dims, size):
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False)
    oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
    fields = h.box_fields(oi)
    orc, hip = ring_pair(oi, size, fields)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    moved = 0
    for it in range(4):
        a = [x.copy() for x in (th, rv, rhod)]
        b = [x.copy() for x in (th, rv, rhod)]
        orc.step(opts, *a, **C)
        hip.step(opts, *b, **C)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            for a_ in ("x", "y", "z", "rw2", "rd3", "kappa"):
                if a_ in ("x", "y", "z") and not getattr(oi, "n" + a_):
                    continue
                np.testing.assert_allclose(ph.get_attr(a_), po.get_attr(a_), rtol=1e-14, atol=1e-9, err_msg="%s slab %d" % (a_, r))
    tot0 = 24 * max(nx, 1) * max(ny, 1) * nz
    assert sum(p.n_part for p in hip.prts) <= tot0


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 2), ((9, 3, 4), 3)])
def test_pred_corr_on_a_ring_matches_oracle(dims, size):
    """pred_corr advection on a decomposed domain: Courant halos (2 x-planes per side) exchanged between the slabs
    (xchng_courants.ipp:15-160) -- device ring against the oracle ring, every slab, positions bit for bit"""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, adve_scheme=lgrngn.as_t.pred_corr)
    oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
    fields = h.box_fields(oi)
    orc, hip = ring_pair(oi, size, fields)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    for it in range(4):
        a = [x.copy() for x in (th, rv, rhod)]
        b = [x.copy() for x in (th, rv, rhod)]
        orc.step(opts, *a, **C)
        hip.step(opts, *b, **C)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            for a_ in ("x", "y", "z"):
                if getattr(oi, "n" + a_):
                    np.testing.assert_allclose(ph.get_attr(a_), po.get_attr(a_), rtol=1e-14, atol=1e-9, err_msg="%s slab %d" % (a_, r))
            np.testing.assert_array_equal(ph.state_real("courant_x"), po.state_real("courant_x"))


def test_migration_carries_perparticle_state():
    """exact_sstp_cond + sstp_cond_act: the private (rv, th, rhod) of a droplet and rc2 are attributes that are packed,
    unpacked and compacted with it (particles_impl.ipp:452-491); condensation on, so the carried values matter"""
    nx, ny, nz, size = 6, 0, 5, 2
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, sstp_cond=2, exact_sstp_cond=True, sstp_cond_mix=False,
                    adaptive_sstp_cond=True, sstp_cond_act=4)
    oi.n_sd_max = 24 * nx * nz * 3
    fields = h.box_fields(oi)
    orc, hip = ring_pair(oi, size, fields)
    assert hip.prts[0].migrate_record_bytes() == 8 + 8 * (4 + 2 + 4)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = False
    for it in range(3):
        a = [x.copy() for x in (th, rv, rhod)]
        b = [x.copy() for x in (th, rv, rhod)]
        orc.step(opts, *a, **C)
        hip.step(opts, *b, **C)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-7)
        np.testing.assert_allclose(b[1], a[1], rtol=1e-7)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            for nm in ("sstp_tmp_rv", "sstp_tmp_th", "sstp_tmp_rh"):
                np.testing.assert_allclose(ph.state_real(nm), po.state_real(nm), rtol=1e-6, err_msg="%s slab %d" % (nm, r))
            np.testing.assert_allclose(ph.state_real("rc2"), po.state_real("rc2"), rtol=1e-5)
            np.testing.assert_allclose(ph.get_attr("rw2"), po.get_attr("rw2"), rtol=2e-4)


def test_ring_round_trip_bit_identical_hip():
    """tests/mpi/mpi_adve_test.cpp:196-255 on the GPU: nx steps with C = 1 bring every SD back to its cell"""
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.coal_switch = oi.sedi_switch = False
    oi.dt = 1
    oi.nx, oi.nz, oi.dx, oi.dz = 7, 4, 1, 1
    oi.x1, oi.z1 = 7., 4.
    oi.sd_conc = 8
    oi.n_sd_max = 8 * 7 * 4 * 3
    oi.adve_scheme = lgrngn.as_t.euler
    ring = h.LocalRing(oi, 3, h.hip_particles, h.dev_alloc)
    th, rv, rhod = 300. * np.ones((7, 4)), .01 * np.ones((7, 4)), np.ones((7, 4))
    Cx, Cz = np.ones((8, 4)), np.zeros((7, 5))
    ring.init(th, rv, rhod, Cx=Cx, Cz=Cz)
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = opts.sedi = False

    def diags():
        out = []
        for fn, k in (("diag_sd_conc", None), ("diag_dry_mom", 1), ("diag_wet_mom", 1), ("diag_kappa_mom", 1)):
            def one(p):
                p.diag_all()
                getattr(p, fn)(*([k] if k is not None else []))
                return p.outbuf_array()
            out.append(ring.gather(one))
        return np.stack(out)
    before = diags()
    for step in range(oi.nx):
        ring.step(opts, th, rv, rhod, Cx=Cx, Cz=Cz)
    after = diags()
    assert np.array_equal(before, after)


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 2), ((9, 3, 4), 3)])
def test_immigrants_reuse_the_slots_of_emigrants(dims, size):
    """Production order (opts_init.reorder_every >= 0, no replayed stream): immigrants are written into the storage slots that the
    emigrants of the same step have vacated, so a slab whose inflow balances its outflow keeps its storage extent.  Same
    super-droplets as with the reference's append-and-compact order (reorder_every = -1), slab by slab, as multisets."""
    nx, ny, nz = dims
    runs = []
    for every in (-1, 0):
        oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, reorder_every=every)
        oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
        th, rv, rhod, C = h.box_fields(oi)
        ring = h.LocalRing(oi, size, h.hip_particles, h.dev_alloc)
        ring.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = opts.cond = False
        for _ in range(6):
            ring.step(opts, th.copy(), rv.copy(), rhod, **C)
        per_slab = []
        for p in ring.prts:
            key = np.lexsort((p.get_attr("z"), p.get_attr("x"), p.get_attr("rd3")))
            per_slab.append((p.n_part, {a: p.get_attr(a)[key] for a in ("rd3", "rw2", "x", "z")}, p.state_u64("n")[key],
                             len(p.state_u64("ijk"))))
        runs.append(per_slab)
    moved = False
    for (na, attrs_a, mult_a, _), (nb, attrs_b, mult_b, _) in zip(*runs):
        assert na == nb
        assert np.array_equal(mult_a, mult_b)
        for k in attrs_a:
            assert np.array_equal(attrs_a[k], attrs_b[k]), k
    # the diagnostics see the same cells either way
    def conc(ring_prts):
        out = []
        for p in ring_prts:
            p.diag_all(); p.diag_sd_conc(); out.append(p.outbuf_array().copy())
        return np.concatenate(out)


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 2), ((9, 3, 4), 3)])
def test_open_side_walls_on_a_decomposed_domain(dims, size):
    """opts_init.open_side_walls with slabs: the end ranks have no neighbour beyond the wall, what crosses it is removed
    (bcond open, bcnd.ipp:160-205); interior faces exchange as usual.  Slab by slab against the oracle running the same protocol."""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, open_side_walls=True)
    oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
    fields = h.box_fields(oi)
    orc, hip = ring_pair(oi, size, fields)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    n0 = sum(p.n_part for p in orc.prts)
    for it in range(5):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        hip.step(opts, th.copy(), rv.copy(), rhod, **C)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            np.testing.assert_allclose(ph.get_attr("x"), po.get_attr("x"), rtol=1e-14, atol=1e-9)
    assert sum(p.n_part for p in hip.prts) < n0            # the walls did swallow super-droplets


def test_precipitating_super_droplets_that_also_cross_a_slab_face():
    """A super-droplet that falls out of the bottom (or leaves through the top) in the very pass that takes it across a slab
    face dies on the slab it left: it is neither shipped nor counted.  Both engines must agree on n_part, on the sorted order
    and on the puddle, slab by slab, and the device's CSR end must equal its n_part (a stale slot there once let k_cellrank
    run over garbage)."""
    nx, ny, nz, size = 8, 0, 4, 2
    oi = h.box_opts(nx, ny, nz, 32, dx=20., coal_switch=False)
    oi.n_sd_max = 32 * nx * nz * 3
    th, rv, rhod, C = h.box_fields(oi, supersat=False)
    C["Cx"] = 0.9 * np.ones_like(C["Cx"])                 # nine in ten of the SDs of an edge column change slabs
    C["Cz"] = np.zeros_like(C["Cz"])
    orc, hip = ring_pair(oi, size, (th, rv, rhod, C))
    # rain drops just above the floor: r = 1 mm falls ~6.5 m per step, z in [0, 3) m
    for po, ph in zip(orc.prts, hip.prts):
        g = lambda nm: po.state_real(nm)
        z = g("z").copy(); rw2 = g("rw2").copy()
        low = z < oi.dz                                    # the bottom row of cells
        z[low] = 3. * (z[low] / oi.dz)
        rw2[low] = 1e-6
        for p in (po, ph):
            p.set_particles(po.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), None, z)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    n0 = sum(p.n_part for p in orc.prts)
    for it in range(3):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        hip.step(opts, th.copy(), rv.copy(), rhod, **C)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            assert int(ph.state_u64("cell_start")[-1]) == ph.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            pud_o, pud_h = po.diag_puddle(), ph.diag_puddle()
            for k in pud_o:
                np.testing.assert_allclose(pud_h[k], pud_o[k], rtol=1e-12, err_msg="%s slab %d" % (k, r))
    assert sum(p.n_part for p in hip.prts) < n0            # it did rain
