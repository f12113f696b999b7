"""GPU tests of the native multi-device object, factory(multi_HIP | multi_CUDA) = lcx_create_multi (libcloudphxx_amd/csrc/lcx_multi.hpp):
ONE object, N slabs, worker threads, the device-driven neighbour exchange (pack straight into the neighbour's inbox, counts in the
message header, one host synchronisation per step).  The GPU box has one device, so the slabs are all mapped to device 0
(LCX_MULTI_DEVICE_MAP) -- the same code path as on N devices except that the "peer" writes stay on the device.

Checker: the oracle running the reference's protocol (pack -> unpack -> post_copy, tests/_harness.py LocalRing) slab by slab."""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu


def make_multi(oi, size, monkeypatch):
    monkeypatch.setenv("LCX_MULTI_DEVICE_MAP", ",".join(["0"] * size))
    oi.dev_count = size
    return lgrngn.factory(lgrngn.backend_t.multi_CUDA, oi)


def multi_pair(oi, size, fields, monkeypatch):
    """oracle ring + multi-device object, same initial state (the oracle's random draws replayed slab by slab)"""
    th, rv, rhod, C = fields
    orc = h.LocalRing(oi, size, h.oracle_particles, h.host_alloc)
    mul = make_multi(oi, size, monkeypatch)
    assert mul.dev_count == size
    slabs = [mul.slab(r) for r in range(size)]
    for po, ph in zip(orc.prts, slabs):
        for arr in h.oracle_rng_preview(po, h.init_replay_calls(po.opts_init)):
            ph.rng_replay_push(0, arr)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    mul.init(th.copy(), rv.copy(), rhod.copy(), **C)
    for po, ph in zip(orc.prts, slabs):
        ph.opts_init = po.opts_init
        h.copy_state(po, ph)
    return orc, mul, slabs


def compare_slabs(orc, slabs, oi, tag, attrs=("x", "y", "z", "rw2", "rd3", "kappa"), rtol=1e-14):
    for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
        assert ph.n_part == po.n_part, (tag, r)
        for nm in ("n", "ijk", "sorted_id"):
            assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (tag, r, nm)
        for a_ in attrs:
            if a_ in ("x", "y", "z") and not getattr(oi, "n" + a_):
                continue
            np.testing.assert_allclose(ph.get_attr(a_), po.get_attr(a_), rtol=rtol, atol=1e-9, err_msg="%s slab %d" % (a_, r))


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 4), ((9, 3, 4), 3), ((6, 0, 5), 2)])
def test_multi_device_object_matches_oracle_ring(dims, size, monkeypatch):
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False)
    oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
    fields = h.box_fields(oi)
    orc, mul, slabs = multi_pair(oi, size, fields, monkeypatch)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = False
    tho, rvo, thm, rvm = th.copy(), rv.copy(), th.copy(), rv.copy()
    for it in range(4):
        orc.step(opts, tho, rvo, rhod, **C)
        mul.step_sync(opts, thm, rvm, rhod, **C)
        mul.step_async(opts)
        compare_slabs(orc, slabs, oi, it, attrs=("x", "y", "z", "rd3", "kappa"))
        for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
            np.testing.assert_allclose(ph.get_attr("rw2"), po.get_attr("rw2"), rtol=2e-4)
        np.testing.assert_allclose(thm, tho, rtol=1e-7)          # each slab wrote its planes of the global arrays
        np.testing.assert_allclose(rvm, rvo, rtol=1e-7)
    assert mul.n_part == sum(p.n_part for p in orc.prts)
    # outbuf() is the global field (particles_multi_gpu_diag.ipp:274-307)
    mul.diag_all(); mul.diag_sd_conc()
    got = mul.outbuf_array()

    def one(p):
        p.diag_all(); p.diag_sd_conc()
        return p.outbuf_array()
    assert np.array_equal(got, orc.gather(one))


@pytest.mark.parametrize("dims,size,cx", [((7, 0, 5), 3, 0.95), ((15, 0, 4), 8, 0.95), ((11, 2, 3), 4, -0.9)])
def test_multi_device_uneven_slabs_at_courant_one(dims, size, cx, monkeypatch):
    """nx not divisible by the number of slabs: the last slab takes the remainder (distmem_opts.hpp:10-16: 15 planes over 8 slabs =
    seven of 1 plane and one of 8) and the thin slabs send nearly a whole plane per step into its inbox.  Round 2 sized every inbox
    from the slab's OWN plane count while the senders checked against theirs: the thick slab's inbox was the smallest.  One
    capacity for all slabs now (Particles::exch_capacity); against the oracle ring, six steps"""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False)
    oi.n_sd_max = 24 * nx * max(ny, 1) * nz * size          # (every slab gets n_sd_max / size + 1: room for the thick one)
    th, rv, rhod, C = h.box_fields(oi)
    C["Cx"] = cx * np.ones_like(C["Cx"])
    orc, mul, slabs = multi_pair(oi, size, (th, rv, rhod, C), monkeypatch)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    for it in range(6):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_async(opts)
        compare_slabs(orc, slabs, oi, it, attrs=("x", "y", "z"))
    assert mul.n_part == sum(p.n_part for p in orc.prts)


def test_multi_device_crowded_cells_in_the_overlapped_resort(monkeypatch):
    """cells above k_cellrank's limit (300 super-droplets per cell: listed and sorted by one wave each) on slabs with neighbours: the
    interior's crowded cells are listed before the messages arrive, the boundary planes' behind the unpack, all of them sorted once
    the host knows the counts -- plain and shuffled order (coalescence on, replayed streams) against the oracle ring"""
    nx, ny, nz, size = 8, 0, 3, 2
    oi = h.box_opts(nx, ny, nz, 300, dx=20.)
    oi.n_sd_max = 300 * nx * nz * 3
    fields = h.box_fields(oi)
    orc, mul, slabs = multi_pair(oi, size, fields, monkeypatch)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.cond = False
    for it in range(3):
        for po, ph in zip(orc.prts, slabs):
            h.push_coal_replay(po, ph)
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_async(opts)
        compare_slabs(orc, slabs, oi, it, attrs=("x", "z", "rw2", "rd3"))
        for ph in slabs:
            cnt = np.diff(ph.state_u64("cell_start").astype(np.int64))
            assert cnt.max() > 256


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 2), ((9, 3, 4), 3), ((16, 0, 5), 2)])
def test_multi_device_pred_corr_and_open_walls(dims, size, monkeypatch):
    """pred_corr (Courant halo of two planes per side; (16, 0, 5) over 2 slabs: 8 planes each, so that the overlapped re-sort runs with
    its two boundary planes per side) and open side walls through the multi-device object"""
    nx, ny, nz = dims
    for kw in (dict(adve_scheme=lgrngn.as_t.pred_corr), dict(open_side_walls=True)):
        oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, **kw)
        oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
        fields = h.box_fields(oi)
        orc, mul, slabs = multi_pair(oi, size, fields, monkeypatch)
        th, rv, rhod, C = fields
        opts = lgrngn.opts_t()
        opts.coal = opts.cond = False
        n0 = mul.n_part
        for it in range(4):
            orc.step(opts, th.copy(), rv.copy(), rhod, **C)
            mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
            mul.step_async(opts)
            compare_slabs(orc, slabs, oi, (tuple(kw), it), attrs=("x", "y", "z"))
        if "open_side_walls" in kw:
            assert mul.n_part < n0


def test_multi_device_ring_round_trip_bit_identical(monkeypatch):
    """tests/mpi/mpi_adve_test.cpp:196-255 through factory(multi_CUDA): nx steps with C = 1 bring every SD back to its cell"""
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.coal_switch = oi.sedi_switch = False
    oi.dt = 1
    oi.nx, oi.nz, oi.dx, oi.dz = 8, 4, 1, 1
    oi.x1, oi.z1 = 8., 4.
    oi.sd_conc = 8
    oi.n_sd_max = 8 * 8 * 4 * 4
    oi.adve_scheme = lgrngn.as_t.euler
    mul = make_multi(oi, 4, monkeypatch)
    th, rv, rhod = 300. * np.ones((8, 4)), .01 * np.ones((8, 4)), np.ones((8, 4))
    Cx, Cz = np.ones((9, 4)), np.zeros((8, 5))
    mul.init(th, rv, rhod, Cx=Cx, Cz=Cz)
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = opts.sedi = False

    def diags():
        out = []
        for fn, k in (("diag_sd_conc", None), ("diag_dry_mom", 1), ("diag_wet_mom", 1), ("diag_kappa_mom", 1)):
            mul.diag_all()
            getattr(mul, fn)(*([k] if k is not None else []))
            out.append(mul.outbuf_array())
        return np.stack(out)
    before = diags()
    assert before[0].sum() == 8 * 8 * 4
    for step in range(oi.nx):
        mul.step_sync(opts, th, rv, rhod, Cx, None, Cz)
        mul.step_async(opts)
    after = diags()
    assert np.array_equal(before, after)
    # what left every slab came back: each slab sent nz * sd_conc * (planes that crossed a face) super-droplets
    assert mul.n_part == 8 * 8 * 4


def test_multi_device_precipitation_while_crossing(monkeypatch):
    """rain that leaves through the floor in the pass that takes it across a slab face: counted once, on the slab it left"""
    nx, ny, nz, size = 8, 0, 4, 2
    oi = h.box_opts(nx, ny, nz, 32, dx=20., coal_switch=False)
    oi.n_sd_max = 32 * nx * nz * 3
    th, rv, rhod, C = h.box_fields(oi, supersat=False)
    C["Cx"] = 0.9 * np.ones_like(C["Cx"])
    C["Cz"] = np.zeros_like(C["Cz"])
    orc, mul, slabs = multi_pair(oi, size, (th, rv, rhod, C), monkeypatch)
    for po, ph in zip(orc.prts, slabs):
        g = lambda nm: po.state_real(nm)
        z = g("z").copy(); rw2 = g("rw2").copy()
        low = z < oi.dz
        z[low] = 3. * (z[low] / oi.dz)
        rw2[low] = 1e-6
        for p in (po, ph):
            p.set_particles(po.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), None, z)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    n0 = mul.n_part
    for it in range(3):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_async(opts)
        compare_slabs(orc, slabs, oi, it, attrs=("x", "z"))
        for ph in slabs:
            assert int(ph.state_u64("cell_start")[-1]) == ph.n_part
    pud = mul.diag_puddle()
    tot = {k: sum(p.diag_puddle()[k] for p in orc.prts) for k in pud}
    for k in pud:
        np.testing.assert_allclose(pud[k], tot[k], rtol=1e-12, err_msg=k)
    assert mul.n_part < n0 and pud["liquid_volume"] > 0


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 4), ((9, 3, 4), 3)])
def test_multi_device_production_order_same_droplets(dims, size, monkeypatch):
    """production mode (no replayed stream: immigrants take over the storage slots of the emigrants, storage re-ordered now and
    then) holds the same super-droplets, slab by slab, as the slabs stepped one by one with the host-driven protocol in the
    reference's storage order"""
    nx, ny, nz = dims
    runs = []
    for native in (False, True):
        oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, reorder_every=0 if native else -1)
        oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
        th, rv, rhod, C = h.box_fields(oi)
        opts = lgrngn.opts_t()
        opts.coal = opts.cond = False
        if native:
            obj = make_multi(oi, size, monkeypatch)
            obj.init(th, rv, rhod, **C)
            for _ in range(6):
                obj.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
                obj.step_async(opts)
            prts = [obj.slab(r) for r in range(size)]
        else:
            obj = h.LocalRing(oi, size, h.hip_particles, h.dev_alloc)
            obj.init(th, rv, rhod, **C)
            for _ in range(6):
                obj.step(opts, th.copy(), rv.copy(), rhod, **C)
            prts = obj.prts
        per_slab = []
        for p in prts:
            key = np.lexsort((p.get_attr("z"), p.get_attr("x"), p.get_attr("rd3")))
            per_slab.append((p.n_part, {a: p.get_attr(a)[key] for a in ("rd3", "rw2", "x", "z")}, p.state_u64("n")[key]))
        runs.append(per_slab)
        del obj
    for (na, attrs_a, mult_a), (nb, attrs_b, mult_b) in zip(*runs):
        assert na == nb
        assert np.array_equal(mult_a, mult_b)
        for k in attrs_a:
            assert np.array_equal(attrs_a[k], attrs_b[k]), k


@pytest.mark.parametrize("dims,size", [((8, 0, 5), 4), ((9, 3, 4), 3)])
def test_multi_device_production_order_against_the_oracle_ring(dims, size, monkeypatch):
    """the multi-device object in PRODUCTION mode (immigrants take over the emigrants' slots, storage re-ordered, the overlapped
    re-sort) against the ORACLE ring stepped in the reference's order: no coalescence, so no random number and no replayed stream;
    slab by slab the same droplets (matched by dry radius), multiplicities exact, positions to 1e-13"""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False, reorder_every=2)
    oi.n_sd_max = 24 * max(nx, 1) * max(ny, 1) * nz * 3
    th, rv, rhod, C = h.box_fields(oi)
    orc = h.LocalRing(oi, size, h.oracle_particles, h.host_alloc)
    mul = make_multi(oi, size, monkeypatch)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    mul.init(th.copy(), rv.copy(), rhod.copy(), **C)
    slabs = [mul.slab(r) for r in range(size)]
    for po, ph in zip(orc.prts, slabs):
        ph.opts_init = po.opts_init
        h.copy_state(po, ph)                                   # (set_particles: no replayed stream, the device keeps its production rules)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    for it in range(7):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_async(opts)
    for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
        assert ph.n_part == po.n_part, r
        ko = np.lexsort((po.get_attr("z"), po.get_attr("x"), po.get_attr("rd3")))
        kh = np.lexsort((ph.get_attr("z"), ph.get_attr("x"), ph.get_attr("rd3")))
        assert np.array_equal(po.get_attr("rd3")[ko], ph.get_attr("rd3")[kh]), r
        assert np.array_equal(po.state_u64("n")[ko], ph.state_u64("n")[kh]), r
        for a_ in ("x", "y", "z"):
            if getattr(oi, "n" + a_):
                np.testing.assert_allclose(ph.get_attr(a_)[kh], po.get_attr(a_)[ko], rtol=1e-13, atol=1e-9, err_msg="%s slab %d" % (a_, r))


def test_multi_device_slab_arrays_on_the_device(monkeypatch):
    """on_device == 2: every field as one device array per slab (a host model decomposed the same way) -- same result as the
    global host arrays"""
    import torch
    nx, ny, nz, size = 8, 3, 4, 4
    res = []
    for dev_arrays in (False, True):
        oi = h.box_opts(nx, ny, nz, 16, dx=20., coal_switch=False)
        oi.n_sd_max = 16 * nx * ny * nz * 3
        th, rv, rhod, C = h.box_fields(oi)
        mul = make_multi(oi, size, monkeypatch)
        opts = lgrngn.opts_t()
        opts.coal = False
        per = nx // size
        keep = []

        def split(a, ext=0):
            if a is None:
                return None
            parts = [torch.from_numpy(np.ascontiguousarray(a[r * per:(r + 1) * per + ext])).cuda() for r in range(size)]
            keep.append(parts)
            return lgrngn.DeviceArrays([p.data_ptr() for p in parts], parts[0].shape), parts
        if dev_arrays:
            (tha, thp), (rva, rvp), (rha, _) = split(th), split(rv), split(rhod)
            (cxa, _), (cya, _), (cza, _) = split(C["Cx"], 1), split(C["Cy"]), split(C["Cz"])
            torch.cuda.synchronize()
            mul.init(tha, rva, rha, Cx=cxa, Cy=cya, Cz=cza)
            for _ in range(3):
                mul.step_sync(opts, tha, rva, rha, cxa, cya, cza)
                mul.step_async(opts)
            torch.cuda.synchronize()
            tho = np.concatenate([p.cpu().numpy() for p in thp])
            rvo = np.concatenate([p.cpu().numpy() for p in rvp])
        else:
            tho, rvo = th.copy(), rv.copy()
            mul.init(tho, rvo, rhod, **C)
            for _ in range(3):
                mul.step_sync(opts, tho, rvo, rhod, **C)
                mul.step_async(opts)
        mul.diag_all(); mul.diag_wet_mom(3)
        res.append((tho, rvo, mul.outbuf_array(), mul.n_part))
        del mul
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_multi_device_api_rules(monkeypatch):
    """the reference's multi_CUDA rules (particles_multi_gpu_impl.ipp:46-82, particles_multi_gpu_ctor.ipp:53-57,
    particles_multi_gpu_step.ipp:63-65)"""
    oi = h.box_opts(4, 0, 3, 8, dx=20., coal_switch=False)
    mul = make_multi(oi, 2, monkeypatch)
    th, rv, rhod, C = h.box_fields(oi)
    mul.init(th, rv, rhod, **C)
    with pytest.raises(RuntimeError, match="get_attr"):
        mul.get_attr("rw2")
    opts = lgrngn.opts_t()
    opts.rcyc = True
    mul.step_sync(opts, th, rv, rhod, **C)
    with pytest.raises(RuntimeError, match="recycling"):
        mul.step_async(opts)
    oi2 = h.box_opts(2, 0, 3, 8, dx=20., coal_switch=False)
    with pytest.raises(RuntimeError, match="greater than nx"):
        make_multi(oi2, 3, monkeypatch)
    oi0 = h.box_opts(0, 0, 0, 8)
    with pytest.raises(RuntimeError, match="0D"):
        make_multi(oi0, 2, monkeypatch)
    monkeypatch.delenv("LCX_MULTI_DEVICE_MAP")
    oi3 = h.box_opts(64, 0, 3, 8, dx=20., coal_switch=False)
    oi3.dev_count = 64
    with pytest.raises(RuntimeError, match="number of available GPUs"):
        lgrngn.factory(lgrngn.backend_t.multi_HIP, oi3)


def test_multi_device_without_peer_mapping(monkeypatch):
    """devices that cannot map each other's memory (opts_init.dbg_flags & MULTI_NO_PEER forces it here): emigrants are packed at home and moved by a
    peer copy of exactly the bytes used -- same slabs as the direct path, against the oracle ring"""
    nx, ny, nz, size = 9, 3, 4, 3
    oi = h.box_opts(nx, ny, nz, 24, dx=20., coal_switch=False)
    oi.dbg_flags = int(lgrngn.dbg.MULTI_NO_PEER)
    oi.n_sd_max = 24 * nx * ny * nz * 3
    fields = h.box_fields(oi)
    orc, mul, slabs = multi_pair(oi, size, fields, monkeypatch)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    for it in range(4):
        orc.step(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
        mul.step_async(opts)
        compare_slabs(orc, slabs, oi, it, attrs=("x", "y", "z", "rw2"))


def _run_skewed(monkeypatch, overlap, delay_us, steps=12):
    import bench
    nx, ny, nz, size = 64, 128, 128, 4
    oi = bench.make_opts_init(nx, ny, nz, 64, 40., 1, 1, 44)
    oi.dbg_flags = 0 if overlap else int(lgrngn.dbg.NO_OVERLAP)
    oi.dbg_pack_delay_us = delay_us
    oi.n_sd_max = int(oi.n_sd_max * 1.1)
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(nx, ny, nz, 0, nx, np, np.float64)
    mul = make_multi(oi, size, monkeypatch)
    mul.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz)
    opts = lgrngn.opts_t()
    for _ in range(3):
        mul.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
        mul.step_async(opts)
    mul.set_profiling(True)
    for _ in range(steps):
        mul.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
        mul.step_async(opts)
    t = mul.timings()
    state = [(s.n_part, s.state_u64("n"), s.get_attr("rw2"), s.get_attr("x")) for s in (mul.slab(r) for r in range(size))]
    return {k: v / steps for k, v in t.items() if k != "rendezvous_hidden_share"}, state


def test_overlapped_resort_hides_a_late_neighbour(monkeypatch):
    """north_star: the neighbour exchange overlapped with interior work.  Four slabs of 16 x 128 x 128 cells x 64 super-droplets on
    the one device; every second slab is 300 us late with its messages (a spin kernel ahead of its pack kernel).  Without the overlap a
    slab's stream sits in `exchange_wait` until the late neighbour's message is there and only then scans, scatters and ranks; with
    it the interior's re-sort is queued ahead of the wait and fills the gap.  Same droplets either way, bit for bit."""
    t_ov, s_ov = _run_skewed(monkeypatch, True, 300)
    t_no, s_no = _run_skewed(monkeypatch, False, 300)
    for (na, ma, ra, xa), (nb, mb, rb, xb) in zip(s_ov, s_no):
        assert na == nb and np.array_equal(ma, mb) and np.array_equal(ra, rb) and np.array_equal(xa, xb)
    print("per step, slowest slab: overlap", {k: round(v, 3) for k, v in t_ov.items() if k.startswith(("exchange", "post_copy"))},
          "no overlap", {k: round(v, 3) for k, v in t_no.items() if k.startswith(("exchange", "post_copy"))})
    assert "exchange_sort_interior" in t_ov and "exchange_sort_interior" not in t_no
    # the time a stream spends blocked on its neighbours shrinks by (most of) the interior work that now runs ahead of the wait
    assert t_ov["exchange_wait"] < t_no["exchange_wait"] - 0.5 * min(t_ov["exchange_sort_interior"], 0.3)
