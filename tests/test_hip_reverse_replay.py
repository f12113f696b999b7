"""Reverse replay: the ORACLE run on the DEVICE's random stream, the device on its production path.

Every other exact coalescence test feeds the oracle's random arrays into the device (lcx_rng_replay_push), which also switches the
device to the reference's storage order.  Here nothing is pushed into the device: it keeps its production path -- Philox uniforms
drawn inside k_coal<T, false, true>, the salted-bijection shuffle keys, the re-sort at the end of step_async pre-shuffled for the next
coalescence and left to the storage-order condensation kernel (deferred scatter), lazy compaction, storage re-ordering -- and
lcx_rng_dump (include/lcx.h) hands what each coalescence call consumed to the oracle (orc_rng_replay_push): the shuffle key un[id] of
every super-droplet (hskpng_sort.ipp:28-47) and the uniform u01[p] of every candidate pair (coal.ipp:369-450).  Super-droplets are
matched by a persistent tag (LCX_DBG_TAG): the device renumbers its ids when it re-orders its storage, the oracle compacts eagerly.

With condensation in the step, the production kernel (k_cond_lean) returns the root of the backward-Euler equation and the reference the
midpoint of TOMS748's last bracket (tests/_harness.py, cond_bars): the wet radii agree to the root finder's tolerance, not to the
last bit, and a free run would let that difference wander into the terminal velocities and positions.  So each step compares the
condensation at its own bars (rw2 1e-4, th and rv at cond_bars) and then RE-BASES the oracle on the device's wet radii, th and rv
(orc_set_state_real): coalescence, sedimentation, advection and the re-sort are compared from identical inputs at their own tight bars
(n, cells and tags exact, rd3 1e-14, rw2 1e-13, positions 1e-13).  Without condensation nothing is re-based: a free run of 12 steps.

The box collides: the exponential-in-volume spectrum of the reference's Golovin test (tests/python/physics/coalescence_golovin.py:
31-44,68-74: r_zero = 30.084 um, n_zero = 2^23, kappa = 1e-10) under the hall_davis_no_waals kernel, drizzle falling out of the bottom.
"""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu

DEAD = 0xFFFFFFFF


def golovin_spectrum(lnr, r_zero=30.084e-6, n_zero=2 ** 23):
    r = np.exp(lnr)
    return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(-np.power((r / r_zero), 3))


def colliding_box(nx, ny, nz, sd_conc, dt, kernel=None, **kw):
    oi = h.box_opts(nx, ny, nz, sd_conc, dx=10., strict_fp=False, **kw)
    oi.dt = dt
    oi.dry_distros = {(1e-10, 0.): golovin_spectrum}
    oi.kernel = kernel if kernel is not None else lgrngn.kernel_t.hall_davis_no_waals
    oi.terminal_velocity = lgrngn.vt_t.beard77fast
    oi.dbg_flags = int(lgrngn.dbg.TAG)
    oi.n_sd_max = int(sd_conc * nx * ny * nz * 1.1) + 64
    return oi


def device_by_tag(hip):
    """the device's living super-droplets in tag order, read without disturbing the production storage ("raw_*")"""
    ijk = hip.state_u64("raw_ijk")
    alive = ijk != DEAD
    tag = hip.state_real("raw_tag")[alive]
    order = np.argsort(tag, kind="stable")
    out = {"tag": tag[order], "ijk": ijk[alive][order], "n": hip.state_u64("raw_n")[alive][order]}
    for nm in ("rw2", "rd3", "kappa", "x", "y", "z"):
        out[nm] = hip.state_real("raw_" + nm)[alive][order]
    return out


def oracle_by_tag(orc):
    tag = orc.state_real("tag")
    order = np.argsort(tag, kind="stable")
    out = {"tag": tag[order], "ijk": orc.state_u64("ijk")[order], "n": orc.state_u64("n")[order]}
    for nm in ("rw2", "rd3", "kappa", "x", "y", "z"):
        out[nm] = orc.state_real(nm)[order]
    return out


def reverse_replay_step(orc, hip, opts, fo, fh, rhod, C, sstp_coal=1, rebase=True, free_bars=(1e-4, 1e-6, 1e-7), free_field_bars=None,
                        other_root=None):
    """one full step of both; returns (collisions in the oracle, device storage extent before the step's end)
    other_root: None, or a dict {"allow": k, "seen": 0} -- up to k droplets of the step may sit on ANOTHER ROOT of the backward-Euler
    equation than the oracle's (see test_reverse_replay_at_production_size); they are counted in its "seen" entry"""
    orc.step_sync(opts, fo[0], fo[1], rhod, **C)
    hip.step_sync(opts, fh[0], fh[1], rhod, **C)
    tag_o = orc.state_real("tag")                       # the oracle's ids at coalescence time (it compacts at the END of step_async)
    if opts.cond and not rebase:
        # opts_init.cond_solver = 1: the device follows the reference's TOMS748 iterates -- nothing is re-based, the two runs are FREE;
        # th and rv at the strict bars, the wet radii identical but where an ulp moved a stopping decision (a handful of droplets)
        np.testing.assert_allclose(fh[0], fo[0], rtol=(free_field_bars or h.cond_bars(True))[0])
        np.testing.assert_allclose(fh[1], fo[1], rtol=(free_field_bars or h.cond_bars(True))[1])
        d = device_by_tag(hip)
        order = np.argsort(tag_o, kind="stable")
        assert np.array_equal(d["tag"], tag_o[order])
        err = np.abs(d["rw2"] / orc.state_real("rw2")[order] - 1)
        # (this spectrum's drops are all but insoluble, kappa = 1e-10, and as large dry as wet: rw^3 - rd^3 cancels to eight digits, and
        # the fast arithmetic forms it with one rounding where the strict order has two -- 1e-7 here, 1e-14 on an aerosol's droplets)
        # free_bars: (every droplet, 99.9 % of them, the median)
        far = int((err >= free_bars[0]).sum())
        if other_root is not None:
            other_root["seen"] = max(other_root["seen"], far)         # (a free run keeps them: the count of the latest step)
        assert far <= (other_root["allow"] if other_root else 0) and np.quantile(err, .999) < free_bars[1] and np.median(err) < free_bars[2], (far, err.max(), np.quantile(err, .999), np.median(err))
    elif opts.cond:
        # This box holds 1 g of liquid water per m^3.  The root finder's tolerance on rw2 (2^-14 relative, config.hpp:39 through
        # toms748.hpp:267-282: both the reference's midpoint and the lean solver's root lie within it of each other) is 1.5 x 2^-14 of
        # a droplet's mass; were every droplet of a cell off to the same side, rv would differ by that share of the cell's liquid
        # water, th by d_th/d_rv ~ -(th / T) l_v / c_pd times as much: the bound the reference's own tolerance implies, cell by cell.
        bar_med = h.cond_bars(False)[2]
        orc.diag_all(); orc.diag_wet_mom(3)
        r_liq = 4. / 3. * np.pi * 1e3 * orc.outbuf_array().reshape(fo[1].shape)
        bound_rv = 1.5 * 2. ** -14 * r_liq + 1e-12
        assert (np.abs(fh[1] - fo[1]) <= bound_rv).all(), float((np.abs(fh[1] - fo[1]) / bound_rv).max())
        assert (np.abs(fh[0] - fo[0]) <= 2.7e3 * bound_rv).all(), float((np.abs(fh[0] - fo[0]) / (2.7e3 * bound_rv)).max())
        d = device_by_tag(hip)
        order = np.argsort(tag_o, kind="stable")
        assert np.array_equal(d["tag"], tag_o[order])
        rw2_o = orc.state_real("rw2")
        err = np.abs(d["rw2"] / rw2_o[order] - 1)
        far = int((err >= 1e-4).sum())
        if other_root is not None:
            other_root["seen"] += far
        assert far <= (other_root["allow"] if other_root else 0) and np.median(err) < bar_med, (far, err.max(), np.median(err))
        # re-base: the oracle goes on from the device's wet radii, th and rv
        rw2_new = np.empty_like(rw2_o)
        rw2_new[order] = d["rw2"]
        orc.set_state_real("rw2", rw2_new)
        fo[0][...] = fh[0]
        fo[1][...] = fh[1]
        orc.set_state_real("th", fh[0].ravel())          # (the cell fields that step_async derives T, p and the viscosity from)
        orc.set_state_real("rv", fh[1].ravel())
    n_before = orc.state_u64("n").copy()
    hip.step_async(opts)
    if opts.coal:
        for call in range(sstp_coal):
            u01 = hip.rng_dump(call, 0)
            un_dev, tag_dev, ijk_dev = hip.rng_dump(call, 1), hip.rng_dump(call, 2), hip.rng_dump(call, 3)
            alive = ijk_dev != DEAD
            assert int(alive.sum()) == u01.size == tag_o.size, (int(alive.sum()), u01.size, tag_o.size)
            # un[oracle id] = the key of the device's copy of that super-droplet
            order = np.argsort(tag_dev[alive], kind="stable")
            tags_sorted, keys_sorted = tag_dev[alive][order], un_dev[alive][order]
            assert np.unique(tags_sorted).size == tags_sorted.size
            pos = np.searchsorted(tags_sorted, tag_o)
            assert np.array_equal(tags_sorted[pos], tag_o)
            orc.rng_replay_push(1, keys_sorted[pos])
            orc.rng_replay_push(0, u01)
    orc.step_async(opts)
    assert orc.rng_replay_pending() == 0
    col = orc.state_real("col") if opts.coal else np.zeros(0)
    return int((col >= 1).sum()) if col.size else 0, n_before


@pytest.mark.parametrize("cond", [True, False])
def test_production_coalescence_on_the_devices_own_stream_matches_the_oracle(cond):
    nx, ny, nz, sd_conc, steps = 6, 5, 7, 64, 12
    oi = colliding_box(nx, ny, nz, sd_conc, dt=5.)
    fields = h.box_fields(oi)
    th, rv, rhod, C = fields
    orc = h.oracle_particles(oi)
    hip = h.hip_particles(oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)           # (its own Philox stream: no replay is ever pushed into the device)
    # spin-up on the oracle: the drops of this spectrum are far from equilibrium with the box's humidity and move th by 3.7 K in their
    # first steps; the comparison starts from the settled state
    so = lgrngn.opts_t()
    so.coal = so.adve = so.sedi = False
    for _ in range(8):
        orc.step_sync(so, th, rv, rhod, **C)
        orc.step_async(so)
    h.copy_state(orc, hip)
    assert hip.rng_replay_pending() == 0
    opts = lgrngn.opts_t()
    opts.cond = cond
    fo, fh = [th.copy(), rv.copy()], [th.copy(), rv.copy()]
    n0 = orc.n_part
    collisions, multi, reorderings, last_first_tag = 0, 0, 0, None
    for it in range(steps):
        ncol, _ = reverse_replay_step(orc, hip, opts, fo, fh, rhod, C)
        collisions += ncol
        multi += int((orc.state_real("col") > 1).sum())
        assert hip.n_part == orc.n_part, it
        d, o = device_by_tag(hip), oracle_by_tag(orc)
        assert np.array_equal(d["tag"], o["tag"]), it                       # the same super-droplets are alive
        assert np.array_equal(d["n"], o["n"]), (it, int((d["n"] != o["n"]).sum()))
        assert np.array_equal(d["ijk"], o["ijk"]), it
        assert np.array_equal(d["kappa"], o["kappa"]), it
        np.testing.assert_allclose(d["rd3"], o["rd3"], rtol=1e-14, err_msg="rd3, step %d" % it)
        np.testing.assert_allclose(d["rw2"], o["rw2"], rtol=1e-13, err_msg="rw2, step %d" % it)
        for a in ("x", "y", "z"):
            np.testing.assert_allclose(d[a], o[a], rtol=1e-13, atol=1e-10, err_msg="%s, step %d" % (a, it))
        # a storage re-ordering shows as the device's storage beginning with another super-droplet / shrinking to the living ones
        raw_tag = hip.state_real("raw_tag")
        if last_first_tag is not None and (raw_tag.size == hip.n_part and raw_tag[0] != last_first_tag):
            reorderings += 1
        last_first_tag = raw_tag[0]
    # the run did what it is meant to cover: collisions (some of them multiple), super-droplets lost, the storage compacted and re-ordered
    assert collisions > 50 * steps / 12 and multi > 0, (collisions, multi)
    assert orc.n_part < n0 * 31 // 32
    assert reorderings >= 1
    # and the observable state at the end through the ordinary getters: the same droplets (the device's storage has been re-ordered, so as sets)
    assert np.array_equal(np.sort(hip.state_u64("n")), np.sort(orc.state_u64("n")))
    assert np.array_equal(np.sort(hip.state_u64("ijk")), np.sort(orc.state_u64("ijk")))
    for prt in (orc, hip):
        prt.diag_all(); prt.diag_sd_conc()
    assert np.array_equal(hip.outbuf_array(), orc.outbuf_array())
    print("collisions %d (multiple %d), super-droplets %d -> %d, re-orderings %d" % (collisions, multi, n0, orc.n_part, reorderings))


def test_free_run_with_the_references_iterates_stays_on_the_oracle():
    """The same colliding box with opts_init.cond_solver = 1 (TOMS748's iterates in fast arithmetic, in the storage-order kernel) and
    NOTHING re-based: twelve full steps -- condensation, coalescence on the device's own stream, sedimentation, advection, lazy
    compaction, storage re-ordering -- of two free runs.  Multiplicities, cells and tags stay exact; the wet radii stay identical but
    for the few droplets where an ulp moved a stopping decision of the root finder (every one within its tolerance); th, rv at the
    strict bars.  What the lean solver's test has to re-base, this mode does not."""
    nx, ny, nz, sd_conc, steps = 6, 5, 7, 64, 12
    oi = colliding_box(nx, ny, nz, sd_conc, dt=5., cond_solver=1)
    fields = h.box_fields(oi)
    th, rv, rhod, C = fields
    orc = h.oracle_particles(oi)
    hip = h.hip_particles(oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)
    so = lgrngn.opts_t()
    so.coal = so.adve = so.sedi = False
    for _ in range(8):
        orc.step_sync(so, th, rv, rhod, **C)
        orc.step_async(so)
    h.copy_state(orc, hip)
    opts = lgrngn.opts_t()
    fo, fh = [th.copy(), rv.copy()], [th.copy(), rv.copy()]
    n0, collisions = orc.n_part, 0
    for it in range(steps):
        ncol, _ = reverse_replay_step(orc, hip, opts, fo, fh, rhod, C, rebase=False)
        collisions += ncol
        assert hip.n_part == orc.n_part, it
        d, o = device_by_tag(hip), oracle_by_tag(orc)
        assert np.array_equal(d["tag"], o["tag"]), it
        assert np.array_equal(d["n"], o["n"]), (it, int((d["n"] != o["n"]).sum()))
        assert np.array_equal(d["ijk"], o["ijk"]), it
        err = np.abs(d["rw2"] / o["rw2"] - 1)
        assert err.max() < 1e-4 and np.quantile(err, .999) < 1e-6, (it, err.max(), np.quantile(err, .999))
        np.testing.assert_allclose(d["rd3"], o["rd3"], rtol=1e-14)
        for a in ("x", "y"):
            np.testing.assert_allclose(d[a], o[a], rtol=1e-13, atol=1e-10, err_msg="%s, step %d" % (a, it))
        np.testing.assert_allclose(d["z"], o["z"], rtol=1e-13, atol=2e-3)      # (dt * vt of a drop whose rw2 differs by 1e-5 ... 1e-4, twelve steps)
    assert collisions > 50 and orc.n_part < n0


def test_reverse_replay_with_crowded_cells_and_substeps():
    """400 super-droplets per cell (the listed-cell sorts: one wave per cell) and two coalescence substeps (no fused death marks, the
    plain re-sort path), the geometric kernel"""
    nx, ny, nz, sd_conc, steps = 3, 2, 3, 400, 5
    oi = colliding_box(nx, ny, nz, sd_conc, dt=4., kernel=lgrngn.kernel_t.geometric, sstp_coal=2)
    fields = h.box_fields(oi)
    th, rv, rhod, C = fields
    orc = h.oracle_particles(oi)
    hip = h.hip_particles(oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)
    h.copy_state(orc, hip)
    opts = lgrngn.opts_t()
    opts.cond = False
    fo, fh = [th.copy(), rv.copy()], [th.copy(), rv.copy()]
    for it in range(steps):
        # (two coalescence calls per step; the oracle's ids do not change between them: it compacts at the end of step_async)
        reverse_replay_step(orc, hip, opts, fo, fh, rhod, C, sstp_coal=2)
        assert hip.n_part == orc.n_part, it
        d, o = device_by_tag(hip), oracle_by_tag(orc)
        assert np.array_equal(d["tag"], o["tag"]) and np.array_equal(d["n"], o["n"]) and np.array_equal(d["ijk"], o["ijk"]), it
        np.testing.assert_allclose(d["rd3"], o["rd3"], rtol=1e-14)
        np.testing.assert_allclose(d["rw2"], o["rw2"], rtol=1e-13)


@pytest.mark.parametrize("size,reorder_every", [(3, 0), (2, 3)])
def test_reverse_replay_on_the_slabs_of_the_multi_device_object(size, reorder_every, monkeypatch):
    """The same on a decomposed domain: the native multi-device object (all slabs on the one GPU), production path -- fast
    arithmetic, the slabs' re-sort left to the next condensation kernel (round 4), immigrants taking over the emigrants' storage slots,
    storage re-ordering -- against a ring of oracles, each fed the random stream that ITS slab's coalescence consumed on the device.
    Tags are unique over all slabs (lcx_set_state_real), so a droplet is followed across the slab faces: after every step every slab
    holds the same tags, multiplicities and cells as its oracle, rd3 to 1e-14, rw2 and positions to 1e-13."""
    monkeypatch.setenv("LCX_MULTI_DEVICE_MAP", ",".join(["0"] * size))
    nx, ny, nz, sd_conc, steps = 4 * size, 4, 6, 48, 8
    oi = colliding_box(nx, ny, nz, sd_conc, dt=4.)
    oi.reorder_every = reorder_every
    oi.n_sd_max = int(sd_conc * nx * ny * nz * 1.6) + 64 * size
    th, rv, rhod, C = h.box_fields(oi)
    orc = h.LocalRing(oi, size, h.oracle_particles, h.host_alloc)
    oi.dev_count = size
    mul = lgrngn.factory(lgrngn.backend_t.multi_HIP, oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    mul.init(th.copy(), rv.copy(), rhod.copy(), **C)
    so = lgrngn.opts_t()
    so.coal = so.adve = so.sedi = False
    for _ in range(6):                                         # (spin-up on the oracles, see the single-device test)
        orc.step(so, th, rv, rhod, **C)
    slabs = [mul.slab(r) for r in range(size)]
    for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
        ph.opts_init = po.opts_init
        h.copy_state(po, ph)
        tags = 1e7 * r + np.arange(po.n_part, dtype=np.float64)          # unique over the slabs
        po.set_state_real("tag", tags)
        ph.set_state_real("tag", tags)
    opts = lgrngn.opts_t()
    fo, fh = [th.copy(), rv.copy()], [th.copy(), rv.copy()]
    crossed = 0
    mul.set_profiling(True)
    for it in range(steps):
        orc.step_sync(opts, fo[0], fo[1], rhod, **C)
        mul.step_sync(opts, fh[0], fh[1], rhod, **C)
        tags_o = []
        for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
            # condensation at its own bars, then the oracle slab goes on from the device's wet radii, th and rv (see the module's head)
            d = device_by_tag(ph)
            t_o = po.state_real("tag")
            order = np.argsort(t_o, kind="stable")
            assert np.array_equal(d["tag"], t_o[order]), (it, r)
            rw2_o = po.state_real("rw2")
            err = np.abs(d["rw2"] / rw2_o[order] - 1)
            assert err.max() < 1e-4, (it, r, err.max())
            rw2_new = np.empty_like(rw2_o)
            rw2_new[order] = d["rw2"]
            po.set_state_real("rw2", rw2_new)
            sl = slice(orc.bfr[r], orc.bfr[r] + orc.nxl[r])
            np.testing.assert_allclose(fh[0][sl], fo[0][sl], rtol=2e-6)
            po.set_state_real("th", fh[0][sl].ravel())
            po.set_state_real("rv", fh[1][sl].ravel())
            tags_o.append(t_o)
        fo[0][...] = fh[0]
        fo[1][...] = fh[1]
        mul.step_async(opts)
        for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
            u01 = ph.rng_dump(0, 0)
            un_dev, tag_dev, ijk_dev = ph.rng_dump(0, 1), ph.rng_dump(0, 2), ph.rng_dump(0, 3)
            alive = ijk_dev != DEAD
            assert int(alive.sum()) == u01.size == tags_o[r].size, (it, r)
            order = np.argsort(tag_dev[alive], kind="stable")
            tags_sorted, keys_sorted = tag_dev[alive][order], un_dev[alive][order]
            pos = np.searchsorted(tags_sorted, tags_o[r])
            assert np.array_equal(tags_sorted[pos], tags_o[r]), (it, r)
            po.rng_replay_push(1, keys_sorted[pos])
            po.rng_replay_push(0, u01)
        orc.step_async(opts)
        for r, (po, ph) in enumerate(zip(orc.prts, slabs)):
            assert po.rng_replay_pending() == 0
            assert ph.n_part == po.n_part, (it, r)
            d, o = device_by_tag(ph), oracle_by_tag(po)
            assert np.array_equal(d["tag"], o["tag"]), (it, r)
            assert np.array_equal(d["n"], o["n"]) and np.array_equal(d["ijk"], o["ijk"]), (it, r)
            np.testing.assert_allclose(d["rd3"], o["rd3"], rtol=1e-14)
            np.testing.assert_allclose(d["rw2"], o["rw2"], rtol=1e-13)
            for a in ("x", "y", "z"):
                np.testing.assert_allclose(d[a], o[a], rtol=1e-13, atol=1e-9, err_msg="%s, step %d, slab %d" % (a, it, r))
            crossed += int(((d["tag"] // 1e7).astype(int) != r).sum())
    assert crossed > 0                                         # droplets did change slabs
    # the path under test: without a storage re-ordering in the run no slab re-sorted inside its exchange (the re-sort rode on the
    # condensation kernel); with one every third step the overlapped interior re-sort ran in those steps
    stages = mul.timings()
    assert ("exchange_sort_interior" in stages) == (reorder_every > 0), sorted(stages)
    assert "exchange_unpack" in stages and "cond" in stages


def oracle_from_device(oi, hip, th, rv, rhod, C, make_oracle):
    """an oracle holding the DEVICE's droplets: the device was initialised by its own generator (the built-in dry spectra of bench.py
    are sampled on the device) and has run on its production path -- nothing has been pushed into it, no set_particles (which would
    make it read the hygroscopicity array: the benchmarked kernel takes the run's single value as a scalar).  The oracle takes the
    living super-droplets in tag order (orc_set_particles) with the device's tags."""
    orc = make_oracle(oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    ijk = hip.state_u64("raw_ijk")
    alive = ijk != DEAD
    tag = hip.state_real("raw_tag")[alive]
    order = np.argsort(tag, kind="stable")
    g = lambda nm: hip.state_real("raw_" + nm)[alive][order]
    orc.set_particles(hip.state_u64("raw_n")[alive][order], g("rd3"), g("rw2"), g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_state_real("tag", tag[order])
    return orc


@pytest.mark.parametrize("workload,cond_solver,free", [("stratocumulus", 0, False), ("coal-stress", 0, False), ("stratocumulus", 1, True), ("stratocumulus", 0, True)],
                         ids=["stratocumulus", "coal_stress", "stratocumulus_toms748_free_run", "stratocumulus_lean_free_run"])
def test_reverse_replay_at_production_size(workload, cond_solver, free):
    """VERDICT r04 missing 2: the benchmarked path against the oracle AT A SIZE WHERE ITS LAUNCH GEOMETRY IS THE BENCHMARK'S.
    bench.py's options and fields (make_opts_init / make_fields; fast arithmetic, set HERE -- see the note below), 128 x 128 x 16 cells x
    64 = 2^24 super-droplets: the multi-workgroup windows of the bucket ranking on its side stream, the scatter carried by the
    storage-order condensation kernel with the run's hygroscopicity as a scalar, k_coal on the device's own Philox stream, storage
    re-ordering every third step (the dead dropped there: a lazy compaction), nothing replayed into the device and no set_particles.
    The device spins up alone for three steps; the oracle (OpenMP build, bit-identical to the serial one) then takes over ITS droplets
    and both run seven more steps, the oracle on the random numbers the device's coalescence consumed.  After every step: the same tags
    alive, multiplicities, cells, kappa exact, rd3 1e-14.
      * stratocumulus / coal_stress (cond_solver = 0, the headline): the lean solver's wet radii, th and rv are compared after each
        condensation at their own bars (th and rv inside what the root finder's tolerance implies cell by cell, the wet radii's median
        3e-5) and the oracle is re-based on them; rw2 and positions then 1e-13 from identical inputs.  Every wet radius within 1e-4,
        every one (without the list of droplets whose bracket can hold several roots -- k_cond_lean_listed, see `other_root` below --
        up to 65 of 1.7e7 per step sat on another root of the step's equation than TOMS748's);
      * stratocumulus_toms748_free_run (cond_solver = 1, the API default): nothing is re-based.  th 1e-8, rv 1e-7 (measured 3e-10 / 6e-9),
        the wet radii's median below 1e-10, 99.9 % of them below 5e-5 (1.4e-5: one bisection of TOMS748's last bracket where an ulp of the
        fast arithmetic moved a stopping decision), every one below 5e-2 (2e-3: an activating droplet amplifies its difference);
      * stratocumulus_lean_free_run (round 5): the headline solver, nothing re-based, seven steps: th 3e-7, rv 3e-6 (measured 6e-9 / 1.3e-7),
        the wet radii's median 2e-6 after seven steps (it grows from 4e-11 by a factor of five a step: roots that differ from the
        reference's bracket midpoints by up to the tolerance add up to 1e-7 of a cell's vapour, and the cell's haze follows the humidity),
        99.9 % of them below 3e-4 (5.5e-5); the same collisions, cells and multiplicities throughout.
    On the coal-stress spectrum of bench.py the pairs collide.
    NOTE: as first committed (round 5, 4556076) this test did not set strict_fp and, the suite's opts_init_t being pinned to the
    parity mode (_harness.py), ran the STRICT kernels -- not the benchmarked path; the figures above are of the test as it is now."""
    import bench
    nx, ny, nz, steps = 128, 128, 16, 7
    oi = bench.make_opts_init(nx, ny, nz, 64, 40., 1, 1, 44, workload)
    oi.dbg_flags = int(lgrngn.dbg.TAG)
    oi.reorder_every = 3
    oi.strict_fp = False               # (bench.py's main sets it from its arguments; the suite's opts_init_t is pinned to the parity mode, _harness.py)
    oi.cond_solver = cond_solver
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(nx, ny, nz, 0, nx, np, np.float64)
    sh = (nx, ny, nz)
    th, rv, rhod = [np.ascontiguousarray(np.broadcast_to(a_, sh)) for a_ in (th, rv, rhod)]
    C = {"Cx": np.ascontiguousarray(np.broadcast_to(Cx, (nx + 1, ny, nz))), "Cy": np.ascontiguousarray(np.broadcast_to(Cy, (nx, ny + 1, nz))),
         "Cz": np.ascontiguousarray(np.broadcast_to(Cz, (nx, ny, nz + 1)))}
    hip = h.hip_particles(oi)
    fh = [th.copy(), rv.copy()]
    hip.init(fh[0], fh[1], rhod, **C)
    assert hip.n_part >= (1 << 24) - 64
    opts = lgrngn.opts_t()
    for _ in range(3):                                          # the device alone, on its production path
        hip.step_sync(opts, fh[0], fh[1], rhod, **C)
        hip.step_async(opts)
    orc = oracle_from_device(oi, hip, fh[0], fh[1], rhod, C, h.oracle_omp_particles)
    fo = [fh[0].copy(), fh[1].copy()]
    orc.set_state_real("th", fo[0].ravel())
    orc.set_state_real("rv", fo[1].ravel())
    assert orc.n_part == hip.n_part
    n0, collisions, reorderings, last_first_tag = orc.n_part, 0, 0, None
    # The lean solver returns A root of rw2' = rw2 + dt F(rw2') inside the reference's bracket.  A handful of droplets per step have
    # several there -- a 0.8 um droplet on a 5 nm core that can either shrink to half its radius or evaporate down to the core within the
    # step (the Kelvin term takes over next to the core), a dry particle of no hygroscopicity in supersaturated air -- and TOMS748's
    # iterates may pick another one: WITHOUT the list of such droplets (dbg COND_NO_LIST, the state of rounds 3-4) up to 65 of 1.7e7 per
    # step on the stratocumulus box and 12-40 on the coal-stress spectrum (126 and 119 droplet-steps in the seven steps).  Since late
    # round 5 k_cond_lean hands the droplets whose bracket can hold several roots to the reference's iterates (k_cond_lean_listed):
    # NONE is left in the re-based runs; the free run of the lean solver, whose state drifts from the oracle's at the tolerance's scale,
    # ends with four.
    other_root = {"allow": (int(1e-6 * orc.n_part) if free else 0) if cond_solver == 0 else 0, "seen": 0}
    for it in range(steps):
        # (the free run's wet radii: an aerosol's droplets are identical to the oracle's -- median below 1e-10 -- but where an ulp of the
        # fast arithmetic moved one of TOMS748's stopping decisions, and then by what one bisection of its last bracket is worth: 8e-6
        # for one droplet in a thousand of this box, 3e-5 at most)
        # ... and a FREE run carries a droplet's difference into its next step, where a droplet that is activating amplifies it (the
        # growth of a droplet at its critical radius is unstable): up to 8e-3 for the worst few of 1.7e7 in the course of seven steps --
        # the thousandth-worst stays at one bisection (1.3e-5), the median at 1e-11
        ncol, _ = reverse_replay_step(orc, hip, opts, fo, fh, rhod, C, rebase=not free, other_root=other_root,
                                      free_bars=(5e-2, 5e-5, h.cond_bars(True)[2]) if cond_solver == 1 else (5e-2, 3e-4, h.cond_bars(False)[2]),
                                      free_field_bars=None if cond_solver == 1 else h.cond_bars(False)[:2])
        collisions += ncol
        # (the benchmarked path, read back from the object: fast arithmetic, the solver asked for, the storage-order kernels, none of
        # the switches that would take it off bench.py's path)
        h.assert_mode(hip, False, cond_solver, "lean" if cond_solver == 0 else "fold_toms748",
                      no_dbg=("NO_DEFERRED_SORT", "COND_SORTED_ORDER", "RANK_BY_COUNTING", "SHUFFLE_PHILOX", "EAGER_COMPACT", "COND_NO_LIST", "NO_COND_PRE"))
        assert hip.n_part == orc.n_part, it
        d, o = device_by_tag(hip), oracle_by_tag(orc)
        assert np.array_equal(d["tag"], o["tag"]), it
        assert np.array_equal(d["n"], o["n"]), (it, int((d["n"] != o["n"]).sum()))
        assert np.array_equal(d["ijk"], o["ijk"]), (it, int((d["ijk"] != o["ijk"]).sum()))
        assert np.array_equal(d["kappa"], o["kappa"]), it
        np.testing.assert_allclose(d["rd3"], o["rd3"], rtol=1e-14, err_msg="rd3, step %d" % it)
        if not free:
            np.testing.assert_allclose(d["rw2"], o["rw2"], rtol=1e-13, err_msg="rw2, step %d" % it)
            for a_ in ("x", "y", "z"):
                np.testing.assert_allclose(d[a_], o[a_], rtol=1e-13, atol=1e-9, err_msg="%s, step %d" % (a_, it))
        else:
            err = np.abs(d["rw2"] / o["rw2"] - 1)
            print("free run, step %d: rw2 max %.2e  99.9 %% %.2e  median %.2e; th %.2e rv %.2e" % (it, err.max(), np.quantile(err, .999), np.median(err), np.abs(fh[0] / fo[0] - 1).max(), np.abs(fh[1] / fo[1] - 1).max()))
            near = err < 5e-2
            assert (~near).sum() <= other_root["allow"] and np.quantile(err, .999) < (1e-4 if cond_solver == 1 else 3e-4), (it, err.max(), np.quantile(err, .999))
            assert np.median(err) < (1e-9 if cond_solver == 1 else h.cond_bars(False)[2]), (it, np.median(err))
            for a_ in ("x", "y"):
                np.testing.assert_allclose(d[a_], o[a_], rtol=1e-13, atol=1e-9, err_msg="%s, step %d" % (a_, it))
            np.testing.assert_allclose(d["z"][near], o["z"][near], rtol=1e-13, atol=2e-3)
        # a storage re-ordering shows as the living super-droplets standing in another order in the device's storage
        raw_tag = hip.state_real("raw_tag")[hip.state_u64("raw_ijk") != DEAD]
        if last_first_tag is not None and not np.array_equal(raw_tag, last_first_tag):
            reorderings += 1
        last_first_tag = raw_tag
    assert reorderings >= 2, reorderings
    if workload == "coal-stress":
        assert collisions > 1e4 and orc.n_part < n0, (collisions, n0, orc.n_part)
    print("%s: collisions %d, super-droplets %d -> %d, re-orderings %d, droplets on another root %d (allowed per step: %d)" % (
        workload, collisions, n0, orc.n_part, reorderings, other_root["seen"], other_root["allow"]))
