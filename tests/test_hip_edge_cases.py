"""Edge cases of the hot path on the device against the oracle: empty and nearly empty populations, single cells / planes,
one super-droplet per cell, everything switched off, all super-droplets raining out."""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu


def exact(a, b, what):
    assert np.array_equal(a, b), what


def run_pair(oi, steps, opts=None, fields=None, coal_replay=True):
    fields = fields or h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = opts or lgrngn.opts_t()
    th, rv, rhod, C = fields
    for _ in range(steps):
        tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
        orc.step_sync(opts, tho, rvo, rhod, **C)
        hip.step_sync(opts, thh, rvh, rhod, **C)
        if opts.coal and coal_replay:
            h.push_coal_replay(orc, hip, oi.sstp_coal)
        orc.step_async(opts)
        hip.step_async(opts)
        assert hip.n_part == orc.n_part
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
    return orc, hip


@pytest.mark.parametrize("dims", [(1, 0, 0), (1, 0, 1), (1, 1, 1), (2, 1, 1), (1, 0, 7), (9, 0, 1)])
def test_single_cells_and_planes(dims):
    oi = h.box_opts(*dims, 32, sedi_switch=dims[2] > 0)
    opts = lgrngn.opts_t()
    opts.sedi = dims[2] > 0
    orc, hip = run_pair(oi, 3, opts)
    exact(hip.state_u64("n"), orc.state_u64("n"), "n")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=2e-4)


def test_one_super_droplet_per_cell():
    """no pair anywhere: coalescence must be a no-op that still consumes its random numbers"""
    oi = h.box_opts(4, 3, 5, 1)
    orc, hip = run_pair(oi, 3)
    exact(hip.state_u64("n"), orc.state_u64("n"), "n")
    assert np.all(orc.state_real("col")[:-1] == 0)


def test_everything_switched_off():
    oi = h.box_opts(3, 2, 4, 16)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = opts.coal = False
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    before = {a: hip.get_attr(a) for a in ("rw2", "x", "y", "z")}
    orc2, hip2 = run_pair(oi, 2, opts, fields)
    exact(hip2.get_attr("rw2"), before["rw2"], "rw2")
    for a in ("x", "y", "z"):               # the boundary condition is applied regardless (particles_step.ipp:480-481): the periodic
        exact(hip2.get_attr(a), orc2.get_attr(a), a)                      # wrap a + fmod(x - a + 10 L, L) rounds, identically
        np.testing.assert_allclose(hip2.get_attr(a), before[a], rtol=1e-11)


def test_all_super_droplets_rain_out_and_the_steps_go_on():
    """mm-sized drops in a shallow box: after a few steps no super-droplet is left; stepping and diagnosing an empty
    population must work (and agree with the oracle)"""
    oi = h.box_opts(3, 2, 3, 8, dx=10., coal_switch=False)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    g = lambda nm: orc.state_real(nm)
    rw2 = np.full(orc.n_part, (2e-3) ** 2)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    th, rv, rhod, C = fields
    for it in range(12):
        for pr in (orc, hip):
            pr.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
            pr.step_async(opts)
        assert hip.n_part == orc.n_part
    assert orc.n_part == 0
    for pr in (orc, hip):
        pr.diag_all(); pr.diag_sd_conc()
    exact(hip.outbuf_array(), orc.outbuf_array(), "sd_conc of an empty population")
    assert not hip.outbuf_array().any()
    for pr in (orc, hip):
        pr.diag_all(); pr.diag_wet_mom(3)
    exact(hip.outbuf_array(), orc.outbuf_array(), "moment of an empty population")
    po, ph = orc.diag_puddle(), hip.diag_puddle()
    np.testing.assert_allclose(ph["particle_number"], po["particle_number"], rtol=1e-12)
    assert len(hip.get_attr("rw2")) == 0


def test_n_sd_max_is_enforced():
    oi = h.box_opts(2, 2, 2, 16)
    oi.n_sd_max = 16 * 8 - 1
    th, rv, rhod, C = h.box_fields(oi)
    with pytest.raises(RuntimeError, match="n_sd_max"):
        h.hip_particles(oi).init(th, rv, rhod, **C)


@pytest.mark.parametrize("name,kw", [
    ("turb_all", dict(turb_adve_switch=True, turb_cond_switch=True, turb_coal_switch=True, kernel=lgrngn.kernel_t.onishi_hall,
                      kernel_parameters=np.array([66.]), SGS_mix_len=np.linspace(20., 40., 5))),
    ("incloud_adaptive", dict(diag_incloud_time=True, exact_sstp_cond=True, sstp_cond=4, sstp_cond_act=8, sstp_cond_mix=False,
                              adaptive_sstp_cond=True)),
    ("pp_mix", dict(exact_sstp_cond=True, sstp_cond=3)),
    ("pred_corr", dict(adve_scheme=lgrngn.as_t.pred_corr))])
def test_float_build_tracks_double_with_the_widened_options(name, kw):
    """particles_t<float>: every option family of SURVEY 8(f) steps in single precision and stays close to the double run"""
    res = {}
    for real in (np.float32, np.float64):
        oi = h.box_opts(4, 3, 5, 32, **kw)
        th, rv, rhod, C = h.box_fields(oi)
        f = [a.astype(real) for a in (th, rv, rhod)]
        Cf = {k: v.astype(real) for k, v in C.items()}
        pr = lgrngn.particles_t(oi, real)
        pr.init(*f, **Cf)
        opts = lgrngn.opts_t()
        opts.turb_adve = opts.turb_cond = opts.turb_coal = name == "turb_all"
        extra = dict(diss_rate=(1e-3 * np.ones(th.shape)).astype(real)) if name == "turb_all" else {}
        for _ in range(4):
            pr.step_sync(opts, f[0].copy(), f[1].copy(), f[2], **extra, **Cf)
            pr.step_async(opts)
        pr.diag_all(); pr.diag_wet_mom(3)
        res[real] = (pr.n_part, float(pr.outbuf_array().sum()))
        assert np.isfinite(res[real][1]) and res[real][1] > 0
    assert abs(res[np.float32][0] - res[np.float64][0]) <= 8
    assert abs(res[np.float32][1] / res[np.float64][1] - 1) < 2e-2


@pytest.mark.parametrize("walls", [dict(open_side_walls=True), dict(periodic_topbot_walls=True),
                                   dict(open_side_walls=True, periodic_topbot_walls=True)])
@pytest.mark.parametrize("dims", [(5, 0, 6), (4, 3, 5)])
def test_wall_options_match_oracle(dims, walls):
    """open side walls remove what leaves through x / y faces, periodic top and bottom wrap z instead of raining out
    (bcnd.ipp:114-368); multiplicities, cell indices and positions against the oracle"""
    oi = h.box_opts(*dims, 32, dx=15., coal_switch=False, **walls)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    orc, hip = run_pair(oi, 5, opts)
    for nm in ("n", "ijk", "sorted_id"):
        exact(hip.state_u64(nm), orc.state_u64(nm), nm)
    for a in ("x", "y", "z"):
        if getattr(oi, "n" + a):
            np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14, atol=1e-9)
    po, ph = orc.diag_puddle(), hip.diag_puddle()
    np.testing.assert_allclose(ph["particle_number"], po["particle_number"], rtol=1e-12)


def test_subsidence_matches_oracle():
    """opts.subs with a large-scale vertical velocity profile w_LS (subs.ipp:13-25; unit/lgrngn_subsidence.py runs on the oracle)"""
    oi = h.box_opts(4, 3, 6, 24, dx=25., coal_switch=False, subs_switch=True, w_LS=np.linspace(0.5, 3., 6))
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    opts.subs = True
    orc, hip = run_pair(oi, 4, opts)
    for nm in ("n", "ijk", "sorted_id"):
        exact(hip.state_u64(nm), orc.state_u64(nm), nm)
    np.testing.assert_allclose(hip.get_attr("z"), orc.get_attr("z"), rtol=1e-14, atol=1e-9)


@pytest.mark.parametrize("kw", [dict(aerosol_independent_of_rhod=True), dict(rd_min=5e-9, rd_max=2e-7),
                                dict(aerosol_independent_of_rhod=True, aerosol_conc_factor=np.linspace(1.5, 0.5, 5))])
def test_init_options_match_oracle(kw):
    """initialisation variants: multiplicities independent of the air density, a vertical profile of the aerosol concentration,
    manual bin edges, a separate seed for the initial sampling (init_n.ipp, init_count_num.ipp, particles_init.ipp)"""
    oi = h.box_opts(3, 2, 5, 32, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields, force_state=False)
    assert hip.n_part == orc.n_part
    exact(hip.state_u64("n"), orc.state_u64("n"), "multiplicities")
    np.testing.assert_allclose(hip.get_attr("rd3"), orc.get_attr("rd3"), rtol=1e-13)
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-6)
    for a in ("x", "y", "z"):
        np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14)


def test_init_option_checks_and_the_separate_init_seed():
    oi = h.box_opts(3, 2, 5, 16, aerosol_conc_factor=np.linspace(0.5, 1.5, 5))
    th, rv, rhod, C = h.box_fields(oi)
    with pytest.raises(RuntimeError, match="aerosol_independent_of_rhod"):          # init_sanity_check.ipp:123-127
        h.hip_particles(oi).init(th, rv, rhod, **C)
    # rng_seed_init_switch: the initial sampling has its own seed, the run's stream is seeded by rng_seed afterwards
    def initial(seed, seed_init):
        o = h.box_opts(3, 2, 5, 16, rng_seed=seed, rng_seed_init=seed_init, rng_seed_init_switch=True)
        pr = h.hip_particles(o)
        pr.init(th.copy(), rv.copy(), rhod.copy(), **C)
        return pr.get_attr("rd3"), pr.get_attr("x")
    a, b, c = initial(1, 7), initial(2, 7), initial(1, 8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], c[0])


def test_consecutive_range_selections_match_oracle():
    """diag_*_rng_cons narrow a previous selection (particles_diag.ipp:300-345)"""
    oi = h.box_opts(3, 2, 4, 48)
    orc, hip = h.make_pair(oi, h.box_fields(oi))
    for pr in (orc, hip):
        pr.diag_dry_rng(0., 1.)
        pr.diag_wet_rng_cons(1e-8, 2e-6)
        pr.diag_kappa_rng_cons(.5, 1.)
        pr.diag_dry_rng_cons(2e-8, 1e-7)
        pr.diag_wet_mom(0)
    exact(hip.outbuf_array(), orc.outbuf_array(), "concentration of the narrowed selection")
    assert orc.outbuf_array().sum() > 0
    with pytest.raises(RuntimeError):
        fresh = h.hip_particles(oi)
        th, rv, rhod, C = h.box_fields(oi)
        fresh.init(th, rv, rhod, **C)
        fresh.diag_dry_rng_cons(0., 1.)               # consecutive selection without a selection


def test_sgs_velocity_moments_and_water_selection():
    """diag_up_mom / diag_vp_mom / diag_wp_mom (particles_diag.ipp:463-480) and diag_water_cons (:346-349) against the oracle"""
    oi = h.box_opts(4, 3, 4, 24, coal_switch=False, turb_adve_switch=True, SGS_mix_len=np.linspace(20., 40., 4))
    th, rv, rhod, C = h.box_fields(oi)
    orc, hip = h.make_pair(oi, (th, rv, rhod, C))
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False
    opts.turb_adve = True
    diss = 1e-3 * np.ones(th.shape)
    for _ in range(2):
        for pr in (orc, hip):
            pr.step_sync(opts, th.copy(), rv.copy(), rhod, diss_rate=diss, **C)
        for arr in h.oracle_rng_preview(orc, [(2, orc.n_part)] * 3):
            hip.rng_replay_push(2, arr)
        orc.step_async(opts)
        hip.step_async(opts)
    for fn in ("diag_up_mom", "diag_vp_mom", "diag_wp_mom"):
        for pr in (orc, hip):
            pr.diag_all()
            getattr(pr, fn)(2)
        np.testing.assert_allclose(hip.outbuf_array(), orc.outbuf_array(), rtol=1e-9)
        assert orc.outbuf_array().sum() > 0
    for pr in (orc, hip):
        pr.diag_dry_rng(2e-8, 1.)
        pr.diag_water_cons()
        pr.diag_wet_mom(0)
    exact(hip.outbuf_array(), orc.outbuf_array(), "water among a dry-radius selection")
    plain = h.hip_particles(h.box_opts(4, 3, 4, 8))
    plain.init(th, rv, rhod, **C)
    plain.diag_all()
    with pytest.raises(RuntimeError, match="SGS velocity"):
        plain.diag_up_mom(1)


@pytest.mark.parametrize("strict_fp", [True, False])
def test_production_storage_order_holds_the_oracles_droplets(strict_fp):
    """The production rules for the storage -- dead super-droplets dropped lazily, storage gathered into the cell order every
    reorder_every steps (ids renumbered), the cells left in the next coalescence's shuffled order -- against the ORACLE, which keeps
    the reference's id order: without coalescence a step needs no random number, so no replayed stream forces the reference's order on
    the device (the state is handed over with set_particles).  Eight steps of condensation + advection + sedimentation with
    reorder_every = 3 and precipitation through the floor: the same droplets (matched by dry radius, which nothing changes here),
    multiplicities exact, positions to 1e-13, wet radii at the substep tolerance times the steps, the same th / rv."""
    oi = h.box_opts(6, 5, 7, 48, coal_switch=False, strict_fp=strict_fp, reorder_every=3)
    fields = h.box_fields(oi)
    th, rv, rhod, C = fields
    orc = h.oracle_particles(oi)
    hip = h.hip_particles(oi)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)           # (its own Philox draws: overwritten by the oracle's state below)
    g = orc.state_real
    rw2, z, n = g("rw2").copy(), g("z").copy(), orc.state_u64("n").copy()
    low = np.nonzero(z < oi.dz)[0][::7]                        # a few drizzle drops just above the floor: they precipitate within the run
    rw2[low] = (1e-4) ** 2                                     # (multiplicity 1: their water is nothing to the cell)
    z[low] = 0.5 + 3.0 * (np.arange(low.size) % 5) / 5.
    n[low] = 1
    args = (n, g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), z)
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = False
    tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
    n0 = orc.n_part
    for it in range(8):
        orc.step_sync(opts, tho, rvo, rhod, **C)
        hip.step_sync(opts, thh, rvh, rhod, **C)
        orc.step_async(opts)
        hip.step_async(opts)
        assert hip.n_part == orc.n_part
    assert orc.n_part < n0                                     # (precipitation happened: dead super-droplets went through the lazy removal)
    ko, kh = np.argsort(orc.get_attr("rd3"), kind="stable"), np.argsort(hip.get_attr("rd3"), kind="stable")
    assert np.array_equal(orc.get_attr("rd3")[ko], hip.get_attr("rd3")[kh])
    assert np.array_equal(orc.state_u64("n")[ko], hip.state_u64("n")[kh])
    for a_ in ("x", "y"):
        np.testing.assert_allclose(hip.get_attr(a_)[kh], orc.get_attr(a_)[ko], rtol=1e-13, atol=1e-10)
    # (sedimentation: 8 x dt x vt(rw2), rw2 to 1e-4 per step; fast arithmetic measured 7.7e-5 m)
    np.testing.assert_allclose(hip.get_attr("z")[kh], orc.get_attr("z")[ko], rtol=1e-13, atol=1e-5 if strict_fp else 2e-4)
    np.testing.assert_allclose(hip.get_attr("rw2")[kh], orc.get_attr("rw2")[ko], rtol=8e-4)
    th_tol, rv_tol, _ = h.cond_bars(strict_fp)
    np.testing.assert_allclose(thh, tho, rtol=8 * th_tol)
    np.testing.assert_allclose(rvh, rvo, rtol=8 * rv_tol)
    # the device's own order is a valid cell-sorted order of its renumbered storage
    sid, sijk, ijk = hip.state_u64("sorted_id"), hip.state_u64("sorted_ijk"), hip.state_u64("ijk")
    assert np.array_equal(ijk[sid], sijk) and np.all(np.diff(sijk.astype(np.int64)) >= 0)


@pytest.mark.parametrize("strict_fp", [True, False])
def test_a_nan_wet_radius_poisons_its_cell_and_nothing_else(strict_fp):
    """The per-cell sums of fast arithmetic are taken in fixed point (order-independent, lcx_kernels.hpp k_cond_cellfinish): a NaN among
    a cell's addends must still make that cell's th and rv NaN, as the floating-point sums of the strict arithmetic and of the
    reference do -- and leave every other cell alone"""
    oi = h.box_opts(4, 3, 5, 32, strict_fp=strict_fp, coal_switch=False)
    th, rv, rhod, C = h.box_fields(oi)
    hip = h.hip_particles(oi)
    hip.init(th, rv, rhod, **C)
    rw2 = hip.get_attr("rw2")
    ijk = hip.state_u64("ijk")
    victim = 7 * 32 + 3
    rw2[victim] = np.nan
    hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), rw2, hip.get_attr("kappa"), hip.state_real("vt"), hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    thh, rvh = th.copy(), rv.copy()
    hip.step_sync(opts, thh, rvh, rhod, **C)
    bad = np.isnan(thh.ravel())
    assert bad.sum() == 1 and bad[int(ijk[victim])] and np.isnan(rvh.ravel()[int(ijk[victim])])
    assert np.isfinite(rvh.ravel()[~bad]).all()
