"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: integers (cell index, sort permutation, counts, multiplicities under a replayed random
stream) bit-exact; cell thermodynamics rtol 1e-12; vt rtol 1e-12; positions rtol 1e-14; rw2 after a
condensation substep rtol 1e-4 (the root finder stops at a 2^-15 bracket: a 1-ulp libm difference may move
the stopping iteration) and exact for the overwhelming majority; th / rv after step_cond rtol 1e-7.
"""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu


def exact(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), (what, int(np.sum(a != b)), "mismatches of", a.size)


def step_pair(orc, hip, opts, fields, coal_replay=True):
    th, rv, rhod, C = fields
    tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
    orc.step_sync(opts, tho, rvo, rhod, **C)
    hip.step_sync(opts, thh, rvh, rhod, **C)
    if opts.coal and coal_replay:
        h.push_coal_replay(orc, hip, orc.opts_init.sstp_coal)
    orc.step_async(opts)
    hip.step_async(opts)
    return (tho, rvo), (thh, rvh)


# ------------------------------------------------------------------ cell thermodynamics (a5)
@pytest.mark.parametrize("RH_formula", list(lgrngn.RH_formula_t))
def test_cell_fields(RH_formula):
    oi = h.box_opts(5, 3, 7, 8, RH_formula=RH_formula)
    orc, hip = h.make_pair(oi, h.box_fields(oi, supersat=False))
    for st in ("hskpng_Tpr", "hskpng_mfp"):
        orc.stage(st)
        hip.stage(st)
    for f in ("T", "p", "RH", "eta", "dv", "lambda_D", "lambda_K"):
        np.testing.assert_allclose(hip.state_real(f), orc.state_real(f), rtol=1e-12, err_msg=f)


# ------------------------------------------------------------------ init with the oracle's random stream (a20)
@pytest.mark.parametrize("dims", [(0, 0, 0), (6, 0, 0), (6, 0, 5), (4, 3, 5)])
def test_init_replay(dims):
    nx, ny, nz = dims
    kw = dict(sedi_switch=False) if nz == 0 else {}
    oi = h.box_opts(nx, ny, nz, 32, dx=25., **kw)
    if nx == 0:
        oi.dx = oi.dy = oi.dz = 1.
        oi.x1 = oi.y1 = oi.z1 = 1.
    fields = h.box_fields(oi, supersat=False)
    orc, hip = h.make_pair(oi, fields, force_state=False)
    assert hip.n_part == orc.n_part == 32 * max(nx, 1) * max(ny, 1) * max(nz, 1)
    exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
    exact(hip.state_u64("count_num"), orc.state_u64("count_num"), "count_num")
    exact(hip.state_u64("count_ijk"), orc.state_u64("count_ijk"), "count_ijk")
    # the dry volumes and the multiplicities: bit for bit (round 4: with a dry spectrum given as a function, which the host evaluates as
    # in the reference, the host also takes rd3 = exp(3 ln rd) and ln rd back from it -- init_dry_sd_conc.ipp:26-34, init_n.ipp:48-143)
    exact(hip.get_attr("rd3"), orc.get_attr("rd3"), "rd3")
    exact(hip.state_u64("n"), orc.state_u64("n"), "n")
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-4)   # TOMS748 tolerance 2^-15
    rel = np.abs(hip.get_attr("rw2") / orc.get_attr("rw2") - 1)
    assert np.median(rel) < 1e-14      # pow/exp/cbrt differ from glibc by an ulp; the root finder rarely takes another path
    for a, present in (("x", nx), ("y", ny), ("z", nz)):
        if present:
            np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14)
    np.testing.assert_allclose(hip.state_real("vt"), orc.state_real("vt"), rtol=1e-4)


# ------------------------------------------------------------------ sort (a2-a4)
def test_sort_and_count_exact():
    oi = h.box_opts(7, 5, 6, 40)
    orc, hip = h.make_pair(oi, h.box_fields(oi))
    # scramble cells: advect a few steps without anything else
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = opts.sedi = False
    for _ in range(3):
        step_pair(orc, hip, opts, h.box_fields(oi))
    for nm in ("ijk", "sorted_id", "sorted_ijk", "count_ijk", "count_num"):
        exact(hip.state_u64(nm), orc.state_u64(nm), nm)
    # sortedness properties
    sid, sijk, ijk = hip.state_u64("sorted_id"), hip.state_u64("sorted_ijk"), hip.state_u64("ijk")
    assert np.all(np.diff(sijk.astype(np.int64)) >= 0)
    exact(ijk[sid], sijk, "ijk[sorted_id]")
    same = np.diff(sijk.astype(np.int64)) == 0
    assert np.all(np.diff(sid.astype(np.int64))[same] > 0)      # stable: ids ascend inside a cell


def test_shuffle_sort_replay_exact():
    oi = h.box_opts(5, 4, 3, 70)            # > 64 SD per cell: segments span more than one wave
    orc, hip = h.make_pair(oi, h.box_fields(oi))
    n = orc.n_part
    (un,) = h.oracle_rng_preview(orc, [(1, n)])
    hip.rng_replay_push(1, un)
    orc.stage("hskpng_shuffle_and_sort")
    hip.stage("hskpng_shuffle_and_sort")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "shuffled sorted_id")
    exact(hip.state_u64("sorted_ijk"), orc.state_u64("sorted_ijk"), "sorted_ijk")


@pytest.mark.parametrize("sd_conc", [300, 700, 1500])
def test_mid_segment_sort(sd_conc):
    """cells of 257..2048 SDs take the per-cell LDS bitonic network (k_cellsort_lds) in both the plain and the
    shuffled order; a few advection steps make the cell populations ragged"""
    oi = h.box_opts(3, 2, 2, sd_conc)
    orc, hip = h.make_pair(oi, h.box_fields(oi))
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = opts.sedi = False
    for _ in range(2):
        step_pair(orc, hip, opts, h.box_fields(oi))
    for nm in ("ijk", "sorted_id", "sorted_ijk", "count_ijk", "count_num"):
        exact(hip.state_u64(nm), orc.state_u64(nm), nm)
    (un,) = h.oracle_rng_preview(orc, [(1, orc.n_part)])
    hip.rng_replay_push(1, un)
    orc.stage("hskpng_shuffle_and_sort")
    hip.stage("hskpng_shuffle_and_sort")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "shuffled sorted_id")


def test_big_segment_sort_0d():
    """one cell with 5000 SDs: exercises the > LDS-segment path (bitonic in global scratch)"""
    oi = h.box_opts(0, 0, 0, 5000, sedi_switch=False)
    oi.dx = oi.dy = oi.dz = 1.
    oi.x1 = oi.y1 = oi.z1 = 1.
    orc, hip = h.make_pair(oi, h.box_fields(oi, supersat=False))
    (un,) = h.oracle_rng_preview(orc, [(1, orc.n_part)])
    hip.rng_replay_push(1, un)
    orc.stage("hskpng_shuffle_and_sort")
    hip.stage("hskpng_shuffle_and_sort")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "0-D shuffled sorted_id")


# ------------------------------------------------------------------ terminal velocity (a6)
@pytest.mark.parametrize("vt", [lgrngn.vt_t.beard76, lgrngn.vt_t.beard77, lgrngn.vt_t.beard77fast,
                                lgrngn.vt_t.khvorostyanov_spherical, lgrngn.vt_t.khvorostyanov_nonspherical])
def test_vterm(vt):
    oi = h.box_opts(4, 3, 5, 32, terminal_velocity=vt)
    orc, hip = h.make_pair(oi, h.box_fields(oi))
    # spread radii over all regimes of the formulas (0.5 um .. 3 mm)
    n = orc.n_part
    rw2 = np.exp(np.linspace(np.log(0.5e-6), np.log(3e-3), n)) ** 2
    g = lambda nm: orc.state_real(nm)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    orc.stage("hskpng_vterm_all")
    hip.stage("hskpng_vterm_all")
    vo, vh = orc.state_real("vt"), hip.state_real("vt")
    np.testing.assert_allclose(vh, vo, rtol=1e-12)               # SURVEY 8a's bar (measured: <= 2.3e-13 for every formula, both arithmetic modes)
    if vt == lgrngn.vt_t.beard77fast:
        np.testing.assert_allclose(hip.state_real("vt_0"), orc.state_real("vt_0"), rtol=1e-12)


# ------------------------------------------------------------------ condensation (a7-a10)
def _arith(mode):
    """the three arithmetic modes of condensation as (opts_init fields, held to the strict bars?): strict (the API default, IEEE order),
    fast (strict_fp = 0: the lean solver, the bench headline), toms (strict_fp = 0, cond_solver = 1: the reference's TOMS748 iterates on the
    fast growth-rate arithmetic -- the same answer as the reference's up to where an ulp moves a stopping decision: the STRICT bars)"""
    return dict(strict_fp=mode == "strict", cond_solver=int(mode == "toms")), mode != "fast"


ARITH = ["strict", "fast", "toms"]


@pytest.mark.parametrize("mode", ARITH)
@pytest.mark.parametrize("sstp", [1, 4])
def test_cond_step(sstp, mode):
    """strict_fp=False: the growth rate collected into one rational expression + FMA contraction (lcx_math.hpp,
    cond_fun_fast) -- held to the SAME bars as the IEEE-order form"""
    kw, strict_fp = _arith(mode)
    oi = h.box_opts(4, 4, 6, 64, sstp_cond=sstp, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    for it in range(3):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
        np.testing.assert_allclose(rh, ro, rtol=1e-4)
        # (this box is a stress case -- fresh aerosol activating at RH 1.01, a step moves th by 0.8 K: see _harness.cond_bars)
        th_tol, rv_tol, med_tol = h.cond_bars(strict_fp)
        np.testing.assert_allclose(thh, tho, rtol=th_tol)
        np.testing.assert_allclose(rvh, rvo, rtol=rv_tol)
        h.copy_state(orc, hip)      # keep later iterations comparable one step at a time
    assert np.median(np.abs(rh / ro - 1)) < (med_tol if strict_fp else med_tol * sstp)    # (strict: ulp-level differences of the moment sums feed back through th / rv)


@pytest.mark.parametrize("mode", ARITH)
def test_cond_step_with_drizzle_and_rain_drops(mode):
    """the ventilated branch of the growth rate (Re Sc above 2^-8: drops above ~8 um; Re > 1 with its Re^0.077: above ~40 um), which
    the fast form keeps out of its straight-line path (cube roots without range check, pow as exp(y ln x)): wet radii from 5 um to
    1 mm with their terminal velocities, one and four substeps, against the oracle at the bars of test_cond_step"""
    kw, strict_fp = _arith(mode)
    oi = h.box_opts(4, 4, 6, 64, sstp_cond=1, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    g = orc.state_real
    rw2 = orc.get_attr("rw2")
    rw2[::2] = np.geomspace(5e-6, 1e-3, len(rw2[::2])) ** 2
    vt = np.minimum(1.2e8 * rw2, 9.)                              # (Stokes-like fall speeds up to 9 m/s: Re from 1e-4 to ~1000)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), vt, g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
        assert (orc.state_real("vt") > 0.3).sum() > 100           # drops with Re > 1 are there
        np.testing.assert_allclose(rh, ro, rtol=1e-4)
        # millimetre drops at the multiplicities of aerosol particles hold far more water than the vapour (10 kg per kg of air in a
        # cell): the fast arithmetic's root and the reference's bracket midpoint (_harness.cond_bars), 1e-5 apart in rw2, show in a
        # cell's rv at 1e-4
        # (toms: the reference's iterates in fast arithmetic -- where an ulp moves TOMS748's stopping decision for ONE such drop, the
        # midpoint of its last bracket moves by up to 1.5e-5 and the cell's rv by 1.4e-6 (1 cell of 96): 3e-6 here, 1e-6 in test_cond_step)
        np.testing.assert_allclose(thh, tho, rtol=1e-7 if mode == "strict" else 2e-7 if mode == "toms" else 2e-6)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-6 if mode == "strict" else 3e-6 if mode == "toms" else 2e-4)
        big = ro > (8e-6) ** 2
        assert np.median(np.abs(rh[big] / ro[big] - 1)) < h.cond_bars(strict_fp)[2]
        h.copy_state(orc, hip)


@pytest.mark.parametrize("mode", ARITH)
def test_cond_step_with_invalid_terminal_velocities(mode):
    """A droplet that coalesced in the previous step_async carries the reference's flag vt = -1 through the next condensation (the
    last coalescence substep is not followed by hskpng_vterm_invalid, particles_step.ipp:386-392): its Reynolds number is negative,
    the ventilation factors' 1 + Re Sc goes below 1 and, for drops above ~15 um, below zero.  Round 2's fast form sent such
    droplets through the small-argument series of the cube root (5 of 2.1e6 droplets per step off by up to 20 % on C5): every
    third droplet is flagged here, wet radii from 1 to 300 um, all arithmetic modes at the bars of test_cond_step"""
    kw, strict_fp = _arith(mode)
    oi = h.box_opts(4, 4, 6, 64, sstp_cond=1, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    g = orc.state_real
    rw2 = orc.get_attr("rw2")
    rw2[::3] = np.geomspace(1e-6, 3e-4, len(rw2[::3])) ** 2
    vt = g("vt")
    vt[::3] = -1.
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), vt, g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
    ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
    assert np.all(np.isfinite(rh))
    np.testing.assert_allclose(rh, ro, rtol=1e-4)
    assert np.median(np.abs(rh[::3] / ro[::3] - 1)) < h.cond_bars(strict_fp)[2]
    np.testing.assert_allclose(thh, tho, rtol=1e-7 if mode == "strict" else 2e-7 if mode == "toms" else 2e-6)      # (drops of 0.3 mm at aerosol multiplicities, as in the test above)
    np.testing.assert_allclose(rvh, rvo, rtol=1e-6 if mode == "strict" else 3e-6 if mode == "toms" else 2e-4)


@pytest.mark.parametrize("mode", ARITH)
def test_cond_step_with_near_dry_particles(mode):
    """Particles of almost no hygroscopicity (kappa = 1e-10, the set-up of the reference's coalescence tests and of bench.py's coal-stress
    leg) a relative 1e-9 ... 1e-3 above their dry radius: in subsaturated cells the bracket's lower end is clamped by the dry radius and
    often already inside the root finder's tolerance -- the reference evaluates f at the dry radius (zero water activity: the particle
    would grow there), sees the sign change and returns the bracket's midpoint (toms748.hpp:305-313); the lean solver answers that
    midpoint without the evaluation since round 5 (lcx_math.hpp lean2_head).  In supersaturated cells the same particles take the
    explicit-Euler jump of a bracket without a sign change.  All arithmetic modes against the oracle, two steps"""
    kw, strict_fp = _arith(mode)
    oi = h.box_opts(4, 4, 6, 64, sstp_cond=1, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    g = orc.state_real
    rd3 = g("rd3")
    rng = np.random.default_rng(11)
    rw2 = np.cbrt(rd3) ** 2 * (1. + 10. ** rng.uniform(-9, -3, size=rd3.shape))
    rw2[::16] = np.cbrt(rd3[::16]) ** 2                           # (and some exactly on it)
    args = (orc.state_u64("n"), rd3, rw2, np.full_like(rd3, 1e-10), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
        np.testing.assert_allclose(rh, ro, rtol=1e-4)
        if it == 0:
            stay = np.abs(ro / rw2 - 1.) < 2. ** -15
            assert .02 < stay.mean() < .95, stay.mean()            # both kinds of cell are in the box (most of it is supersaturated)
        np.testing.assert_allclose(thh, tho, rtol=2e-6)
        np.testing.assert_allclose(rvh, rvo, rtol=2e-4)
        h.copy_state(orc, hip)


@pytest.mark.parametrize("strict_fp", [True, False])
@pytest.mark.parametrize("mode", ["nomix", "mix", "adaptive", "adaptive_act"])
def test_perparticle_cond_step(mode, strict_fp):
    """per-particle substepping (exact_sstp_cond) on a 3-D box against the oracle, advection on so that droplets carry
    their private (rv, th, rhod) across cells; rc2 invalidation through coalescence in the adaptive_act variant"""
    kw = dict(sstp_cond=4, exact_sstp_cond=True, strict_fp=strict_fp)
    if mode != "mix":
        kw["sstp_cond_mix"] = False
    if mode.startswith("adaptive"):
        kw.update(adaptive_sstp_cond=True, sstp_cond_adapt_drw2_eps=1e-3, sstp_cond_adapt_drw2_max=2.)
    if mode == "adaptive_act":
        kw["sstp_cond_act"] = 8
    oi = h.box_opts(4, 3, 5, 48, **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    opts.coal = mode == "adaptive_act"
    opts.sedi = False
    for nm in ("sstp_tmp_rv", "sstp_tmp_th", "sstp_tmp_rh"):
        np.testing.assert_allclose(hip.state_real(nm), orc.state_real(nm), rtol=1e-12, err_msg=nm)
    if mode == "adaptive_act":       # rc2 is itself the result of a tolerance-terminated root search (2^-16 in rw3)
        np.testing.assert_allclose(hip.state_real("rc2"), orc.state_real("rc2"), rtol=1e-5)
    for it in range(3):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
        np.testing.assert_allclose(rh, ro, rtol=2e-4)
        assert np.median(np.abs(rh / ro - 1)) < 2e-9      # no state copy between the steps here: 12 substeps of drift
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-7)
        for nm in ("sstp_tmp_rv", "sstp_tmp_th", "sstp_tmp_rh"):
            np.testing.assert_allclose(hip.state_real(nm), orc.state_real(nm), rtol=1e-6, err_msg=nm)
        if mode == "adaptive_act":
            np.testing.assert_allclose(hip.state_real("rc2"), orc.state_real("rc2"), rtol=1e-5)


def test_cond_moment_feedback_conservation():
    """size-independent property: d(rv) summed over cells == -4/3 pi rho_w d(sum n rw^3)/(dv rhod)"""
    oi = h.box_opts(6, 5, 4, 64)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False

    def m3():
        hip.diag_all()
        hip.diag_wet_mom(3)
        return hip.outbuf_array().reshape(rv.shape)
    before = m3()
    thh, rvh = th.copy(), rv.copy()
    hip.step_sync(opts, thh, rvh, rhod, **C)
    hip.step_async(opts)
    after = m3()
    drv = -(after - before) * 4. / 3 * np.pi * 1e3
    np.testing.assert_allclose(rvh - rv, drv, rtol=1e-9, atol=1e-16)


# ------------------------------------------------------------------ coalescence (a11, a12)
@pytest.mark.parametrize("kernel,params", [(lgrngn.kernel_t.geometric, []), (lgrngn.kernel_t.geometric, [0.5]),
                                           (lgrngn.kernel_t.golovin, [1500.]), (lgrngn.kernel_t.Long, []),
                                           (lgrngn.kernel_t.hall, []), (lgrngn.kernel_t.hall_davis_no_waals, []),
                                           (lgrngn.kernel_t.vohl_davis_no_waals, []), (lgrngn.kernel_t.hall_pinsky_cumulonimbus, [])])
def test_coal_replay(kernel, params):
    oi = h.box_opts(3, 3, 3, 96, kernel=kernel, kernel_parameters=np.array(params), dx=1.)
    oi.dt = 30.      # long step + small cells -> many collisions
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    # rain-sized droplets so that the collision probability is appreciable
    n = orc.n_part
    rng = np.random.default_rng(1)
    rw2 = (10 ** rng.uniform(-5.3, -3.5, n)) ** 2
    g = lambda nm: orc.state_real(nm)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    for st in ("hskpng_Tpr", "hskpng_vterm_all"):
        orc.stage(st)
        hip.stage(st)
    np.testing.assert_allclose(hip.state_real("vt"), orc.state_real("vt"), rtol=1e-12)
    hip.set_particles(orc.state_u64("n"), g("rd3"), g("rw2"), g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    h.push_coal_replay(orc, hip)
    orc.stage("coal")
    hip.stage("coal")
    exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
    n_o, n_h = orc.state_u64("n"), hip.state_u64("n")
    assert np.sum(n_o != args[0]) > 10, "test needs collisions to happen"
    exact(n_h, n_o, "multiplicities after coalescence")
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-14)
    np.testing.assert_allclose(hip.get_attr("rd3"), orc.get_attr("rd3"), rtol=1e-14)
    exact(hip.state_real("vt") == -1, orc.state_real("vt") == -1, "invalidated vt flags")
    np.testing.assert_allclose(hip.state_real("col")[:-1], orc.state_real("col")[:-1], rtol=0, atol=0)


@pytest.mark.parametrize("turb_coal", [False, True])
@pytest.mark.parametrize("kernel", [lgrngn.kernel_t.onishi_hall, lgrngn.kernel_t.onishi_hall_davis_no_waals])
def test_onishi_turbulent_kernel_replay(kernel, turb_coal):
    """f4: Onishi turbulent kernel (kernels.hpp:209-250, kernel_onishi_nograv.hpp, wang_collision_enhancement.hpp) through
    full steps with the cell field diss_rate; opts.turb_coal off = dissipation rate 0 (coal.ipp:392-451)"""
    oi = h.box_opts(3, 3, 3, 96, kernel=kernel, kernel_parameters=np.array([66.]), dx=1., turb_coal_switch=True)
    oi.dt = 30.
    th, rv, rhod, C = h.box_fields(oi)
    orc, hip = h.make_pair(oi, (th, rv, rhod, C))
    n = orc.n_part
    rng = np.random.default_rng(2)
    rw2 = (10 ** rng.uniform(-5.3, -3.7, n)) ** 2
    g = lambda nm: orc.state_real(nm)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    diss = rng.uniform(1e-3, 0.1, th.shape)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = False
    opts.coal = True
    opts.turb_coal = turb_coal
    for pr in (orc, hip):
        pr.step_sync(opts, th.copy(), rv.copy(), rhod.copy(), diss_rate=diss, **C)
    h.push_coal_replay(orc, hip)
    orc.step_async(opts)
    hip.step_async(opts)
    col = orc.state_real("col")[:-1]
    assert np.all(np.isfinite(col))
    n_o, n_h = orc.state_u64("n"), hip.state_u64("n")
    assert np.sum(n_o != args[0]) > 10, "test needs collisions to happen"
    exact(n_h, n_o, "multiplicities after coalescence")
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-14)
    np.testing.assert_allclose(hip.state_real("col")[:-1], col, rtol=0, atol=0)


def test_coal_two_kappas_replay():
    """kappa mixing (weighted_summator) with two aerosol modes"""
    oi = h.box_opts(0, 0, 0, 2048, sedi_switch=False)
    oi.dx = oi.dy = oi.dz = 1.
    oi.x1 = oi.y1 = oi.z1 = 1.
    oi.dt = 100.
    f = h.lognormal_fn(20e-6, 1.5, 3e7)
    oi.dry_distros = {(.1, 0.): f, (.9, 0.): f}
    oi.n_sd_max = 2048
    th, rv, rhod = np.array([300.]), np.array([.01]), np.array([1.])
    orc, hip = h.oracle_particles(oi), h.hip_particles(oi)
    for arr in h.oracle_rng_preview(orc, [(0, 1024)] * 2):
        hip.rng_replay_push(0, arr)
    orc.init(th, rv, rhod)
    hip.init(th, rv, rhod)
    h.copy_state(orc, hip)
    for st in ("hskpng_Tpr", "hskpng_vterm_all"):
        orc.stage(st)
        hip.stage(st)
    h.copy_state(orc, hip)
    h.push_coal_replay(orc, hip)
    orc.stage("coal")
    hip.stage("coal")
    exact(hip.state_u64("n"), orc.state_u64("n"), "n")
    assert np.sum(orc.state_real("col")[:-1] > 0) > 10
    np.testing.assert_allclose(hip.get_attr("kappa"), orc.get_attr("kappa"), rtol=1e-13)


# ------------------------------------------------------------------ advection / sedimentation / boundary (a13-a16)
@pytest.mark.parametrize("dims", [(7, 0, 0), (6, 0, 5), (5, 4, 6)])
def test_pred_corr_falls_back_to_euler_beyond_the_halo(dims):
    """particles_step.ipp:127-142: a Courant number outside [-2, 2] makes the step first order (Courant arrays keep their
    2-plane halo, read with the halo offset); the next step with tame Courant numbers is predictor-corrector again"""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 24, adve_scheme=lgrngn.as_t.pred_corr, dx=30., sedi_switch=nz > 0)
    th, rv, rhod, C = h.box_fields(oi)
    wild = {k: v.copy() for k, v in C.items()}
    wild["Cx"].flat[3] = 2.5
    orc, hip = h.make_pair(oi, (th, rv, rhod, C))
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = opts.sedi = False
    for fields in ((th, rv, rhod, wild), (th, rv, rhod, C), (th, rv, rhod, C)):
        step_pair(orc, hip, opts, fields)
        assert hip.n_part == orc.n_part
        for a in ("x", "y", "z"):
            if getattr(oi, "n" + a):
                np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14, atol=1e-9, err_msg=a)
        for nm in ("ijk", "sorted_id"):
            exact(hip.state_u64(nm), orc.state_u64(nm), nm)


# ------------------------------------------------------------------ advection / sedimentation / boundary (a13-a16)
@pytest.mark.parametrize("scheme", [lgrngn.as_t.euler, lgrngn.as_t.implicit, lgrngn.as_t.pred_corr])
@pytest.mark.parametrize("dims", [(6, 0, 5), (5, 4, 6)])
def test_move_and_post_copy(scheme, dims):
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 48, adve_scheme=scheme, dx=30.)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    # make a good fraction of SDs rain-sized so that they fall through the bottom
    n = orc.n_part
    rw2 = orc.get_attr("rw2")
    rw2[::3] = (1.5e-3) ** 2
    g = lambda nm: orc.state_real(nm)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y") if ny else None, g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.cond = opts.coal = False
    for it in range(4):
        step_pair(orc, hip, opts, fields)
        assert hip.n_part == orc.n_part
        for a in ("x", "y", "z"):
            if getattr(oi, "n" + a):
                np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14, atol=1e-9, err_msg=a)
        for nm in ("ijk", "sorted_id", "count_num", "n"):
            exact(hip.state_u64(nm), orc.state_u64(nm), nm)
        h.copy_state(orc, hip)
    assert orc.n_part < n, "test needs precipitation to happen"
    po, ph = orc.diag_puddle(), hip.diag_puddle()
    for k in po:
        np.testing.assert_allclose(ph[k], po[k], rtol=1e-12, err_msg=k)


@pytest.mark.parametrize("eager", [False, True])
def test_lazy_compaction_is_unobservable(eager, monkeypatch):
    """Dead SDs stay in storage until a compaction is due (post_copy in lcx_core.hip); nothing observable may depend on
    that: run several precipitating steps WITHOUT touching any getter, then compare everything with the oracle."""
    oi = h.box_opts(5, 4, 6, 48, dx=30., coal_switch=False)
    oi.dbg_flags = int(lgrngn.dbg.EAGER_COMPACT) if eager else 0
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    n0 = orc.n_part
    rw2 = orc.get_attr("rw2")
    rw2[::5] = (1.2e-3) ** 2
    g = lambda nm: orc.state_real(nm)
    args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.coal = opts.cond = False         # (mm-sized drops with aerosol multiplicities are not a state to condense on)
    th, rv, rhod, C = fields
    tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
    for it in range(6):
        orc.step_sync(opts, tho, rvo, rhod, **C)
        orc.step_async(opts)
        hip.step_sync(opts, thh, rvh, rhod, **C)
        hip.step_async(opts)
        assert hip.n_part == orc.n_part          # n_part() itself must already report living SDs only
    assert orc.n_part < n0
    # diagnostics walk the sorted order: they must not see dead SDs either (still no storage getter called)
    for prt in (orc, hip):
        prt.diag_all()
        prt.diag_sd_conc()
    exact(hip.outbuf_array(), orc.outbuf_array(), "sd_conc")
    for nm in ("n", "ijk", "sorted_id", "count_num"):
        exact(hip.state_u64(nm), orc.state_u64(nm), nm)
    for a in ("x", "y", "z"):
        np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-13, atol=1e-9)
    exact(hip.get_attr("rw2"), orc.get_attr("rw2"), "rw2")
    po, ph = orc.diag_puddle(), hip.diag_puddle()
    np.testing.assert_allclose(ph["particle_number"], po["particle_number"], rtol=1e-12)


def test_advection_shifts_by_one_cell():
    """tests/python/unit/lgrngn_adve.py:97-105: C = +-1 moves the sd_conc field by exactly one cell"""
    for Cx, roll in ((1., -1), (-1., 1)):
        oi = lgrngn.opts_init_t()
        oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
        oi.coal_switch = oi.sedi_switch = False
        oi.dt = 1
        oi.nz, oi.nx, oi.dz, oi.dx = 5, 6, 1, 1
        oi.z1, oi.x1 = oi.nz * oi.dz, oi.nx * oi.dx
        oi.sd_conc = 10
        oi.n_sd_max = 10 * oi.nx * oi.nz
        opts = lgrngn.opts_t()
        opts.sedi = opts.cond = opts.coal = False
        rhod, th, rv = 1. * np.ones((oi.nx, oi.nz)), 300. * np.ones((oi.nx, oi.nz)), 0.01 * np.ones((oi.nx, oi.nz))
        pr = h.hip_particles(oi)
        pr.init(th, rv, rhod, Cx=Cx * np.ones((oi.nx + 1, oi.nz)), Cz=np.zeros((oi.nx, oi.nz + 1)))
        pr.step_sync(opts, th, rv, rhod)
        pr.diag_all()
        pr.diag_sd_conc()
        tab_in = np.frombuffer(pr.outbuf()).reshape(oi.nx, oi.nz).copy()
        pr.step_async(opts)
        pr.step_sync(opts, th, rv, rhod)
        pr.diag_all()
        pr.diag_sd_conc()
        tab_out = np.frombuffer(pr.outbuf()).reshape(oi.nx, oi.nz).copy()
        assert (tab_in == np.roll(tab_out, roll, 0)).all()
        assert tab_in.sum() == oi.sd_conc * oi.nx * oi.nz


# ------------------------------------------------------------------ full steps, everything on, replayed stream
@pytest.mark.parametrize("mode", ARITH)
@pytest.mark.parametrize("dims,sstp", [((4, 4, 4), (1, 1)), ((8, 0, 8), (3, 2))])
def test_full_steps_replay(dims, sstp, mode):
    nx, ny, nz = dims
    kw, strict_fp = _arith(mode)
    oi = h.box_opts(nx, ny, nz, 64, sstp_cond=sstp[0], sstp_coal=sstp[1], **kw)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(3):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        assert hip.n_part == orc.n_part
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(thh, tho, rtol=h.cond_bars(strict_fp)[0])
        np.testing.assert_allclose(rvh, rvo, rtol=h.cond_bars(strict_fp)[1])
        np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-4 if strict_fp else 1e-4 * sstp[0])
        h.copy_state(orc, hip)


def test_the_api_default_mode_keeps_the_strict_bars():
    """What a caller that sets NO arithmetic option gets since round 5 (both mirrors, lcx_opts_init_default): fast arithmetic with the
    reference's TOMS748 iterates.  The suite pins the parity mode for every other test (tests/_harness.py); this one takes the
    constructor's own values and holds three replayed full steps to the bars of the strict mode."""
    oi = h.api_default_opts(h.box_opts(5, 4, 6, 64, sstp_cond=2))
    assert (oi.strict_fp, oi.cond_solver) == (False, 1)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(3):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        h.assert_mode(hip, False, 1, ("fold_toms748", "lean_toms748_sorted", "substeps"))      # (the default's kernels; two substeps: both in one launch)
        assert hip.n_part == orc.n_part
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        np.testing.assert_allclose(thh, tho, rtol=h.cond_bars(True)[0])
        np.testing.assert_allclose(rvh, rvo, rtol=h.cond_bars(True)[1])
        err = np.abs(hip.get_attr("rw2") / orc.get_attr("rw2") - 1)
        assert err.max() < 1e-4 and np.median(err) < h.cond_bars(True)[2], (err.max(), np.median(err))
        h.copy_state(orc, hip)


def test_incloud_time_matches_oracle():
    """opts_init.diag_incloud_time (update_incloud_time.ipp:36-66, the selector after collisions coal.ipp:17-31,505-525,
    diag_incloud_time_mom particles_diag.ipp:482-490): the attribute after replayed full steps and its moments"""
    oi = h.box_opts(4, 3, 4, 64, diag_incloud_time=True, dx=2.)
    oi.dt = 2.
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(4):
        step_pair(orc, hip, opts, fields)
    to, th_ = orc.state_real("incloud_time"), hip.state_real("incloud_time")
    assert len(np.unique(to)) > 2 and to.max() == 3 * oi.dt          # nothing is activated at the first update (RH < 0.95 at init)
    # rw2 > rc2 compares two numbers that carry root-finder tolerances, and a collision test may flip for the same reason
    assert hip.n_part == orc.n_part
    assert np.mean(to != th_) < 1e-2
    for pr in (orc, hip):
        pr.diag_all(); pr.diag_incloud_time_mom(1)
    np.testing.assert_allclose(hip.outbuf_array(), orc.outbuf_array(), rtol=2e-2)
    plain = h.hip_particles(h.box_opts(4, 3, 4, 8))
    plain.init(*[f.copy() for f in fields[:3]], **fields[3])
    plain.diag_all()
    with pytest.raises(RuntimeError, match="diag_incloud_time==false"):
        plain.diag_incloud_time_mom(1)


# ------------------------------------------------------------------ diagnostics (a21)
def test_diagnostics_match_oracle():
    oi = h.box_opts(4, 3, 5, 40)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    opts.coal = False
    step_pair(orc, hip, opts, fields)
    h.copy_state(orc, hip)

    def both(fn):
        out = []
        for p in (orc, hip):
            fn(p)
            out.append(p.outbuf_array())
        return out
    checks = [
        lambda p: (p.diag_all(), p.diag_sd_conc()),
        lambda p: (p.diag_all(), p.diag_wet_mom(0)), lambda p: (p.diag_all(), p.diag_wet_mom(1)),
        lambda p: (p.diag_all(), p.diag_wet_mom(3)), lambda p: (p.diag_all(), p.diag_dry_mom(3)),
        lambda p: (p.diag_all(), p.diag_kappa_mom(1)),
        lambda p: (p.diag_wet_rng(.5e-6, 25e-6), p.diag_wet_mom(3)),
        lambda p: (p.diag_dry_rng(0, 5e-8), p.diag_wet_rng_cons(0, 1e-6), p.diag_sd_conc()),
        lambda p: (p.diag_kappa_rng(.5, 1.), p.diag_dry_mom(0)),
        lambda p: (p.diag_water(), p.diag_wet_mom(2)),
        lambda p: (p.diag_all(), p.diag_precip_rate()),
        lambda p: p.diag_max_rw(), lambda p: p.diag_RH(), lambda p: p.diag_temperature(), lambda p: p.diag_pressure(),
        lambda p: (p.diag_RH_ge_Sc(), p.diag_wet_mom(0)), lambda p: (p.diag_rw_ge_rc(), p.diag_wet_mom(0)),
        lambda p: (p.diag_rw_ge_rc(), p.diag_sd_conc()),
        lambda p: (p.diag_all(), p.diag_wet_mass_dens(8e-6, .62)), lambda p: (p.diag_wet_rng(1e-6, 1.), p.diag_wet_mass_dens(2e-6, .4)),
    ]
    for i, fn in enumerate(checks):
        o, g_ = both(fn)
        # #10 (precip rate) recomputes vt on both sides, #12-14 (RH, T, p) are functions of th/rv AFTER the condensation
        # feedback of the step above: exp/log/pow/cbrt differ by an ulp between glibc and the device
        # #15-17 select by the critical radius / supersaturation (exact counts unless an SD sits on the threshold), #18-19 kernel estimate
        np.testing.assert_allclose(g_, o, rtol=1e-8 if i in (10, 12, 13, 14, 18, 19) else 1e-11, atol=0, err_msg="diag #%d" % i)
        if i in (15, 16, 17):
            assert o.sum() > 0 and o.sum() < orc.n_part * 1e12, "selection must be neither empty nor everything"
    with pytest.raises(RuntimeError):
        fresh = h.hip_particles(oi)
        fresh.init(fields[0].copy(), fields[1].copy(), fields[2].copy(), **fields[3])
        fresh.diag_wet_mom(1)            # counting before selecting


# ------------------------------------------------------------------ API behaviour (api_lgrngn.py:123-133,166-170)
def test_call_order_exceptions():
    oi = h.box_opts(0, 0, 0, 64, sedi_switch=False)
    oi.dx = oi.dy = oi.dz = 1.
    oi.x1 = oi.y1 = oi.z1 = 1.
    th, rv, rhod = np.array([300.]), np.array([.01]), np.array([1.])
    pr = h.hip_particles(oi)
    opts = lgrngn.opts_t()
    opts.sedi = opts.adve = False
    with pytest.raises(RuntimeError):
        pr.step_sync(opts, th, rv, rhod)          # before init
    pr.init(th, rv, rhod)
    with pytest.raises(RuntimeError):
        pr.init(th, rv, rhod)                     # init twice
    with pytest.raises(RuntimeError):
        pr.step_async(opts)                       # async before sync
    pr.step_sync(opts, th, rv, rhod)
    with pytest.raises(RuntimeError):
        pr.step_sync(opts, th, rv, rhod)          # sync twice
    pr.step_async(opts)
    with pytest.raises(RuntimeError):
        pr.step_async(opts)                       # async twice
    pr.diag_all()
    pr.diag_sd_conc()
    assert np.frombuffer(pr.outbuf())[0] == 64


def test_float_build_runs_and_tracks_double():
    oi = h.box_opts(4, 4, 4, 64)
    th, rv, rhod, C = h.box_fields(oi)
    res = {}
    for rt in (np.float64, np.float32):
        pr = h.hip_particles(oi, rt)
        f = lambda a: np.ascontiguousarray(a, dtype=rt)
        tt, rr, dd = f(th), f(rv), f(rhod)
        CC = {k: f(v) for k, v in C.items()}
        pr.init(tt, rr, dd, **CC)
        opts = lgrngn.opts_t()
        for it in range(3):
            pr.step_sync(opts, tt, rr, dd, **CC)
            pr.step_async(opts)
        res[rt] = (tt.astype(np.float64), rr.astype(np.float64), pr.n_part)
    # real_t=float follows the reference's float configuration: the root finder's tolerance is 2^-7 there
    # (src/detail/config.hpp:39, sizeof(real_t)*8/4 bits), which biases the condensate by O(1%) -- so this is a
    # "same physics" check, not a precision claim (the float build against the oracle iterating to float's tolerance:
    # tests/test_hip_configs.py::test_c2_icicle_2d_float_against_the_double_oracle)
    np.testing.assert_allclose(res[np.float32][0], res[np.float64][0], rtol=5e-3)
    np.testing.assert_allclose(res[np.float32][1], res[np.float64][1], rtol=0.15)
    assert res[np.float32][2] == res[np.float64][2]


def test_fast_math_accuracy():
    """the seeded cbrt / reduced exp of the fast-mode growth rate (csrc/lcx_math.hpp) against numpy in long double:
    <= 1 ulp on their domains, library fallback outside"""
    import ctypes
    from libcloudphxx_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)

    def probe(which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        rc = lib.lcx_math_probe(ctypes.c_int(which), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_size_t(x.size))
        assert rc == 0
        return y

    # cbrt: arguments 1 + Re*Sc span [1, ~1e4] in practice; test the whole seeded domain and the fallback
    x = np.concatenate([1 + 10 ** rng.uniform(-12, 5, 200000), 10 ** rng.uniform(-0.9, 29.9, 100000), [1., 8., 27., 0.125]])
    ref = np.cbrt(x.astype(np.longdouble))
    for which in (0, 2):
        y = probe(which, x)
        ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
        assert float(ulp.max()) <= 1.0, (which, float(ulp.max()))
    xf = np.array([0., -8., 1e-40, 1e300, 0.1, -1e-3, np.inf])
    np.testing.assert_array_equal(probe(0, xf), probe(2, xf))
    # exp: Kelvin term A / r_w in (0, ~2); test [-50, 50] and the fallback
    x = np.concatenate([10 ** rng.uniform(-12, 0.5, 200000), rng.uniform(-50, 50, 100000), [0., 1., -1.]])
    ref = np.exp(x.astype(np.longdouble))
    for which in (1, 3):
        y = probe(which, x)
        ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
        assert float(ulp.max()) <= 1.0, (which, float(ulp.max()))
    xf = np.array([800., -800., 710., -745., np.inf, -np.inf])
    np.testing.assert_array_equal(probe(1, xf), probe(3, xf))
    # the production kernel's form of the same exponential (two interleaved chains, no range check): Kelvin term A / r_w in (0, ~2)
    x = np.concatenate([10 ** rng.uniform(-12, 0.5, 200000), rng.uniform(-50, 50, 100000), [0., 1., -1.]])
    ref = np.exp(x.astype(np.longdouble))
    y = probe(8, x)
    ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
    assert float(ulp.max()) <= 1.0, float(ulp.max())
    # reciprocal of the fast-mode root finder's divisions (differences of function values and abscissae, ~1e-25 ... 1e5)
    x = np.concatenate([10 ** rng.uniform(-40, 40, 200000), -10 ** rng.uniform(-40, 40, 100000), [1., 3., -7., 1e-300, 1e300]])
    ref = 1 / x.astype(np.longdouble)
    y = probe(4, x)
    ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
    assert float(ulp.max()) <= 1.0, float(ulp.max())
    # the same with ONE Newton step (k_cond_fast's root finder: only the interpolated abscissae c = a - f ... / ... are formed with
    # it, the function values at them are computed in full): the hardware reciprocal is good to ~2^-25, one step squares that
    y = probe(6, x)
    ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
    assert float(ulp.max()) <= 16.0, float(ulp.max())
    # cbrt(1 + x) of the ventilation factors: series below 2^-8, seeded cube root above
    x = np.concatenate([10 ** rng.uniform(-14, -2.4, 200000), 10 ** rng.uniform(-2.5, 4, 100000), [0., 2. ** -8, np.nextafter(2. ** -8, 0)]])
    ref = np.cbrt(1 + x.astype(np.longdouble))
    y = probe(7, x)
    ulp = np.abs((y.astype(np.longdouble) - ref) / np.spacing(ref.astype(np.float64)))
    assert float(ulp.max()) <= 1.0, float(ulp.max())
    # lean logarithm of the fast-mode terminal-velocity pass (argument: wet radius squared, 1e-20 ... 1e-4; test far beyond)
    x = np.concatenate([10 ** rng.uniform(-30, 10, 300000), 1 + rng.uniform(-1e-3, 1e-3, 1000), [1., 2., .5, np.e, 1e-300, 1e300]])
    ref = np.log(x.astype(np.longdouble))
    y = probe(5, x)
    err = np.abs(y.astype(np.longdouble) - ref)
    assert float((err / np.maximum(np.spacing(np.abs(ref).astype(np.float64)), 2.3e-16)).max()) <= 2.0


def test_production_cond_kernel_against_the_plain_fast_form(monkeypatch):
    """k_cond_fast (per-cell set-up hoisted into k_cond_cellpre, tuned root-finder arithmetic, one scratch value per droplet)
    against the plain fast form that evaluates everything per droplet (opts_init.dbg_flags & NO_COND_PRE selects it)"""
    oi = h.box_opts(4, 3, 5, 64, sstp_cond=2, strict_fp=False)
    fields = h.box_fields(oi)
    oi.cond_solver = 1                                 # (round 2's fast kernels: TOMS748 iterates in fast arithmetic)
    res = []
    for off in (False, True):
        oi.dbg_flags = int(lgrngn.dbg.NO_COND_PRE if off else lgrngn.dbg.COND_TOMS_TWO_PASS)
        orc, hip = h.make_pair(oi, fields)
        opts = lgrngn.opts_t()
        opts.coal = opts.adve = opts.sedi = False
        th, rv, rhod, C = fields
        thh, rvh = th.copy(), rv.copy()
        hip.step_sync(opts, thh, rvh, rhod, **C)
        res.append((hip.get_attr("rw2"), thh, rvh))
    # the production kernel (k_cond_fast) also takes the root finder's reciprocals with one Newton step and the ventilation
    # factors' cube roots by series, and sums the droplets' changes of n rw^3 instead of the before / after pair: same
    # iterates to a few ulp, not the same bits
    err = np.abs(res[0][0] / res[1][0] - 1)
    assert err.max() < 2e-4 and np.quantile(err, .99) < 1e-6 and np.median(err) < 1e-12, (err.max(), np.quantile(err, .99), np.median(err))
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-10)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-8)


def test_lean_solver_against_toms748_in_fast_arithmetic(monkeypatch):
    """The fast arithmetic's bracketed secant (k_cond_lean) against TOMS748 on the SAME growth-rate arithmetic (opts_init.cond_solver = 1: round 2's
    kernels), 2^20 droplets, two steps: both solve rw2_new = rw2_old + dt f(rw2_new) on the reference's bracket to 2^-15 -- every
    droplet's answers within that tolerance of each other (no droplet on another root), th and rv to 1e-9"""
    oi = h.box_opts(32, 16, 32, 64, strict_fp=False)
    fields = h.box_fields(oi)
    res = []
    for toms in (False, True):
        oi.cond_solver = int(toms)
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = opts.adve = opts.sedi = False
        thh, rvh = th.copy(), rv.copy()
        for _ in range(2):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        res.append((hip.get_attr("rw2"), thh, rvh))
    err = np.abs(res[0][0] / res[1][0] - 1)
    assert err.max() < 1e-4 and np.median(err) < 3e-6, (err.max(), np.median(err), int((err > 3e-5).sum()))
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=2e-8)      # (measured 4.4e-9: the spin-up steps of fresh aerosol, see _harness.cond_bars)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=2e-7)


@pytest.mark.parametrize("budget", [6, 3, 1])
def test_two_pass_condensation_is_bit_identical_to_one_pass(monkeypatch, budget):
    """k_cond_fast with a short iteration budget + the dense second launch over the droplets that ran out of it (the production
    form from 2^25 super-droplets upwards; opts_init.dbg_cond_budget forces it here) does the same arithmetic per droplet as the single pass:
    identical bits in rw2, th and rv.  Budget 6 defers the far tail, 3 and 1 defer most of the droplets that iterate at all."""
    oi = h.box_opts(16, 8, 8, 64, sstp_cond=2, strict_fp=False)
    fields = h.box_fields(oi)
    oi.cond_solver = 1
    oi.dbg_flags = int(lgrngn.dbg.COND_TOMS_TWO_PASS)
    res = []
    for b in (-1, budget):
        oi.dbg_cond_budget = b
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = opts.adve = opts.sedi = False
        thh, rvh = th.copy(), rv.copy()
        for _ in range(2):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        res.append((hip.get_attr("rw2"), thh, rvh))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.abs(res[0][2] - rv).max() > 0                       # (condensation did happen)


def test_toms748_in_the_storage_order_kernel_gives_round_2s_bits():
    """opts_init.cond_solver = 1 since round 4: k_cond_lean<.., SOLVER = 2> -- TOMS748 on the growth rate of round 2's kernels, in the
    storage-order walk that carries the re-sort's scatter -- against round 2's pair of launches (dbg_flags & COND_TOMS_TWO_PASS): the same
    arithmetic per droplet, identical rw2; th and rv to rounding (the change of n rw^3 is formed with rsqrt here, with sqrt there, and summed
    in fixed point here).  A full step without condensation first, so that the droplets have moved and the deferred re-sort is what the
    kernel carries, and both forms see the same input bits."""
    oi = h.box_opts(16, 8, 8, 64, strict_fp=False, cond_solver=1)       # (one substep: the second would start from the first one's rounding)
    fields = h.box_fields(oi)
    res = []
    # round 5: the default is the kernel FOLDED behind TOMS748's head (k_cond_lean_fold<.., SOLVER = 2>); COND_NO_FOLD the plain
    # storage-order kernel; a stage of 8 slots (dbg_cond_budget) leaves most of the droplets that enter the loop in their own lanes
    for flags, budget in ((0, 0), (int(lgrngn.dbg.COND_TOMS_TWO_PASS), 0), (int(lgrngn.dbg.COND_NO_FOLD), 0), (0, 8), (int(lgrngn.dbg.KPA_ARRAY), 0)):
        oi.dbg_flags = flags
        oi.dbg_cond_budget = budget
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        opts.cond = False
        hip.step_sync(opts, thh, rvh, rhod, **C)
        hip.step_async(opts)
        opts.cond = True
        hip.step_sync(opts, thh, rvh, rhod, **C)
        res.append((hip.get_attr("rw2"), thh, rvh))
    assert np.array_equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-13)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-11)
    for k in (2, 3, 4):                          # the storage-order kernel's forms among themselves: every bit
        for a_, b_ in zip(res[0], res[k]):
            assert np.array_equal(a_, b_), k
    assert np.abs(res[0][2] - rv).max() > 0


@pytest.mark.parametrize("budget", [-1, 6, 2])
def test_folded_condensation_kernel_is_bit_identical_to_the_plain_one(monkeypatch, budget):
    """k_cond_fast_fold (the workgroup's droplets that enter the root finder's loop handed through LDS to its lowest lanes; the
    production first pass) against k_cond_fast (opts_init.dbg_flags & COND_NO_FOLD): the same arithmetic per droplet on another lane"""
    oi = h.box_opts(16, 8, 8, 64, sstp_cond=2, strict_fp=False)
    fields = h.box_fields(oi)
    oi.dbg_cond_budget = budget
    oi.cond_solver = 1
    res = []
    for plain in (False, True):
        oi.dbg_flags = int(lgrngn.dbg.COND_TOMS_TWO_PASS | (lgrngn.dbg.COND_NO_FOLD if plain else 0))
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = opts.adve = opts.sedi = False
        thh, rvh = th.copy(), rv.copy()
        for _ in range(2):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        res.append((hip.get_attr("rw2"), thh, rvh))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("sd_conc,dims", [(1, (20, 18, 22)), (3, (9, 7, 11)), (64, (6, 5, 7)), (100, (6, 5, 7)), (127, (4, 4, 4)), (128, (4, 4, 4)),
                                          (150, (5, 4, 6)), (230, (4, 4, 5)), (64, (40, 0, 30))])
@pytest.mark.parametrize("flags", [0, "NO_RANK_OVERLAP", "NO_DEFERRED_SORT"])
def test_in_cell_order_by_buckets_is_the_order_by_counting(sd_conc, dims, flags):
    """k_cellrank_bkt (round 4: a key's bucket is its expected rank in its cell, one prefix sum over the workgroup's staged range, compares
    within the bucket only) against k_cellrank<uint32_t, true> (dbg_flags & RANK_BY_COUNTING: every key compared with every key of its cell):
    the keys are unique, so both give THE order -- sorted_id after every step's coalescence and everything that follows from it, bit for bit.
    Few droplets per cell (hundreds of cells per workgroup), the production 64, cells that straddle workgroups with more than the
    speculative window covers (150, 230: the counting fallback inside the kernel), a 2-D box; with the ranking on its side stream (the
    default), on the object's one stream, and without the deferred re-sort."""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, sd_conc, strict_fp=False)
    fields = h.box_fields(oi)
    res = []
    for counting in (False, True):
        oi.dbg_flags = int((lgrngn.dbg[flags] if flags else 0) | (lgrngn.dbg.RANK_BY_COUNTING if counting else 0))
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        out = []
        for _ in range(4):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            out.append(hip.state_u64("raw_sorted_id"))          # (the order the step's coalescence will pair up)
            hip.step_async(opts)
        res.append(out + [hip.get_attr("rw2"), thh, rvh, hip.state_u64("n"), hip.get_attr("x")])
    assert not np.array_equal(res[0][1], res[0][2])            # (a fresh shuffle every step)
    for a_, b_ in zip(res[0], res[1]):
        assert np.array_equal(a_, b_)


def test_in_cell_order_by_buckets_with_a_few_crowded_cells():
    """The bucket ranking in a box whose MEAN occupancy keeps it on (64 per cell) while single cells are crowded: 184 droplets in one cell
    (more than the speculative window of a neighbouring workgroup covers: that workgroup ranks by counting inside the same kernel) and
    394 in another (above k_cellrank's limit: listed, sorted by one wave) -- the same order as ranking everything by counting."""
    oi = h.box_opts(6, 5, 7, 64, strict_fp=False)
    fields = h.box_fields(oi)
    res = []
    for counting in (False, True):
        oi.dbg_flags = int(lgrngn.dbg.RANK_BY_COUNTING) if counting else 0
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        x, y, z = hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z")
        for sl, cell in ((slice(1000, 1120), (1, 1, 1)), (slice(5000, 5330), (3, 2, 4))):
            x[sl], y[sl], z[sl] = (cell[0] + .5) * oi.dx, (cell[1] + .5) * oi.dy, (cell[2] + .5) * oi.dz
        hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), hip.get_attr("rw2"), hip.get_attr("kappa"), hip.state_real("vt"), x, y, z)
        opts = lgrngn.opts_t()
        opts.adve = opts.sedi = False                      # (the crowd stays together)
        thh, rvh = th.copy(), rv.copy()
        out = []
        for _ in range(3):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            out.append(hip.state_u64("raw_sorted_id"))
            hip.step_async(opts)
        cnt = np.diff(hip.state_u64("cell_start").astype(np.int64))
        assert cnt.max() > 300 and ((cnt > 150) & (cnt <= 256)).any()
        res.append(out + [hip.get_attr("rw2"), hip.state_u64("n")])
    for a_, b_ in zip(res[0], res[1]):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize("sd_conc,steps", [(64, 6), (100, 4), (400, 3)])
def test_per_cell_finish_without_its_lds_stage_gives_the_same_bits(sd_conc, steps):
    """k_cond_cellfinish_direct (round 4: the changes lie in the sorted order, eight lanes per cell read them straight from memory, the
    first 64 of a cell stay in registers between the two passes) against the staged k_cond_cellfinish (dbg_flags & FINISH_STAGED): the
    per-cell sums are integer sums, so th and rv -- and with them everything that follows -- are the same bits.  Full steps, ordinary
    cells, cells just above the registers' 64 and crowded ones (the loop that re-reads what lies beyond)"""
    oi = h.box_opts(6, 5, 7, sd_conc, sstp_cond=2, strict_fp=False)
    fields = h.box_fields(oi)
    res = []
    for staged in (False, True):
        # (sstp_cond = 2: the per-substep launches in both, round 6's k_cond_substeps has a finish of its own)
        oi.dbg_flags = int(lgrngn.dbg.FINISH_STAGED if staged else lgrngn.dbg.COND_NO_FUSED_SUBSTEPS)
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        for _ in range(steps):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        h.assert_mode(hip, False, None, ("lean", "lean_toms748"))
        res.append((hip.get_attr("rw2"), thh, rvh, hip.state_u64("n")))
    for a_, b_ in zip(res[0], res[1]):
        assert np.array_equal(a_, b_)
    assert np.abs(res[0][2] - rv).max() > 0


@pytest.mark.parametrize("sd_conc,steps", [(64, 6), (400, 3)])
def test_storage_order_condensation_is_bit_identical_to_the_positional_one(monkeypatch, sd_conc, steps):
    """k_cond_lean takes the droplets in STORAGE order (coalesced attribute reads and writes; the per-cell finish gathers the droplets'
    changes through sorted_id) -- opts_init.dbg_flags & COND_SORTED_ORDER selects the positional form it replaced (every attribute gathered through
    sorted_id).  A droplet's answer does not depend on the lane that computes it and a cell's sum keeps its order: the same bits, in a
    full step with coalescence, advection and sedimentation (dead slots in the storage, the shuffled order of the next coalescence in
    place when condensation runs), for ordinary cells and for crowded ones (400 per cell: the wave-per-cell finish)"""
    oi = h.box_opts(6, 5, 7, sd_conc, sstp_cond=2, strict_fp=False)
    fields = h.box_fields(oi)
    res = []
    for positional in (False, True):
        # (sstp_cond = 2: the per-substep storage-order kernel against the positional one; round 6's k_cond_substeps has its own test)
        oi.dbg_flags = int(lgrngn.dbg.COND_SORTED_ORDER if positional else lgrngn.dbg.COND_NO_FUSED_SUBSTEPS)
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        rw2 = hip.get_attr("rw2")
        rw2[::7] = (60e-6) ** 2                                  # some drizzle: collisions use super-droplets up, a few fall out
        hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), rw2, hip.get_attr("kappa"), np.full(rw2.size, -1.),
                          hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        for _ in range(steps):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        h.assert_mode(hip, False, None, ("lean_sorted", "lean_toms748_sorted") if positional else ("lean", "lean_toms748"))
        res.append((hip.get_attr("rw2"), thh, rvh, hip.state_u64("n"), hip.n_part))
    assert res[0][4] == res[1][4] and res[0][4] < oi.nx * oi.ny * oi.nz * sd_conc      # (super-droplets were used up)
    for a_, b_ in zip(res[0][:4], res[1][:4]):
        assert np.array_equal(a_, b_)
    assert np.abs(res[0][2] - rv).max() > 0


@pytest.mark.parametrize("sd_conc,steps,reorder_every", [(64, 8, 3), (400, 3, 0)])
def test_lean_kernel_round4_and_round3_solver_forms_give_the_same_bits(sd_conc, steps, reorder_every):
    """Round 4's k_cond_lean runs the solver with its bookkeeping pared down (straight-line loop body, helper functions without the
    instructions that are identities for a squared radius) and takes the run's single hygroscopicity as a scalar instead of reading
    8 bytes per droplet.  A droplet's answer depends on neither: the same rw2, th, rv and multiplicities bit for bit with the array
    read (opts_init.dbg_flags & KPA_ARRAY) and with round 3's form of the solver (COND_LEAN_R3), over full steps with coalescence, dead
    slots and storage re-orderings"""
    oi = h.box_opts(12, 10, 14, sd_conc, sstp_cond=2, strict_fp=False)
    oi.reorder_every = reorder_every
    fields = h.box_fields(oi)
    res = []
    # (round 5: COND_FOLD is the kernel FOLDED behind the solver's first loop trip -- k_cond_lean_fold, the unconverged droplets of a
    # workgroup handed to its lowest lanes through LDS -- and a stage of 8 slots (dbg_cond_budget) leaves most of the unconverged
    # droplets in their own lanes: every droplet's numbers see the same operations in all of them)
    # (late round 5: the production kernel hands the droplets whose bracket may hold several roots to TOMS748, cond_list; the variants
    # compared HERE are kernels around the lean solver alone -- COND_NO_LIST in all of them; the list has its own test below)
    # (round 6: COND_WQ is k_cond_lean_wq -- a wave walks several batches of 64 storage slots and keeps the droplets whose first loop
    # trip has not converged on a queue of its own in LDS, taking them up again 64 at a time; dbg_cond_budget = the batches per wave (a box
    # this small gets one by default: the queue then only empties at the wave's end), COND_WQ_CAP128 the longer queue, COND_WQ_PF / _PF2
    # the next batch's loads issued ahead.  Measured, not adopted.  Every droplet's numbers see the same operations in all of them)
    FO, NL = int(lgrngn.dbg.COND_FOLD), int(lgrngn.dbg.COND_NO_LIST)
    WQ, W128, KA = int(lgrngn.dbg.COND_WQ), int(lgrngn.dbg.COND_WQ_CAP128), int(lgrngn.dbg.KPA_ARRAY)
    PF, PF2 = int(lgrngn.dbg.COND_WQ_PF), int(lgrngn.dbg.COND_WQ_PF2)
    # (sstp_cond = 2: without a switch that names a per-substep kernel both substeps are k_cond_substeps' -- the first two variants; NF is
    # the production kernel of sstp_cond = 1 launched per substep.  Every variant says which kernel it ran.)
    NF, R3 = int(lgrngn.dbg.COND_NO_FUSED_SUBSTEPS), int(lgrngn.dbg.COND_LEAN_R3)
    ran = {0: "substeps", KA: "substeps", NF: "lean", NF | KA: "lean", R3: "lean_r3"}
    for flags, budget in ((NF, 0), (NF | KA, 0), (0, 0), (KA, 0), (R3, 0), (FO, 0), (FO | KA, 0), (FO, 8),
                          (WQ, 0), (WQ, 3), (WQ, 16), (WQ | KA, 5), (WQ | W128, 0), (WQ | W128, 7), (WQ | W128 | KA, 2),
                          (WQ | PF, 4), (WQ | W128 | PF | KA, 3), (WQ | PF2, 5), (WQ | PF2 | KA, 2)):
        oi.dbg_flags = flags | NL
        oi.dbg_cond_budget = budget
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)            # (no set_particles: the run keeps its single hygroscopicity, the scalar form is the default)
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        for _ in range(steps):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        h.assert_mode(hip, False, 0, ran.get(flags, "fold_lean" if flags & FO else "lean_wq"))
        res.append((hip.get_attr("rw2"), thh, rvh, hip.state_u64("n"), hip.n_part))
    assert len(set(r_[4] for r_ in res)) == 1
    for k in range(1, len(res)):
        for a_, b_ in zip(res[0][:4], res[k][:4]):
            assert np.array_equal(a_, b_), k
    assert np.abs(res[0][2] - rv).max() > 0


@pytest.mark.parametrize("sd_conc,steps,reorder_every", [(64, 6, 3), (400, 3, 0)])
def test_the_first_pass_budget_is_unobservable(sd_conc, steps, reorder_every):
    """Round 6, measured and not adopted (dbg COND_BUDGET): k_cond_lean gives every droplet's loop a BUDGET of two trips; a droplet that has
    not converged by then leaves the loop's state where it stands in a record (cond_list: 64 parts with a counter each, like the list of
    droplets for the reference's iterates now) and k_cond_lean_resume -- a dense walk of those parts -- goes on with it: a wave of the first
    pass does not wait for its slowest droplet.  A droplet's answer depends on neither the budget nor on who computes it: the same rw2,
    th, rv and multiplicities bit for bit without a budget (the production kernel), with budgets of two (straight-line trips), one and
    three trips (the run-time form of the loop), with room for only 4 records per part (dbg_cond_budget >> 8: the droplets that find
    their part full carry on in their own lanes), over the sorted order, and through k_cond_lean_wq; full steps with coalescence, dead
    slots and storage re-orderings."""
    oi = h.box_opts(12, 10, 14, sd_conc, sstp_cond=2, strict_fp=False)
    oi.reorder_every = reorder_every
    fields = h.box_fields(oi)
    res = []
    BU, SO, WQ, KA, NF = (int(lgrngn.dbg[k]) for k in ("COND_BUDGET", "COND_SORTED_ORDER", "COND_WQ", "KPA_ARRAY", "COND_NO_FUSED_SUBSTEPS"))
    # (sstp_cond = 2: NF keeps the first variant on the per-substep production kernel -- without it both substeps are k_cond_substeps')
    for flags, budget in ((NF, 0), (BU, 0), (BU, 1), (BU, 3), (BU, 2 | 4 << 8), (BU | KA, 0), (SO, 0), (SO | BU, 0), (WQ, 3), (0, 0)):
        oi.dbg_flags = flags
        oi.dbg_cond_budget = budget
        hip = h.hip_particles(oi)
        assert not hip.opts_init.strict_fp and hip.opts_init.cond_solver == 0
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        resumed = []
        for _ in range(steps):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            resumed.append(int(hip.state_u64("raw_cond_resumed")[0]))
            hip.step_async(opts)
        h.assert_mode(hip, False, 0, "substeps" if not flags else "lean_wq" if flags & WQ else "lean_sorted" if flags & SO else "lean")
        res.append((hip.get_attr("rw2"), thh, rvh, hip.state_u64("n"), hip.n_part, resumed))
    assert len(set(r_[4] for r_ in res)) == 1
    for k in range(1, len(res)):
        for a_, b_ in zip(res[0][:4], res[k][:4]):
            assert np.array_equal(a_, b_), k
    # (no budget, no records; a budget of one trip leaves more droplets behind than one of two, and that more than one of three; the
    # counter also counts the droplets that found their part full)
    assert sum(res[0][5]) == 0 and sum(res[2][5]) > sum(res[1][5]) > sum(res[3][5]) > 0, [sum(r_[5]) for r_ in res]
    assert res[4][5] == res[1][5] and max(res[4][5]) > 64 * 4
    print("records per substep (the step's last): two trips %s, one %s, three %s" % tuple(res[k][5][:3] for k in (1, 2, 3)))


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("dims,sd_conc,sstp,real_t", [((12, 10, 14), 64, 4, np.float64), ((9, 0, 11), 40, 10, np.float32), ((5, 4, 6), 300, 3, np.float64),
                                                       ((0, 0, 0), 500, 5, np.float64)])
def test_all_substeps_in_one_launch_give_the_same_bits(dims, sd_conc, sstp, real_t, solver):
    """Round 6.  With sstp_cond > 1 the fast arithmetic makes every condensation substep of a step in ONE launch (k_cond_substeps: a
    workgroup owns a run of cells and their droplets; per substep the cell pass, the droplets, the cells' fixed-point sums and the
    feedback on th and rv, with workgroup barriers in between) instead of a cell pass, a condensation kernel and a per-cell finish per
    substep (dbg COND_NO_FUSED_SUBSTEPS).  The same operations on the same numbers: rw2, th, rv, multiplicities and the sorted order bit
    for bit after full steps with coalescence and advection -- both solvers, 3-D / 2-D / 0-D, float as C2 runs it, cells of 300 droplets
    (more than a workgroup's lanes per cell)."""
    nx, ny, nz = dims
    kw = dict(sstp_cond=sstp, strict_fp=False)
    if not nz:
        kw["sedi_switch"] = False
    oi = h.box_opts(nx, ny, nz, sd_conc, **kw)
    oi.cond_solver = solver
    fields = h.box_fields(oi)
    res = []
    for flags in (int(lgrngn.dbg.COND_NO_FUSED_SUBSTEPS), 0):
        oi.dbg_flags = flags
        hip = h.hip_particles(oi, real_t)
        th, rv, rhod, C = [f.astype(real_t) if isinstance(f, np.ndarray) else {k: v.astype(real_t) for k, v in f.items()} for f in fields]
        hip.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        if not nz:
            opts.sedi = opts.adve = False
        thh, rvh = th.copy(), rv.copy()
        for _ in range(4):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
        h.assert_mode(hip, False, solver, "substeps" if not flags else ("lean", "lean_sorted", "fold_toms748", "lean_toms748", "lean_toms748_sorted"))
        res.append((hip.get_attr("rw2"), thh, rvh, hip.state_u64("n"), hip.state_u64("sorted_id"), hip.n_part))
    assert res[0][5] == res[1][5]
    if real_t is np.float32 and solver == 0:
        # (float's growth rate is left to the compiler's contraction, `#pragma clang fp contract(fast)`, which may fuse differently in two
        # kernels; TOMS748's iterates came out the same here, the lean solver's stopping decisions at float's tolerance 2^-7 flip on an ulp
        # for a tenth of the droplets: the same answers to that tolerance, not the same bits)
        assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=2. ** -7)
        np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-6)
        np.testing.assert_allclose(res[0][2], res[1][2], rtol=3e-6)
    else:
        for a_, b_ in zip(res[0][:5], res[1][:5]):
            assert np.array_equal(a_, b_)
    assert np.abs(res[0][2] - fields[1].astype(real_t)).max() > 0


@pytest.mark.parametrize("dims,sd_conc,sstp,real_t,vt,kernel", [
    ((9, 0, 11), 64, 10, np.float32, "khvorostyanov_spherical", "geometric"),       # C2's shape
    ((7, 6, 8), 64, 4, np.float64, "beard77fast", "hall_davis_no_waals"),
    ((6, 5, 4), 40, 3, np.float64, "beard77", "geometric"),
    ((5, 4, 6), 120, 2, np.float64, "beard76", "Long"),
    ((4, 3, 4), 300, 3, np.float64, "beard77fast", "geometric"),                    # crowded cells: another ranking kernel, the pass of its own
])
@pytest.mark.parametrize("strict", [False, True])
def test_invalid_velocities_refreshed_by_the_next_substeps_ranking(dims, sd_conc, sstp, real_t, vt, kernel, strict):
    """Round 6.  Between two coalescence substeps the reference refreshes the terminal velocities that the collisions have invalidated
    (hskpng_vterm_invalid, particles_step.ipp:389-391).  Here the in-cell ranking of the NEXT substep does it on its way -- every droplet
    of the order passes that kernel, and the coalescence kernel behind it is the first reader (k_cellrank_bkt's EXTRA; dbg
    VTERM_INVALID_OWN_PASS keeps the launch of its own): one launch less per substep, the same multiplicities, radii, velocities and
    order bit for bit after full steps on drizzle that collides -- both arithmetics, every velocity formula's path, float as C2 runs it;
    where another ranking kernel runs (crowded cells) the pass of its own stays."""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, sd_conc, sstp_coal=sstp, sstp_cond=2, strict_fp=strict)
    oi.terminal_velocity = getattr(lgrngn.vt_t, vt)
    oi.kernel = getattr(lgrngn.kernel_t, kernel)
    fields = h.box_fields(oi)
    res = []
    for flags in (int(lgrngn.dbg.VTERM_INVALID_OWN_PASS), 0):
        oi.dbg_flags = flags
        hip = h.hip_particles(oi, real_t)
        th, rv, rhod, C = [f.astype(real_t) if isinstance(f, np.ndarray) else {k: v.astype(real_t) for k, v in f.items()} for f in fields]
        hip.init(th, rv, rhod, **C)
        rw2 = hip.get_attr("rw2")
        rw2[::5] = (70e-6) ** 2                                  # drizzle among the aerosol: pairs collide in every substep
        rw2[1::11] = (25e-6) ** 2
        hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), rw2, hip.get_attr("kappa"), np.full(rw2.size, -1.),
                          hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
        opts = lgrngn.opts_t()
        thh, rvh = th.copy(), rv.copy()
        n0 = hip.state_u64("n").sum()
        launches = []
        for _ in range(3):
            hip.step_sync(opts, thh, rvh, rhod, **C)
            l0 = int(hip.state_u64("raw_launches")[0])
            hip.step_async(opts)
            launches.append(int(hip.state_u64("raw_launches")[0]) - l0)
        res.append((hip.state_u64("n"), hip.get_attr("rw2"), hip.get_attr("rd3"), hip.state_real("vt"), hip.state_u64("sorted_id"), thh, rvh, hip.n_part, n0,
                    launches))
    assert res[0][7] == res[1][7]
    assert res[0][0].sum() < res[0][8]                           # (multiplicity was used up: collisions happened)
    for a_, b_ in zip(res[0][:7], res[1][:7]):
        assert np.array_equal(a_, b_)
    # (a launch less per substep boundary where the bucket ranking runs; the same launches where it does not)
    saved = [a_ - b_ for a_, b_ in zip(res[0][9], res[1][9])]
    assert saved == [sstp - 1 if sd_conc <= 128 else 0] * 3, (res[0][9], res[1][9])


def test_brackets_that_may_hold_several_roots_take_the_references_iterates():
    """Late round 5.  A droplet that can evaporate down to its dry core within the step (a 0.8 um droplet on a 5 nm core in subsaturated
    air: a root where it has shrunk to half its radius, and roots next to the core where the Kelvin term takes over), or whose bracket
    spans more than a factor of four in radius in supersaturated air, has SEVERAL roots of the step's equation inside the reference's
    bracket; which one TOMS748 returns is a property of its iterates (at 2^24 droplets the lean solver differed for up to 65 per step,
    tests/test_hip_reverse_replay.py).  k_cond_lean lists such droplets (lcx_math.hpp lean2_head, `suspicious`) and
    k_cond_lean_listed takes them through TOMS748 on the same growth-rate arithmetic.  One condensation step of a box that holds such
    droplets, three ways: the production kernel, the same without the list (dbg COND_NO_LIST: the lean solver for everybody), and
    cond_solver = 1 (TOMS748 for everybody) -- a droplet of the production run carries EITHER the lean solver's bits or TOMS748's, both
    kinds occur, and every droplet whose lean answer is far from TOMS748's is among the listed."""
    oi = h.box_opts(8, 8, 8, 64, sstp_cond=1, strict_fp=False)
    th, rv, rhod, C = h.box_fields(oi)
    res = {}
    for name, flags, solver in (("prod", 0, 0), ("lean", int(lgrngn.dbg.COND_NO_LIST), 0), ("toms", 0, 1)):
        oi.dbg_flags = flags
        oi.cond_solver = solver
        hip = h.hip_particles(oi)
        hip.init(th, rv, rhod, **C)
        g = hip.state_real
        rw2, rd3 = g("rw2"), g("rd3")
        rng = np.random.default_rng(9)
        rd = np.cbrt(rd3)
        big = rng.random(rw2.shape) < .3
        rw2[big] = np.maximum(rd[big] * 1.05, rng.uniform(.2e-6, 1.5e-6, size=int(big.sum()))) ** 2      # droplets of 0.2 ... 1.5 um on whatever core they have
        hip.set_particles(hip.state_u64("n"), rd3, rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
        opts = lgrngn.opts_t()
        opts.coal = opts.sedi = opts.adve = False
        thh, rvh = th.copy(), rv.copy()
        hip.step_sync(opts, thh, rvh, rhod, **C)
        hip.step_async(opts)
        res[name] = hip.get_attr("rw2")
    prod, lean, toms = res["prod"], res["lean"], res["toms"]
    from_lean, from_toms = prod == lean, prod == toms
    assert (from_lean | from_toms).all(), int((~(from_lean | from_toms)).sum())
    # (this box is made of such droplets -- fresh aerosol activating at RH 1.01, droplets of a micrometre on nanometre cores: most of it
    # is listed; bench.py's settled boxes list about 0.1 %)
    assert (~from_lean).sum() > 100 and (~from_toms).sum() > 100, ((~from_lean).sum(), (~from_toms).sum())
    far = np.abs(lean / toms - 1.) > 1e-4
    assert from_toms[far].all(), int((~from_toms[far]).sum())
    print("TOMS748's bits %d, the lean solver's %d (both: %d) of %d; lean solver far from TOMS748's root: %d, all of them listed" % (
        from_toms.sum(), from_lean.sum(), (from_lean & from_toms).sum(), prod.size, far.sum()))


@pytest.mark.parametrize("sd_conc,reorder_every,cond_every", [(64, 0, 1), (64, 3, 1), (300, 4, 1), (48, 5, 2)])
def test_deferred_sort_is_bit_identical_to_the_immediate_one(monkeypatch, sd_conc, reorder_every, cond_every):
    """The end-of-step re-sort of a single device leaves its scatter and in-cell ranking to the next step: the storage-order condensation
    kernel carries the scatter (its memory pipes idle while its vector ALU is the bottleneck), the ranking follows it, and whoever
    else needs the sorted order first (a diagnostic, a step without condensation, coalescence) finishes the sort where it stands.
    opts_init.dbg_flags & NO_DEFERRED_SORT sorts at once.  The random keys are drawn at the same place of the generator's sequence either way: the
    same bits after full steps with coalescence -- across storage re-orderings (no deferral in those steps), with crowded cells (300 per
    cell: the listed-cell sorts), with condensation switched off every other step, and with a diagnostic between the steps"""
    oi = h.box_opts(6, 5, 7, sd_conc, strict_fp=False)
    oi.reorder_every = reorder_every
    fields = h.box_fields(oi)
    res = []
    for immediate in (False, True):
        oi.dbg_flags = int(lgrngn.dbg.NO_DEFERRED_SORT) if immediate else 0
        hip = h.hip_particles(oi)
        th, rv, rhod, C = fields
        hip.init(th, rv, rhod, **C)
        rw2 = hip.get_attr("rw2")
        rw2[::7] = (60e-6) ** 2
        hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), rw2, hip.get_attr("kappa"), np.full(rw2.size, -1.),
                          hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
        thh, rvh = th.copy(), rv.copy()
        conc = []
        for it in range(9):
            opts = lgrngn.opts_t()
            opts.cond = it % cond_every == 0
            hip.step_sync(opts, thh, rvh, rhod, **C)
            hip.step_async(opts)
            if it in (2, 6):                                      # a reader of the sorted order between two steps
                hip.diag_all(); hip.diag_sd_conc(); conc.append(hip.outbuf_array())
            if it == 4:                                           # the caller replaces the droplets while a sort is left undone: that sort is void
                r2 = hip.get_attr("rw2")
                r2[::11] *= 1.5
                hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), r2, hip.get_attr("kappa"), np.full(r2.size, -1.),
                                  hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
        res.append((hip.get_attr("rw2"), hip.get_attr("x"), thh, rvh, hip.state_u64("n"), np.stack(conc), hip.n_part))
    assert res[0][6] == res[1][6] and res[0][6] < oi.nx * oi.ny * oi.nz * sd_conc
    for a_, b_ in zip(res[0][:6], res[1][:6]):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize("dims", [(0, 0, 0), (4, 3, 5)])
def test_rcyc_matches_oracle(dims):
    """opts.rcyc: the SDs freed by coalescence / precipitation are re-used as halves of the SDs with the highest
    multiplicities (rcyc.ipp:44-140) -- same receivers, same donors (largest n first, higher id first among equals), same
    split as the oracle's full stable sort of the multiplicities"""
    nx, ny, nz = dims
    oi = h.box_opts(nx, ny, nz, 256 if nx == 0 else 40, sedi_switch=nz > 0)
    if nx == 0:
        oi.dx = oi.dy = oi.dz = oi.x1 = oi.y1 = oi.z1 = 1.
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    # rain-sized droplets: plenty of collisions and fall-out, so that several SDs per step reach n == 0
    rw2 = orc.get_attr("rw2")
    rw2[::2] = np.linspace(40e-6, 1.2e-3, len(rw2[::2])) ** 2
    nn = orc.state_u64("n")
    nn[1::3] = 1 + nn[1::3] % 3                                  # low multiplicities: collisions exhaust them
    g = lambda nm: orc.state_real(nm)
    args = (nn, g("rd3"), rw2, g("kappa"), g("vt"), g("x") if nx else None, g("y") if ny else None, g("z") if nz else None)
    orc.set_particles(*args)
    hip.set_particles(*args)
    opts = lgrngn.opts_t()
    opts.cond = False
    opts.rcyc = True
    if nz == 0:
        opts.sedi = opts.adve = False
    recycled = 0
    for it in range(5):
        n_before = orc.n_part
        step_pair(orc, hip, opts, fields)
        assert hip.n_part == orc.n_part
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-13)
        np.testing.assert_allclose(hip.get_attr("rd3"), orc.get_attr("rd3"), rtol=1e-13)
        recycled += int(orc.n_part == n_before)
        h.copy_state(orc, hip)
    assert recycled > 0, "test needs steps in which every freed SD was recycled"


def test_storage_reorder_is_a_permutation_of_the_same_state():
    """opts_init.reorder_every (extension): physically re-ordering the storage into the cell order renumbers the ids but
    must not change any super-droplet: with coalescence off (no id-keyed random numbers) a run that re-orders after every
    step carries the same set of droplets as one that never does (-1: stable compaction only, the reference's storage order);
    th / rv agree to summation-order rounding"""
    runs = []
    for every in (-1, 1):
        oi = h.box_opts(5, 4, 6, 40, reorder_every=every)
        fields = h.box_fields(oi)
        th, rv, rhod, C = fields
        pr = h.hip_particles(oi)
        pr.init(th.copy(), rv.copy(), rhod.copy(), **C)
        opts = lgrngn.opts_t()
        opts.coal = False
        tht, rvt = th.copy(), rv.copy()
        for _ in range(6):
            pr.step_sync(opts, tht, rvt, rhod, **C)
            pr.step_async(opts)
        key = np.argsort(pr.get_attr("rd3"), kind="stable")
        runs.append((tht, rvt, pr.n_part, {a: pr.get_attr(a)[key] for a in ("rd3", "rw2", "x", "y", "z", "kappa")},
                     pr.state_u64("n")[key], pr.state_u64("sorted_id"), pr.state_u64("sorted_ijk"), pr.state_u64("ijk")))
    a, b = runs
    assert a[2] == b[2]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-9)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-8)
    exact(b[4], a[4], "multiplicities")
    exact(b[3]["rd3"], a[3]["rd3"], "rd3")
    for k in ("x", "y", "z"):
        np.testing.assert_allclose(b[3][k], a[3][k], rtol=1e-12, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(b[3]["rw2"], a[3]["rw2"], rtol=2e-4)       # the root finder's tolerance (rounding of th / rv decides its last iteration)
    assert np.median(np.abs(b[3]["rw2"] / a[3]["rw2"] - 1)) < 1e-9
    # the re-ordered run's storage IS the cell order
    sid, sijk, ijk = b[5], b[6], b[7]
    exact(sid, np.arange(len(sid), dtype=sid.dtype), "sorted_id is the identity right after a re-order")
    exact(ijk, sijk, "ijk == sorted_ijk")


def test_compaction_gathers_in_cell_order_by_default():
    """opts_init.reorder_every == 0 (default): when enough super-droplets have died, the pass that drops them gathers the
    survivors in the cell-sorted order; -1 keeps the reference's stable compaction.  Same droplets either way."""
    runs = []
    for every in (-1, 0):
        oi = h.box_opts(5, 4, 6, 48, dx=30., coal_switch=False, reorder_every=every)
        th, rv, rhod, C = h.box_fields(oi)
        pr = h.hip_particles(oi)
        pr.init(th.copy(), rv.copy(), rhod.copy(), **C)
        n0 = pr.n_part
        rw2 = pr.get_attr("rw2")
        rw2[::5] = (1.2e-3) ** 2                      # mm-sized drops: a fifth of the SDs rains out within a few steps
        pr.set_particles(pr.state_u64("n"), pr.state_real("rd3"), rw2, pr.state_real("kappa"), pr.state_real("vt"),
                         pr.state_real("x"), pr.state_real("y"), pr.state_real("z"))
        opts = lgrngn.opts_t()
        opts.coal = opts.cond = False
        for _ in range(8):
            pr.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
            pr.step_async(opts)
        assert pr.n_part < 0.96 * n0                  # more than the 1/32 that triggers a compaction
        key = np.argsort(pr.get_attr("rd3"), kind="stable")
        runs.append((pr.n_part, {a: pr.get_attr(a)[key] for a in ("rd3", "rw2", "x", "y", "z")}, pr.state_u64("n")[key],
                     pr.state_u64("ijk"), pr.state_u64("sorted_ijk"), pr.state_u64("sorted_id")))
    a, b = runs
    assert a[0] == b[0]
    exact(b[2], a[2], "multiplicities")
    for k in a[1]:
        exact(b[1][k], a[1][k], k)
    assert not np.array_equal(a[3], a[4])                    # stable compaction: storage order is creation order
    ijk, sijk, sid = b[3], b[4], b[5]
    nondecreasing_from = int(np.argmax(np.diff(ijk.astype(np.int64)) < 0)) if np.any(np.diff(ijk.astype(np.int64)) < 0) else len(ijk)
    assert nondecreasing_from > 0.5 * len(ijk)               # the last compaction left the storage cell-sorted (later steps moved a few)


# ------------------------------------------------------------------ SGS turbulence
@pytest.mark.parametrize("mode,dims", [("adve", (6, 0, 5)), ("adve", (4, 3, 5)), ("cond", (5, 0, 6)), ("both", (4, 3, 4)),
                                       ("cond_pp_mix", (5, 0, 6)), ("cond_pp_nomix", (5, 0, 6)), ("cond_pp_adaptive", (4, 3, 4))])
def test_sgs_turbulence_matches_oracle(mode, dims):
    """turb_adve (Ornstein-Uhlenbeck velocity perturbations added to the advection) and turb_cond (SGS supersaturation
    perturbation in the condensation) against the oracle, its normal deviates (std::normal_distribution over mt19937)
    replayed: hskpng_tke / _turb_vel / _turb_dot_ss, turb_adve, apply_perparticle_sgs_supersat, RH_sgs"""
    nx, ny, nz = dims
    pp = mode.startswith("cond_pp")            # turb_cond with per-particle substepping (unit/sstp_cond.py runs this combination)
    kw = dict(coal_switch=False, turb_adve_switch=mode in ("adve", "both"), turb_cond_switch=mode in ("cond", "both") or pp,
              SGS_mix_len=np.linspace(20., 40., nz), sstp_cond=2 if mode == "cond" else 4 if pp else 1)
    if pp:
        kw.update(exact_sstp_cond=True, sstp_cond_mix=mode == "cond_pp_mix", adaptive_sstp_cond=mode == "cond_pp_adaptive")
    oi = h.box_opts(nx, ny, nz, 32, dx=30., **kw)
    th, rv, rhod, C = h.box_fields(oi)
    diss = 1e-3 * (1 + np.random.default_rng(3).random(th.shape))
    orc, hip = h.make_pair(oi, (th, rv, rhod, C))
    opts = lgrngn.opts_t()
    opts.coal = False
    opts.turb_adve = mode in ("adve", "both")
    opts.turb_cond = mode in ("cond", "both") or pp
    ndim = sum(1 for n_ in dims if n_ > 0)
    for it in range(3):
        tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
        orc.step_sync(opts, tho, rvo, rhod, diss_rate=diss, **C)
        hip.step_sync(opts, thh, rvh, rhod, diss_rate=diss, **C)
        n_calls = ndim if opts.turb_adve else 1
        for arr in h.oracle_rng_preview(orc, [(2, orc.n_part)] * n_calls):
            hip.rng_replay_push(2, arr)
        orc.step_async(opts)
        hip.step_async(opts)
        assert hip.n_part == orc.n_part
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        for nm in (("up", "wp") + (("vp",) if ny else ())) if opts.turb_adve else ("wp",):
            np.testing.assert_allclose(hip.state_real(nm), orc.state_real(nm), rtol=1e-10, atol=1e-14, err_msg=nm)   # vel e + sqrt(..) r cancels
        if opts.turb_cond:
            # tau_relax comes from the first wet moment, i.e. from radii that carry the root finder's tolerance
            np.testing.assert_allclose(hip.state_real("ssp"), orc.state_real("ssp"), rtol=1e-5, atol=1e-11)
            np.testing.assert_allclose(hip.state_real("dot_ssp"), orc.state_real("dot_ssp"), rtol=1e-5, atol=1e-11)
            np.testing.assert_allclose(thh, tho, rtol=1e-7)
            np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=2e-4)
        for a in ("x", "y", "z"):
            if getattr(oi, "n" + a):
                # with condensation on, z carries dt * vt(rw2) and rw2 the root finder's 2e-4 tolerance
                np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-13, atol=1e-3 if opts.turb_cond and a == "z" else 1e-9, err_msg=a)
    with pytest.raises(RuntimeError):                       # diss_rate is mandatory once a turbulence switch is on
        hip.step_sync(opts, th.copy(), rv.copy(), rhod, **C)
