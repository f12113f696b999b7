"""BASELINE.json's configurations as parity cases (the bench line is configs[2]; tests/test_hip_fullsize.py holds its
full-size properties):
  C1  0-D adiabatic parcel, 1e4 super-droplets, geometric kernel, beard76                      -> vs oracle, replayed stream
  C2  2-D kinematic icicle set-up (GMD 2015): 76 x 76 cells x 64 SDs, sstp 10/10, implicit advection,
      khvorostyanov_spherical, geometric x 0.5 (models/kinematic_2D/src/opts_lgrngn.hpp:262,340-343,
      kin_cloud_2d_lgrngn.hpp:167-196) -- reduced to 24 x 24 cells for the oracle, full size in float as a property run
  C4  1-D decomposition over 8 slabs (256 x 256 x 128 in BASELINE; 16 x 4 x 6 here) as a ring on one GPU -> vs oracle ring
  C5  512 super-droplets per cell (per-cell segments larger than a wave / the ranking crossover)  -> vs oracle
"""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn
from test_hip_parity import step_pair, exact

pytestmark = pytest.mark.gpu


def test_c1_parcel_0d_1e4():
    oi = h.box_opts(0, 0, 0, 10000, sedi_switch=False, kernel=lgrngn.kernel_t.geometric, terminal_velocity=lgrngn.vt_t.beard76)
    oi.dx = oi.dy = oi.dz = 1.
    oi.x1 = oi.y1 = oi.z1 = 1.
    fields = (np.array([300.]), np.array([0.02]), np.array([1.1]), {})          # supersaturated variant, lgrngn_cond.py:70-73
    orc, hip = h.make_pair(oi, fields)
    assert hip.n_part == 10000
    opts = lgrngn.opts_t()
    opts.sedi = opts.adve = False
    for it in range(3):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-4)
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-7)
        h.copy_state(orc, hip)


def icicle_opts(nx, nz, sd_conc=64, sstp=10):
    dx = 1500. / 75
    oi = lgrngn.opts_init_t()
    oi.nx, oi.ny, oi.nz = nx, 0, nz
    oi.dx = oi.dz = dx
    oi.dy = 1.
    oi.x0, oi.z0 = dx / 2, dx / 2                       # kin_cloud_2d_lgrngn.hpp:167-170
    oi.x1, oi.z1 = (nx - .5) * dx, (nz - .5) * dx
    oi.y1 = 1.
    oi.dt = 1.
    oi.sd_conc = sd_conc
    oi.n_sd_max = int(sd_conc * nx * nz * 1.2)
    oi.dry_distros = {(.61, 0.): h.lgrngn_bimodal()}
    oi.kernel = lgrngn.kernel_t.geometric
    oi.kernel_parameters = [0.5]
    oi.terminal_velocity = lgrngn.vt_t.khvorostyanov_spherical
    oi.adve_scheme = lgrngn.as_t.implicit
    oi.sstp_cond = oi.sstp_coal = sstp
    return oi


def icicle_fields(nx, nz, dtype=np.float64):
    """th = 289 K, rv = 7.5e-3, hydrostatic-like rhod(z); stream function psi = -sin(pi z/Z) cos(2 pi x/X), w_max = 0.6 m/s
    (models/kinematic_2D/cases/icmw8_case1.hpp:166-228), Courant numbers divided by rhod"""
    dx = 1500. / 75
    X, Z = nx * dx, nz * dx
    z = (np.arange(nz) + .5) * dx
    rhod = np.broadcast_to(1.2 * np.exp(-z / 8000.), (nx, nz)).astype(dtype).copy()
    th = np.full((nx, nz), 289., dtype=dtype)
    rv = np.full((nx, nz), 7.5e-3, dtype=dtype)
    A = 0.6 / np.pi * X / (2 * np.pi) * 0 + 0.6 * X / (2 * np.pi)        # scale so that max |w| = 0.6
    xe, ze = np.arange(nx + 1) * dx, np.arange(nz + 1) * dx
    xc, zc = (np.arange(nx) + .5) * dx, (np.arange(nz) + .5) * dx
    u = -A * np.pi / Z * np.cos(np.pi * zc[None, :] / Z) * np.cos(2 * np.pi * xe[:, None] / X)      # -d psi / dz
    w = A * 2 * np.pi / X * np.sin(np.pi * ze[None, :] / Z) * np.sin(2 * np.pi * xc[:, None] / X)    #  d psi / dx
    Cx = (u * 1. / dx / 1.2).astype(dtype)
    Cz = (w * 1. / dx / 1.2).astype(dtype)
    return th, rv, rhod, {"Cx": np.ascontiguousarray(Cx), "Cz": np.ascontiguousarray(Cz)}


def test_c2_icicle_2d_reduced_vs_oracle():
    oi = icicle_opts(24, 24, 32, sstp=10)
    fields = icicle_fields(24, 24)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=2e-4)
        np.testing.assert_allclose(hip.get_attr("x"), orc.get_attr("x"), rtol=1e-13)
        np.testing.assert_allclose(hip.get_attr("z"), orc.get_attr("z"), rtol=1e-13, atol=1e-7)   # dt * vt(rw2 to 1e-4)
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-7)
        h.copy_state(orc, hip)


@pytest.mark.parametrize("strict_fp", [True, False])
def test_c2_icicle_2d_full_size_double_vs_oracle(strict_fp):
    """C2 at its full size -- 76 x 76 cells x 64 = 369 664 super-droplets, sstp_cond = sstp_coal = 10, implicit advection -- in
    double against the oracle, two steps (20 condensation and 20 coalescence substeps), both arithmetic modes"""
    nx = nz = 76
    oi = icicle_opts(nx, nz, 64, sstp=10)
    oi.strict_fp = strict_fp
    fields = icicle_fields(nx, nz)
    orc, hip = h.make_pair(oi, fields)
    assert hip.n_part == nx * nz * 64
    opts = lgrngn.opts_t()
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        rw_h, rw_o = hip.get_attr("rw2"), orc.get_attr("rw2")
        np.testing.assert_allclose(rw_h, rw_o, rtol=5e-4)                 # ten substeps of rtol 1e-4 each at the worst
        assert np.median(np.abs(rw_h / rw_o - 1)) < (1e-9 if strict_fp else 10 * h.cond_bars(False)[2])      # (ten substeps; fast arithmetic: _harness.cond_bars)
        np.testing.assert_allclose(hip.get_attr("x"), orc.get_attr("x"), rtol=1e-13)
        np.testing.assert_allclose(hip.get_attr("z"), orc.get_attr("z"), rtol=1e-13, atol=1e-6)   # dt * vt(rw2)
        # (the two steps are the spin-up of fresh aerosol over ten substeps each: fast arithmetic measured th 1.3e-7, rv 1.3e-6)
        np.testing.assert_allclose(thh, tho, rtol=h.cond_bars(strict_fp)[0])
        np.testing.assert_allclose(rvh, rvo, rtol=h.cond_bars(strict_fp)[1])
        h.copy_state(orc, hip)


@pytest.mark.parametrize("strict_fp", [True, False])
def test_c2_icicle_2d_float_against_the_double_oracle(strict_fp):
    """C2 in the arithmetic icicle runs it in, real_t = float (fig_a/calc.cpp:36-39), at its full size against the oracle ITERATING AS A
    FLOAT BUILD DOES: `orc_set_real_bytes(4)` gives its root finders float's tolerance, eps_tolerance(sizeof(float) * 8 / 4) = 2^-7
    (config.hpp:39, toms748.hpp:445-471) -- the one thing that makes a float run of the reference a different algorithm and not just a
    rounded one (with the double tolerance 2^-15 in the oracle this test measured a median of 2e-3 in rw2; now 1e-4).  The oracle's
    arithmetic stays double.  Step by step from the same state (replayed random streams, state copied back each step); what float's
    rounding changes is bounded by what float is:
      * cell index and sort: a position differs by 1e-7 of the domain, so a droplet within that of a cell face lands next door:
        at most 1e-4 of the droplets (measured 5e-5), and everywhere else the permutation inside a cell is the oracle's;
      * multiplicities under the replayed stream: equal except around those droplets (a different partner) and where a collision
        probability sits within float's resolution of the random number: at most 1e-4 (measured 3e-6);
      * wet radii after ten substeps: the median within 2e-4 (strict; measured 8e-5) / 5e-4 (fast; 2e-4), 99 % within 6e-3, 99.9 %
        within 1e-2 / 2.5e-2: a bracket 2^-7 = 8e-3 wide whose next cut falls on the other side of a 24-bit rounding ends one
        bisection earlier or later, and ten substeps give it ten chances (the tail: a handful of droplets at their critical radius);
      * th, rv: sums of the same changes in 24-bit arithmetic, ten substeps: 2e-6 and 6e-5 (measured 1.4e-6, 3.0e-5);
        positions: 1e-2 m of 1500 (5e-4 of a cell).
    (Round 5: the oracle has a float flavour now -- test_c2_icicle_2d_float_against_the_float_oracle below holds the integers EXACTLY;
    this comparison with the double arithmetic stays as the measure of what float itself costs.)"""
    nx = nz = 76
    oi = icicle_opts(nx, nz, 64, sstp=10)
    oi.strict_fp = strict_fp
    th, rv, rhod, C = icicle_fields(nx, nz, np.float32)
    f64 = (th.astype(np.float64), rv.astype(np.float64), rhod.astype(np.float64), {k: v.astype(np.float64) for k, v in C.items()})
    h.oracle_lib().orc_set_real_bytes(4)               # the oracle's root finders stop at float's tolerance, 2^-7 (config.hpp:39)
    try:
        orc = h.oracle_particles(oi)
        hip = h.hip_particles(oi, np.float32)
        for arr in h.oracle_rng_preview(orc, h.init_replay_calls(oi)):
            hip.rng_replay_push(0, arr)
        orc.init(f64[0].copy(), f64[1].copy(), f64[2].copy(), **f64[3])
        hip.init(th.copy(), rv.copy(), rhod.copy(), **C)
        assert hip.n_part == orc.n_part == nx * nz * 64
        h.copy_state(orc, hip)
        opts = lgrngn.opts_t()
        n_sd = orc.n_part
        for it in range(2):
            tho, rvo, thh, rvh = f64[0].copy(), f64[1].copy(), th.copy(), rv.copy()
            orc.step_sync(opts, tho, rvo, f64[2], **f64[3])
            hip.step_sync(opts, thh, rvh, rhod, **C)
            h.push_coal_replay(orc, hip, oi.sstp_coal)
            orc.step_async(opts)
            hip.step_async(opts)
            np.testing.assert_allclose(thh, tho, rtol=2e-6)
            np.testing.assert_allclose(rvh, rvo, rtol=6e-5)
            assert abs(hip.n_part - orc.n_part) <= 1e-4 * n_sd
            if hip.n_part == orc.n_part:
                ijk_h, ijk_o = hip.state_u64("ijk"), orc.state_u64("ijk")
                moved = ijk_h != ijk_o
                assert moved.mean() < 1e-4, moved.mean()
                n_h, n_o = hip.state_u64("n"), orc.state_u64("n")
                assert (n_h != n_o).mean() < 1e-4, (n_h != n_o).mean()
                same = ~moved & (n_h == n_o)
                err = np.abs(hip.get_attr("rw2").astype(np.float64)[same] / orc.get_attr("rw2")[same] - 1)
                q50, q99, q999 = np.median(err), np.quantile(err, .99), np.quantile(err, .999)
                assert q50 < (2e-4 if strict_fp else 5e-4) and q99 < 6e-3 and q999 < (1e-2 if strict_fp else 2.5e-2), (q50, q99, q999, err.max())
                # (positions: the implicit scheme's x + dx (C_l - i (C_r - C_l)) cancels at i ~ 75: a few dozen ulps of the 1500 m domain;
                # measured 3.4e-3 m at most, 2.7 % of the droplets above 2e-4 m)
                np.testing.assert_allclose(hip.get_attr("x").astype(np.float64)[same], orc.get_attr("x")[same], rtol=0, atol=1e-2)
            h.copy_state(orc, hip)
    finally:
        h.oracle_lib().orc_set_real_bytes(8)


@pytest.mark.parametrize("mode", ["strict", "toms", "fast"])
def test_c2_icicle_2d_float_against_the_float_oracle(mode):
    """Round 5: C2 in real_t = float against the oracle's FLOAT FLAVOUR (oracle/liblcx_oracle_f32.so: the same source with real = float --
    float storage, float libm, float literals, double only where the reference forces it), at full size, step by step from the same
    state under replayed random streams.  What the comparison with the double oracle (above) could only bound statistically is exact
    here: after every full step the SAME cells, the SAME multiplicities, the SAME permutation (sorted_id) and the same x for every one
    of the 3.7e5 super-droplets; the initial multiplicities and dry radii equal bit for bit.  The wet radii carry what two float
    implementations of one root finder at float's tolerance (2^-7 = 7.8e-3, config.hpp:39) carry: identical for a sixth of the
    droplets, the median at 1e-5 ... 1e-6 (strict and cond_solver = 1; the lean solver returns the root instead of the bracket's
    midpoint: 1.5e-4 ... 6e-6), 99 % within 4e-3, 99.9 % within 6e-3 (lean 2.5e-2) -- one bisection of a 2^-7 bracket; th 1e-5, rv 1.5e-4
    in the spin-up step of ten substeps, 2e-6 and 2e-5 afterwards; z within the sedimentation of a drop whose radius differs by that."""
    nx = nz = 76
    oi = icicle_opts(nx, nz, 64, sstp=10)
    oi.strict_fp = mode == "strict"
    oi.cond_solver = 1 if mode == "toms" else 0
    th, rv, rhod, C = icicle_fields(nx, nz, np.float32)
    orc = h.oracle_f32_particles(oi)
    hip = h.hip_particles(oi, np.float32)
    for arr in h.oracle_rng_preview(orc, h.init_replay_calls(oi)):
        hip.rng_replay_push(0, arr)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)
    assert hip.n_part == orc.n_part == nx * nz * 64
    exact(hip.state_u64("n"), orc.state_u64("n"), "initial multiplicities")
    exact(hip.state_real("rd3"), orc.state_real("rd3"), "initial dry radii")
    exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "initial cells")
    assert np.abs(hip.state_real("rw2") / orc.state_real("rw2") - 1).max() < 2 * 2. ** -7       # (equilibrium radii: the root finder's tolerance)
    h.copy_state(orc, hip)
    opts = lgrngn.opts_t()
    lean = mode == "fast"
    for it in range(3):
        tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
        orc.step_sync(opts, tho, rvo, rhod, **C)
        hip.step_sync(opts, thh, rvh, rhod, **C)
        np.testing.assert_allclose(thh, tho, rtol=2e-5 if it == 0 else 4e-6)
        np.testing.assert_allclose(rvh, rvo, rtol=3e-4 if it == 0 else 4e-5)
        err = np.abs(hip.state_real("rw2") / orc.state_real("rw2") - 1)
        q50, q99, q999 = np.median(err), np.quantile(err, .99), np.quantile(err, .999)
        assert q50 < (5e-4 if lean else 1e-4) and q99 < 6e-3 and q999 < (4e-2 if lean else 1.2e-2), (it, q50, q99, q999, err.max())
        h.push_coal_replay(orc, hip, oi.sstp_coal)
        orc.step_async(opts)
        hip.step_async(opts)
        assert hip.n_part == orc.n_part
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(hip.get_attr("x"), orc.get_attr("x"), rtol=0, atol=1e-2 if it == 0 else 0)
        np.testing.assert_allclose(hip.get_attr("z"), orc.get_attr("z"), rtol=0, atol=2e-3)
        h.copy_state(orc, hip)


def test_c2_icicle_2d_full_size_float():
    """76 x 76 x 64 = 369 664 super-droplets in real_t = float as icicle runs it: 20 steps; water is conserved between
    vapour and droplets up to precipitation, the SD count only decreases, the cell-sorted order stays a stable argsort"""
    nx = nz = 76
    oi = icicle_opts(nx, nz)
    th, rv, rhod, C = icicle_fields(nx, nz, np.float32)
    pr = lgrngn.factory(lgrngn.backend_t.HIP, oi, np.float32)
    pr.init(th, rv, rhod, **C)
    assert pr.n_part == nx * nz * 64
    opts = lgrngn.opts_t()

    def liquid():
        pr.diag_all()
        pr.diag_wet_mom(3)
        return (pr.outbuf_array().reshape(nx, nz).astype(np.float64) * 4. / 3 * np.pi * 1e3)
    w0 = float(np.sum((rv + liquid()) * rhod))
    n_prev = pr.n_part
    for it in range(20):
        pr.step_sync(opts, th, rv, rhod, **C)
        pr.step_async(opts)
        assert pr.n_part <= n_prev
        n_prev = pr.n_part
    assert np.isfinite(th).all() and np.isfinite(rv).all()
    w1 = float(np.sum((rv + liquid()) * rhod))
    puddle = pr.diag_puddle()
    assert abs(w1 - w0) <= 2e-4 * w0 + 1e3 * abs(puddle["liquid_volume"])      # float accumulation over 200 cond substeps
    sid, sijk, ijk = pr.state_u64("sorted_id"), pr.state_u64("sorted_ijk"), pr.state_u64("ijk")
    assert np.array_equal(ijk[sid], sijk) and np.all(np.diff(sijk.astype(np.int64)) >= 0)


def test_c4_ring_of_8_slabs_vs_oracle():
    import test_hip_migration as mig
    nx, ny, nz, size = 16, 4, 6, 8
    oi = h.box_opts(nx, ny, nz, 16, dx=20., coal_switch=False)
    oi.n_sd_max = 16 * nx * ny * nz * 2
    fields = h.box_fields(oi)
    orc, hip = mig.ring_pair(oi, size, fields)
    th, rv, rhod, C = fields
    opts = lgrngn.opts_t()
    opts.coal = False
    for it in range(4):
        a = [x.copy() for x in (th, rv, rhod)]
        b = [x.copy() for x in (th, rv, rhod)]
        orc.step(opts, *a, **C)
        hip.step(opts, *b, **C)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-7)
        for r, (po, ph) in enumerate(zip(orc.prts, hip.prts)):
            assert ph.n_part == po.n_part, (it, r)
            for nm in ("n", "ijk", "sorted_id"):
                assert np.array_equal(ph.state_u64(nm), po.state_u64(nm)), (it, r, nm)
            np.testing.assert_allclose(ph.get_attr("x"), po.get_attr("x"), rtol=1e-14, atol=1e-9)
    assert sum(p.n_part for p in hip.prts) == sum(p.n_part for p in orc.prts)


@pytest.mark.parametrize("n,strict_fp", [(8, True), (16, True), (16, False)])
def test_c5_512_sd_per_cell_larger_box_vs_oracle(n, strict_fp):
    """C5 at sizes where the crowded-cell kernels are the ones compared: 8^3 cells x 512 and 16^3 x 512 = 2.1e6 super-droplets
    (fast arithmetic: >= 4096 cells with >= 192 SDs each select k_cond_cellfinish_wave), both arithmetic modes at the larger size.
    After the first step the occupancies spread around 512, so k_cellsort_wave sees segments on both sides of its 513 ... 576 head +
    tail merge path.  EVERY droplet inside the substep tolerance: round 2 allowed 1e-5 of them outside it -- those were droplets
    with the invalid-vt flag of a coalescence in the previous step, see test_cond_step_with_invalid_terminal_velocities"""
    oi = h.box_opts(n, n, n, 512, kernel=lgrngn.kernel_t.hall_pinsky_stratocumulus, strict_fp=strict_fp)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        err = np.abs(hip.get_attr("rw2") / orc.get_attr("rw2") - 1)
        assert err.max() < 1e-4 and np.median(err) < h.cond_bars(strict_fp)[2], (int((err > 1e-4).sum()), np.median(err), err.max())
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-7)
        h.copy_state(orc, hip)
    cnt = np.diff(hip.state_u64("cell_start").astype(np.int64))
    assert cnt.max() > 512 and cnt.min() < 512 and ((cnt > 512) & (cnt <= 576)).any()


@pytest.mark.parametrize("strict_fp", [True, False, "toms"], ids=["strict_fp", "fast_fp", "fast_fp_toms748"])
def test_production_size_paths_vs_oracle(strict_fp):
    """128 x 128 x 32 cells x 64 = 2^25 super-droplets: every kernel in the launch geometry of the headline box (XCD-aware workgroup
    order of the condensation kernel, the LDS-staged per-cell passes at their production cell counts), no environment switch -- two
    full steps (cond + coal + adve + sedi) with replayed random streams against the oracle's OpenMP build (bit-identical to the
    serial one, tests/test_oracle_pins.py), both arithmetic modes: strict = the reference's TOMS748 iterates in IEEE order, fast = the
    collected growth rate under the lean bracketed secant (what bench.py runs).  Integers exact, rw2 at the substep tolerance, th / rv
    at SURVEY 8a's bars.  (Measured, strict / fast: rw2 max 3.0e-5 / 2.3e-5, th 1.3e-11 / 4.3e-11, rv 2.5e-10 / 8.1e-10.)"""
    import bench
    nx, ny, nz = 128, 128, 32
    oi = bench.make_opts_init(nx, ny, nz, 64, 40., 1, 1, 44)
    toms = strict_fp == "toms"                            # (round 4: the reference's iterates in fast arithmetic, held to the strict bars)
    oi.strict_fp = strict_fp is True
    oi.cond_solver = int(toms)
    strict_fp = strict_fp is True or toms
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(nx, ny, nz, 0, nx, np, np.float64)
    fields = (th, rv, rhod, {"Cx": Cx, "Cy": Cy, "Cz": Cz})
    orc, hip = h.make_pair(oi, fields, make_oracle=h.oracle_omp_particles)
    assert hip.n_part >= (1 << 25) - 64
    opts = lgrngn.opts_t()
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        # (read back from the object: the arithmetic, the solver and the kernel that the parameter stands for; a replayed run walks the sorted order)
        h.assert_mode(hip, oi.strict_fp, oi.cond_solver, ("strict",) if oi.strict_fp else ("fold_toms748", "lean_toms748_sorted") if toms else ("lean", "lean_sorted"))
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("ijk"), orc.state_u64("ijk"), "ijk")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        err = np.abs(hip.get_attr("rw2") / orc.get_attr("rw2") - 1)
        assert err.max() < 1e-4 and np.median(err) < h.cond_bars(strict_fp)[2], (int((err > 1e-4).sum()), np.median(err), err.max())
        for a_ in ("x", "y"):
            np.testing.assert_allclose(hip.get_attr(a_), orc.get_attr(a_), rtol=1e-14)
        # sedimentation moves a droplet by dt * vt(rw2): rw2 to 1e-4 is vt to 1e-4 (measured: 24 of 3.4e7 droplets above 1e-7 m, 4.4e-6 m at most)
        np.testing.assert_allclose(hip.get_attr("z"), orc.get_attr("z"), rtol=1e-13, atol=2e-5)
        np.testing.assert_allclose(thh, tho, rtol=h.cond_bars(True)[0] if strict_fp else 1e-7)
        np.testing.assert_allclose(rvh, rvo, rtol=1e-7)
        h.copy_state(orc, hip)


def test_c5_512_sd_per_cell_vs_oracle():
    oi = h.box_opts(3, 2, 3, 512, kernel=lgrngn.kernel_t.hall_pinsky_stratocumulus)
    fields = h.box_fields(oi)
    orc, hip = h.make_pair(oi, fields)
    opts = lgrngn.opts_t()
    for it in range(2):
        (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, fields)
        exact(hip.state_u64("n"), orc.state_u64("n"), "n")
        exact(hip.state_u64("sorted_id"), orc.state_u64("sorted_id"), "sorted_id")
        np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=2e-4)
        np.testing.assert_allclose(thh, tho, rtol=1e-7)
        h.copy_state(orc, hip)
