"""Integer-exact API invariants of the reference's tests/python/unit/api_lgrngn.py, for the init modes this backend
supports (sd_conc distros, dry_sizes, both together), in 0-D ... 3-D.  Run against the oracle on CPU and against the HIP
backend on the GPU."""
import numpy as np
import pytest
from numpy import frombuffer, isclose

import _harness as h
from libcloudphxx_amd import lgrngn

rho_stp = 1.2248          # api_lgrngn.py:28
kappa1, kappa2, kappa3, rd_insol = .61, 1.28, 0.8, 0.


def base_opts():
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(kappa1, rd_insol): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.kernel = lgrngn.kernel_t.geometric
    oi.terminal_velocity = lgrngn.vt_t.beard76
    oi.dt = 1
    oi.sd_conc = 64
    oi.n_sd_max = int(1e6)
    oi.sedi_switch = False
    return oi


MAKERS = [pytest.param(h.oracle_particles, id="oracle"), pytest.param(h.hip_particles, id="hip", marks=pytest.mark.gpu)]
th0, rv0, rhod0 = np.array([300.]), np.array([.01]), np.array([1.])


@pytest.mark.parametrize("make", MAKERS)
def test_0d_sd_conc_and_call_order(make):
    """api_lgrngn.py:110-190"""
    oi = base_opts()
    pr = make(oi)
    opts = lgrngn.opts_t()
    opts.sedi = opts.adve = False
    with pytest.raises(RuntimeError):
        pr.step_sync(opts, th0, rv0, rhod0)
    pr.init(th0.copy(), rv0.copy(), rhod0.copy())
    with pytest.raises(RuntimeError):
        pr.step_async(opts)
    th, rv = th0.copy(), rv0.copy()
    pr.step_sync(opts, th, rv, rhod0)
    with pytest.raises(RuntimeError):
        pr.step_sync(opts, th, rv, rhod0)
    pr.step_async(opts)
    with pytest.raises(RuntimeError):
        pr.step_async(opts)
    pr.diag_all()
    pr.diag_sd_conc()
    assert len(frombuffer(pr.outbuf())) == 1
    assert frombuffer(pr.outbuf())[0] == oi.sd_conc
    with pytest.raises(RuntimeError):                 # Courant numbers in 0-D
        pr2 = make(base_opts())
        pr2.init(th0.copy(), rv0.copy(), rhod0.copy(), Cx=np.zeros(1))


@pytest.mark.parametrize("make", MAKERS)
def test_0d_dry_sizes_two_kappas(make):
    """api_lgrngn.py:243-301: exact counts 75 / 50 / 30 / 20 / 15 / 10 and kappa moments"""
    oi = base_opts()
    oi.dry_distros = dict()
    oi.dry_sizes = {(kappa1, rd_insol): {1.e-6: [30. * rho_stp, 15], 15.e-6: [10. * rho_stp, 10]},
                    (kappa2, rd_insol): {1.2e-6: [20. * rho_stp, 10], 12.e-6: [15. * rho_stp, 15]}}
    oi.sd_conc = 0
    pr = make(oi)
    pr.init(th0.copy(), rv0.copy(), rhod0.copy())
    pr.diag_all()
    pr.diag_sd_conc()
    sd_tot = frombuffer(pr.outbuf()).sum()
    pr.diag_all()
    pr.diag_wet_mom(0)
    assert frombuffer(pr.outbuf()).sum() == 75.
    assert sd_tot == 50.
    for (lo, hi, n_exp, kap) in ((1e-6, 1.1e-6, 30, kappa1), (1.2e-6, 1.3e-6, 20, kappa2), (12e-6, 13e-6, 15, kappa2), (15e-6, 15.1e-6, 10, kappa1)):
        pr.diag_dry_rng(lo, hi)
        pr.diag_wet_mom(0)
        n = frombuffer(pr.outbuf()).copy()
        pr.diag_kappa_mom(1)
        k = frombuffer(pr.outbuf())
        assert (n == n_exp).all()
        assert isclose(k, n * kap, rtol=1e-15)


@pytest.mark.parametrize("make", MAKERS)
def test_0d_dry_sizes_plus_sd_conc(make):
    """api_lgrngn.py:308-323, 683-697: 64 from the distro + 20 from sizes; attributes ordered distro first, then sizes"""
    oi = base_opts()
    oi.dry_sizes = {(kappa3, rd_insol): {1.e-6: [30. * rho_stp, 15], 15.e-6: [10. * rho_stp, 5]}}
    pr = make(oi)
    pr.init(th0.copy(), rv0.copy(), rhod0.copy())
    pr.diag_all()
    pr.diag_sd_conc()
    assert frombuffer(pr.outbuf())[0] == 84
    kap = pr.get_attr("kappa")
    assert (kap[:64] == kappa1).all() and (kap[64:] == kappa3).all()
    rd3 = pr.get_attr("rd3")
    assert np.allclose(rd3[64:79], 1e-18, rtol=1e-15) and np.allclose(rd3[79:], (15e-6) ** 3, rtol=1e-15)


@pytest.mark.parametrize("make", MAKERS)
@pytest.mark.parametrize("dims", [(3, 0, 0), (2, 0, 2), (2, 2, 2)])
def test_nd_sd_conc_is_conserved_by_advection(make, dims):
    """api_lgrngn.py:400-565: sum of sd_conc == n_cell * 64 after steps with advection in 1-D / 2-D / 3-D"""
    nx, ny, nz = dims
    oi = base_opts()
    oi.nx, oi.ny, oi.nz = nx, ny, nz
    oi.dx = oi.dy = oi.dz = 10
    oi.x1, oi.y1, oi.z1 = max(nx, 1) * 10, max(ny, 1) * 10, max(nz, 1) * 10
    if ny == 0:
        oi.dy, oi.y1 = 1, 1
    if nz == 0:
        oi.dz, oi.z1 = 1, 1
    shp = tuple(n for n in (nx, ny, nz) if n > 0)
    th, rv, rhod = 300. * np.ones(shp), .01 * np.ones(shp), np.ones(shp)
    C = {}
    C["Cx"] = 0.5 * np.ones((nx + 1,) + shp[1:])
    if ny:
        C["Cy"] = 0.2 * np.ones((nx, ny + 1, nz))
    if nz:
        C["Cz"] = np.zeros(shp[:-1] + (nz + 1,))
    pr = make(oi)
    pr.init(th, rv, rhod, **C)
    opts = lgrngn.opts_t()
    opts.sedi = opts.coal = False
    for _ in range(3):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    pr.diag_all()
    pr.diag_sd_conc()
    assert frombuffer(pr.outbuf()).sum() == 64 * int(np.prod(shp))


def two_distros_opts():
    """api_lgrngn.py:20-52: two lognormal modes, each n_tot = 60e6 per STP m^3"""
    oi = base_opts()
    oi.dry_distros = {(kappa1, rd_insol): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6), (kappa2, rd_insol): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    return oi


@pytest.mark.parametrize("make", MAKERS)
def test_0d_large_tail(make):
    """api_lgrngn.py:188-206: sd_conc_large_tail adds multiplicity-1 SDs beyond the sampled range"""
    oi = two_distros_opts()
    oi.sd_conc_large_tail = True
    pr = make(oi)
    th, rv = th0.copy(), rv0.copy()
    pr.init(th, rv, rhod0.copy())
    opts = lgrngn.opts_t()
    opts.sedi = opts.adve = False
    pr.step_sync(opts, th, rv)
    pr.step_async(opts)
    pr.diag_all()
    pr.diag_sd_conc()
    out = frombuffer(pr.outbuf())
    assert len(out) == 1 and (out > 0).all() and out.sum() >= oi.sd_conc


@pytest.mark.parametrize("make", MAKERS)
def test_0d_const_multi(make):
    """api_lgrngn.py:209-238: every SD carries sd_const_multi particles, before and after two steps (coalescence on)"""
    oi = two_distros_opts()
    oi.sd_conc = 0
    prtcls_per_cell = 2 * 60e6 / rho_stp            # rhod = 1; two distributions
    oi.sd_const_multi = int(prtcls_per_cell / 64)
    pr = make(oi)
    th, rv = th0.copy(), rv0.copy()
    pr.init(th, rv, rhod0.copy())
    pr.diag_all()
    pr.diag_sd_conc()
    sd0 = frombuffer(pr.outbuf()).sum()
    assert abs(sd0 - 64) <= 1                       # 2 x int(integral / const_multi + .5)
    opts = lgrngn.opts_t()
    opts.sedi = opts.adve = False
    pr.step_sync(opts, th, rv)
    pr.step_async(opts)
    pr.step_sync(opts, th, rv, rhod0)
    pr.step_async(opts)
    pr.diag_all()
    pr.diag_sd_conc()
    out = frombuffer(pr.outbuf())
    assert len(out) == 1 and (out > 0).all()
    sd_tot = out.sum()
    pr.diag_all()
    pr.diag_wet_mom(0)
    prtcls_tot = frombuffer(pr.outbuf()).sum()
    assert prtcls_tot / sd_tot == oi.sd_const_multi


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["const_multi", "tail"])
def test_const_multi_init_matches_oracle(mode):
    """constant-multiplicity / large-tail initialisation on a 3-D box against the oracle with its random stream replayed:
    same per-cell counts, cells, dry radii (CDF look-up), multiplicities and positions"""
    kw = dict(sd_const_multi=int(1.5e11)) if mode == "const_multi" else dict(sd_conc_large_tail=True)
    sdc = 256                    # fine bins: the sampled range ends early and the multiplicity-1 tail is not empty
    oi = h.box_opts(3, 2, 4, 0 if mode == "const_multi" else sdc, **kw)
    oi.n_sd_max = 200000
    th, rv, rhod, C = h.box_fields(oi)
    probe = h.oracle_particles(oi)
    probe.init(th.copy(), rv.copy(), rhod.copy(), **C)
    n_tot = probe.n_part
    n1 = sdc * 24 if mode == "tail" else 0
    calls = ([(0, n1)] * 4 if n1 else []) + [(0, n_tot - n1)] * 4
    orc, hip = h.oracle_particles(oi), h.hip_particles(oi)
    for arr in h.oracle_rng_preview(orc, calls):
        hip.rng_replay_push(0, arr)
    orc.init(th.copy(), rv.copy(), rhod.copy(), **C)
    hip.init(th.copy(), rv.copy(), rhod.copy(), **C)
    assert hip.n_part == orc.n_part == n_tot and n_tot > n1 + 24
    for nm in ("n", "ijk", "sorted_id"):
        assert np.array_equal(hip.state_u64(nm), orc.state_u64(nm)), nm
    np.testing.assert_allclose(hip.get_attr("rd3"), orc.get_attr("rd3"), rtol=1e-12)
    np.testing.assert_allclose(hip.get_attr("rw2"), orc.get_attr("rw2"), rtol=1e-4)
    for a in ("x", "y", "z"):
        np.testing.assert_allclose(hip.get_attr(a), orc.get_attr(a), rtol=1e-14)


@pytest.mark.parametrize("make", MAKERS)
@pytest.mark.parametrize("vt", [lgrngn.vt_t.beard76, lgrngn.vt_t.beard77fast, lgrngn.vt_t.khvorostyanov_spherical])
def test_sd_removal_with_recycling(make, vt):
    """tests/python/unit/SD_removal.py (without its chemistry): 900 steps of pure coalescence with opts.rcyc -- recycling
    splits the biggest SDs into the freed slots, so in the end every surviving SD has multiplicity 1"""
    def expvolumelnr(lnr):
        r_zero, n_zero = 30.531e-6, 2 ** 8
        r = np.exp(lnr)
        return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(-np.power((r / r_zero), 3))
    oi = lgrngn.opts_init_t()
    oi.dt = 2 ** 15
    oi.sstp_coal = 1
    oi.dry_distros = {(.01, 0.): expvolumelnr}
    oi.sd_conc = oi.n_sd_max = 64
    oi.sedi_switch = False
    oi.kernel = lgrngn.kernel_t.geometric
    oi.terminal_velocity = vt
    pr = make(oi)
    th, rv, rhod = 300. * np.ones(1), .01 * np.ones(1), np.ones(1)
    pr.init(th, rv, rhod)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = False
    opts.rcyc = True
    for _ in range(900):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    pr.diag_all()
    pr.diag_sd_conc()
    sd_conc = frombuffer(pr.outbuf())[0]
    assert 0 < sd_conc <= 10
    pr.diag_all()
    pr.diag_wet_mom(0)
    assert frombuffer(pr.outbuf())[0] == sd_conc


@pytest.mark.parametrize("make", MAKERS)
def test_turb_adve_moves_super_droplets(make):
    """tests/python/unit/lgrngn_turb_adve.py: no resolved flow, only the SGS velocity perturbations -- after 100 steps the
    SD counts per cell have changed; the perturbation velocities have the stationary variance 2/3 TKE"""
    oi = lgrngn.opts_init_t()
    oi.dt = 1
    oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.coal_switch = oi.sedi_switch = False
    oi.turb_adve_switch = True
    oi.nx, oi.nz, oi.dx, oi.dz = 10, 10, 1, 1
    oi.x1 = oi.z1 = 10
    oi.SGS_mix_len = np.ones(10)
    oi.sd_conc = 100
    oi.n_sd_max = 100 * 100
    pr = make(oi)
    th, rv, rhod = 300. * np.ones((10, 10)), .01 * np.ones((10, 10)), np.ones((10, 10))
    diss = 1e-4 * np.ones((10, 10))
    pr.init(th, rv, rhod)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = opts.coal = False
    opts.turb_adve = True
    pr.diag_all(); pr.diag_sd_conc()
    tab_in = pr.outbuf_array().copy()
    for _ in range(100):
        pr.step_sync(opts, th, rv, rhod, diss_rate=diss)
        pr.step_async(opts)
    pr.diag_all(); pr.diag_sd_conc()
    assert not np.array_equal(tab_in, pr.outbuf_array())
    sigma = np.sqrt(2. / 3 * (1e-4 / 0.845) ** (2. / 3))
    for comp in ("up", "wp"):
        assert abs(np.sqrt(np.mean(pr.state_real(comp) ** 2)) / sigma - 1) < 0.05


KERNELS = ["geometric", "geometric_x10", "Long", "hall", "hall_davis_no_waals", "golovin", "onishi_hall",
           "onishi_hall_davis_no_waals", "vohl_davis_no_waals", "hall_pinsky_cumulonimbus", "hall_pinsky_stratocumulus"]


@pytest.mark.parametrize("make", MAKERS)
@pytest.mark.parametrize("kernel", KERNELS)
def test_every_collision_kernel_steps(make, kernel):
    """tests/python/unit/col_kernels.py:29-79: one coalescence step with each kernel of kernel_t; the Onishi kernels need
    turb_coal_switch, one parameter (Re_lambda) and diss_rate"""
    oi = lgrngn.opts_init_t()
    oi.dt = 1
    oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.sd_conc = 50
    oi.n_sd_max = 50
    oi.terminal_velocity = lgrngn.vt_t.beard76
    oi.kernel = getattr(lgrngn.kernel_t, kernel.replace("_x10", ""))
    oi.sedi_switch = False
    onishi = kernel.startswith("onishi")
    if onishi:
        oi.turb_coal_switch = True
        oi.kernel_parameters = np.array([100.])
    if kernel == "golovin":
        oi.kernel_parameters = np.array([1.])
    if kernel == "geometric_x10":
        oi.kernel_parameters = np.array([10.])
    pr = make(oi)
    th, rv, rhod = 300. * np.ones(1), .01 * np.ones(1), np.ones(1)
    pr.init(th, rv, rhod)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = False
    opts.coal = True
    if onishi:
        opts.turb_coal = True
        pr.step_sync(opts, th, rv, rhod, diss_rate=.04 * np.ones(1))
    else:
        pr.step_sync(opts, th, rv, rhod)
    pr.step_async(opts)
    pr.diag_all(); pr.diag_wet_mom(3)
    assert np.isfinite(pr.outbuf_array()).all() and pr.outbuf_array()[0] > 0


@pytest.mark.parametrize("make", MAKERS)
def test_onishi_kernel_option_checks(make):
    """init_kernel.ipp:185-190,210-215; particles_step.ipp:74-78"""
    oi = base_opts()
    oi.kernel = lgrngn.kernel_t.onishi_hall
    oi.kernel_parameters = np.array([66.])
    with pytest.raises(RuntimeError, match="turb_coal_switch=True"):
        make(oi).init(th0, rv0, rhod0)
    oi.turb_coal_switch = True
    oi.kernel_parameters = np.array([])
    with pytest.raises(RuntimeError, match="Taylor microscale Reynolds number"):
        make(oi).init(th0, rv0, rhod0)
    oi.kernel_parameters = np.array([66.])
    pr = make(oi)
    pr.init(th0, rv0, rhod0)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = False
    with pytest.raises(RuntimeError, match="diss_rate is empty"):
        pr.step_sync(opts, th0.copy(), rv0.copy(), rhod0.copy())


@pytest.mark.parametrize("make", MAKERS)
def test_onishi_speeds_up_rain_formation(make):
    """tests/python/physics/coalescence_onishi_hall.py:20-102 (switched off in the reference's CMake list as 'not specific
    enough'; its set-up needs (kappa, rd_insol) keys, sedi_switch = False and turb_coal_switch = True to run at all): time to
    turn 10 % of the water into r > 40 um drops, Hall / Onishi-Hall ratio inside the reference's band 1.22 ... 1.62."""
    r_zero, n_zero, n_runs = 15e-6, 1.42e8, 24

    def expvolumelnr(lnr):
        r = np.exp(lnr)
        return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(-np.power((r / r_zero), 3))
    th, rv, rhod, diss = 300 * np.ones(1), 1. * np.ones(1), 1.22419 * np.ones(1), 0.04 * np.ones(1)
    t10 = np.zeros((2, n_runs))
    for k in range(2):
        for zz in range(n_runs):
            oi = lgrngn.opts_init_t()
            oi.dt = 1.
            oi.terminal_velocity = lgrngn.vt_t.beard77fast
            oi.sd_conc = oi.n_sd_max = 1024
            oi.sedi_switch = False
            oi.dry_distros = {(0., 0.): expvolumelnr}
            oi.kernel = lgrngn.kernel_t.hall
            if k == 1:
                oi.kernel = lgrngn.kernel_t.onishi_hall
                oi.kernel_parameters = np.array([66.])
                oi.turb_coal_switch = True
            oi.rng_seed = 1000 + zz + 100 * k
            opts = lgrngn.opts_t()
            opts.adve = opts.sedi = opts.cond = False
            opts.turb_coal = k == 1
            pr = make(oi)
            pr.init(th, rv, rhod)
            pr.diag_all(); pr.diag_wet_mom(3)
            total = pr.outbuf_array().mean()
            t = 0
            while t10[k][zz] == 0:
                if k == 0:
                    pr.step_sync(opts, th, rv, rhod)
                else:
                    pr.step_sync(opts, th, rv, rhod, diss_rate=diss)
                pr.step_async(opts)
                t += 1
                pr.diag_wet_rng(40e-6, 1); pr.diag_wet_mom(3)
                if pr.outbuf_array().mean() > total / 10.:
                    t10[k][zz] = t
                assert t < 2000
    ratio = t10[0].mean() / t10[1].mean()
    # standard error of the ratio over n_runs: the oracle's seeded mt19937 stream is deterministic (strict band); the device's
    # Philox stream is another sample of the same distribution, given two standard errors of slack
    se = ratio * np.sqrt(((t10[0].std() / t10[0].mean()) ** 2 + (t10[1].std() / t10[1].mean()) ** 2) / n_runs)
    slack = 0. if make is h.oracle_particles else 2 * se
    print("t10 Hall / Onishi-Hall = %.3f +- %.3f" % (ratio, se))
    assert 1.22 - slack < ratio < 1.62 + slack, ratio


@pytest.mark.parametrize("make", MAKERS)
@pytest.mark.parametrize("dims", [(5, 0, 0), (4, 0, 6), (4, 3, 5)])
def test_diag_vel_div_is_the_courant_divergence(make, dims):
    """particles_diag.ipp:499-555: per cell, (C_y differences, then C_z, then C_x) / opts_init.dt -- the same additions in numpy;
    pred_corr adds a Courant halo of two planes that the face indices have to skip"""
    for scheme in (lgrngn.as_t.euler, lgrngn.as_t.pred_corr):
        oi = h.box_opts(*dims, 8, sedi_switch=dims[2] > 0, adve_scheme=scheme)
        th, rv, rhod, C = h.box_fields(oi)
        pr = make(oi)
        pr.init(th, rv, rhod, **C)
        pr.diag_vel_div()
        div = 0
        if dims[1]:
            div = div + (C["Cy"][:, 1:, :] - C["Cy"][:, :-1, :]) / oi.dt
        if dims[2]:
            div = div + (C["Cz"][..., 1:] - C["Cz"][..., :-1]) / oi.dt
        div = div + (C["Cx"][1:] - C["Cx"][:-1]) / oi.dt
        np.testing.assert_array_equal(pr.outbuf_array(), div.ravel())


@pytest.mark.parametrize("make", MAKERS)
@pytest.mark.parametrize("layout", ["padded", "kij"])
def test_strided_eulerian_arrays(make, layout):
    """arrinfo_t strides (arrinfo.hpp:11-49, init_e2l.ipp:59-102): arrays that are views into padded storage, or stored with
    the two horizontal axes swapped (kij), give the same run as contiguous ones -- and th / rv are written back through the strides"""
    oi = h.box_opts(4, 3, 5, 16, coal_switch=False)
    th, rv, rhod, C = h.box_fields(oi)

    def view(a):
        if layout == "padded":
            big = np.full(tuple(s + 2 for s in a.shape), np.nan)
            v = big[1:-1, 1:-1, :-2]
            assert not v.flags["C_CONTIGUOUS"] and v.strides[2] == a.itemsize
        else:
            v = np.empty((a.shape[1], a.shape[0], a.shape[2])).transpose(1, 0, 2)        # memory order j, i, k
            assert v.strides[0] < v.strides[1]
        v[...] = a
        return v
    res = []
    for strided in (False, True):
        f = (lambda a: view(a)) if strided else (lambda a: a.copy())
        a_th, a_rv, a_rhod = f(th), f(rv), f(rhod)
        Cs = {k: f(v) for k, v in C.items()}
        pr = make(oi)
        pr.init(a_th, a_rv, a_rhod, **Cs)
        opts = lgrngn.opts_t()
        opts.coal = False
        for _ in range(2):
            pr.step_sync(opts, a_th, a_rv, a_rhod, **Cs)
            pr.step_async(opts)
        res.append((np.array(a_th), np.array(a_rv), pr.get_attr("rw2"), pr.get_attr("x"), pr.get_attr("z")))
    assert not np.array_equal(res[0][0], th)                       # condensation did write th back
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["contiguous", "padded", "kij", "pred_corr"])
def test_host_arrays_through_the_staged_rows_equal_the_plain_host_loop(layout):
    """Host arrays (what an unchanged icicle / UWLCM passes) go through page-locked staging filled row by row by a few host threads and
    one asynchronous copy per field (lcx_core.hip, host_copy_rows); opts_init.dbg_flags & HOST_SYNC_LOOP selects the single-threaded
    element loop of rounds 1-3.  Same run bit for bit, on a box big enough for the thread pool (32 x 24 x 48 cells: 295 KB per field)
    -- contiguous arrays, views into padded storage (libmpdata++'s halos), swapped horizontal axes (the innermost stride stays 1),
    and the wrapped Courant halo of pred_corr"""
    oi = h.box_opts(32, 24, 48, 4, coal_switch=False)
    if layout == "pred_corr":
        oi.adve_scheme = lgrngn.as_t.pred_corr
    th, rv, rhod, C = h.box_fields(oi)

    def view(a):
        if layout == "padded":
            big = np.full(tuple(s + 3 for s in a.shape), np.nan)
            v = big[1:-2, 2:-1, 1:-2]
        elif layout == "kij":
            v = np.empty((a.shape[1], a.shape[0], a.shape[2])).transpose(1, 0, 2)
        else:
            return a.copy()
        v[...] = a
        return v
    res = []
    for loop in (False, True):
        oi.dbg_flags = int(lgrngn.dbg.HOST_SYNC_LOOP) if loop else 0
        a_th, a_rv, a_rhod = view(th), view(rv), view(rhod)
        Cs = {k: view(v) for k, v in C.items()}
        pr = h.hip_particles(oi)
        pr.init(a_th, a_rv, a_rhod, **Cs)
        opts = lgrngn.opts_t()
        opts.coal = False
        for _ in range(3):
            pr.step_sync(opts, a_th, a_rv, a_rhod, **Cs)
            pr.step_async(opts)
        res.append((np.array(a_th), np.array(a_rv), pr.get_attr("rw2"), pr.get_attr("x"), pr.get_attr("z"), pr.state_real("courant_x"), pr.state_real("courant_z")))
    assert not np.array_equal(res[0][0], th)
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("host_dirs", ["z", "xy", "x"])
def test_courant_arrays_partly_on_the_host_partly_on_the_device(host_dirs):
    """A combined step_sync whose Courant arrays live on BOTH sides: the host ones are gathered while condensation runs
    (late_courants), and the device ones of the same call must be launched there too -- they were only listed, after step_cond's one
    launch of the listed jobs had gone, so that step_async advected with the PREVIOUS step's numbers for those directions.  The Courant
    numbers change from step to step here; the run must equal the all-host run bit for bit."""
    import torch
    oi = h.box_opts(12, 10, 14, 8, coal_switch=False, strict_fp=False)
    th, rv, rhod, C = h.box_fields(oi)
    res = []
    for mixed in (False, True):
        pr = h.hip_particles(oi)
        a_th, a_rv = th.copy(), rv.copy()
        pr.init(a_th, a_rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = False
        keep = []
        for step in range(4):
            Cs = {k: np.ascontiguousarray(v * (1.0 - 0.4 * step)) for k, v in C.items()}
            args = dict(Cs)
            if mixed:
                for k in list(args):
                    if k[1] not in host_dirs:
                        t = torch.tensor(args[k], device="cuda")
                        keep.append(t)
                        args[k] = lgrngn.DeviceArray(t.data_ptr(), t.shape)
                torch.cuda.synchronize()
            pr.step_sync(opts, a_th, a_rv, rhod, **args)
            pr.step_async(opts)
        res.append((a_th.copy(), pr.get_attr("x"), pr.get_attr("y"), pr.get_attr("z"), pr.state_real("courant_x"), pr.state_real("courant_y"),
                    pr.state_real("courant_z")))
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_stream_ordered_step_sync_gives_the_same_run():
    """opts_init.stream_ordered (extension): with DEVICE arrays step_sync returns once its work is queued on the object's stream -- the
    caller orders its own stream behind lcx_stream() instead of the host waiting.  Same run bit for bit as the waiting form: th and rv
    read after a device synchronisation, the droplets through get_attr (which synchronises by itself); a call with HOST arrays waits
    whatever the option says."""
    import torch
    oi = h.box_opts(16, 12, 20, 16, strict_fp=False)
    th, rv, rhod, C = h.box_fields(oi)
    res = []
    for so in (False, True):
        oi.stream_ordered = so
        t = {k: torch.tensor(v, device="cuda") for k, v in dict(th=th, rv=rv, rhod=rhod, **C).items()}
        torch.cuda.synchronize()
        d = {k: lgrngn.DeviceArray(v.data_ptr(), v.shape) for k, v in t.items()}
        pr = h.hip_particles(oi)
        pr.init(d["th"], d["rv"], d["rhod"], Cx=d["Cx"], Cy=d["Cy"], Cz=d["Cz"])
        opts = lgrngn.opts_t()
        for _ in range(4):
            pr.step_sync(opts, d["th"], d["rv"], d["rhod"], Cx=d["Cx"], Cy=d["Cy"], Cz=d["Cz"])
            pr.step_async(opts)
        torch.cuda.synchronize()
        res.append((t["th"].cpu().numpy(), t["rv"].cpu().numpy(), pr.get_attr("rw2"), pr.get_attr("x"), pr.state_u64("n")))
        if so:                                       # host arrays through the same object: the call waits, the arrays are written on return
            th_h, rv_h = res[-1][0].copy(), res[-1][1].copy()
            pr.step_sync(opts, th_h, rv_h, rhod, **C)
            assert not np.array_equal(th_h, res[-1][0])
            pr.step_async(opts)
    assert not np.array_equal(res[0][0], th)
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_stream_ordered_caller_rewrites_its_arrays_on_its_own_stream():
    """The other direction of opts_init.stream_ordered's contract (include/lcx.h): between two steps the caller -- a
    dynamical core that lives on the GPU -- rewrites th, rv and the Courant numbers on a stream of ITS OWN, behind the library's results
    (its stream waits for lcx_stream), and makes lcx_stream() wait for that work before the next step_sync instead of synchronising the
    device.  The run equals the one whose caller synchronises the device around every call."""
    import torch
    oi = h.box_opts(16, 12, 20, 16, strict_fp=False, stream_ordered=True)
    th, rv, rhod, C = h.box_fields(oi)
    res = []
    for events_only in (False, True):
        t = {k: torch.tensor(v, device="cuda") for k, v in dict(th=th, rv=rv, rhod=rhod, **C).items()}
        torch.cuda.synchronize()
        d = {k: lgrngn.DeviceArray(v.data_ptr(), v.shape) for k, v in t.items()}
        pr = h.hip_particles(oi)
        pr.init(d["th"], d["rv"], d["rhod"], Cx=d["Cx"], Cy=d["Cy"], Cz=d["Cz"])
        lib_stream = torch.cuda.ExternalStream(pr.stream())
        mine = torch.cuda.Stream()
        opts = lgrngn.opts_t()
        for it in range(4):
            pr.step_sync(opts, d["th"], d["rv"], d["rhod"], Cx=d["Cx"], Cy=d["Cy"], Cz=d["Cz"])
            pr.step_async(opts)
            if events_only:
                mine.wait_stream(lib_stream)              # the caller's stream behind the library's results
            else:
                torch.cuda.synchronize()
            with torch.cuda.stream(mine):                 # the "dynamical core": a big, slow rewrite of the fields on the caller's stream
                for _ in range(20):
                    t["th"].mul_(1.0 + 1e-6).add_(1e-5)
                t["rv"].mul_(1.0 - 1e-5)
                t["Cx"].mul_(0.97); t["Cz"].mul_(1.01)
            if events_only:
                lib_stream.wait_stream(mine)              # ... and the library's stream behind the rewrite: no host wait anywhere
            else:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res.append((t["th"].cpu().numpy(), t["rv"].cpu().numpy(), pr.get_attr("rw2"), pr.get_attr("x"), pr.get_attr("z"), pr.state_u64("n")))
    assert not np.array_equal(res[0][0], th)
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("make,alloc", [pytest.param(h.oracle_particles, h.host_alloc, id="oracle"),
                                        pytest.param(h.hip_particles, h.dev_alloc, id="hip", marks=pytest.mark.gpu)])
def test_slabs_indexing_the_global_arrays(make, alloc):
    """The reference's multi_CUDA hands every GPU the GLOBAL Eulerian arrays and an offset of n_x_bfr planes
    (init_e2l.ipp:44-46, distmem_opts.hpp:27); slabs that read / write their part of the global arrays through
    opts_init.n_x_bfr / n_x_tot must run exactly like slabs that are given their own slices."""
    from libcloudphxx_amd import multi
    oi = h.box_opts(7, 3, 4, 16, coal_switch=False)
    oi.n_sd_max = 16 * 7 * 3 * 4 * 3
    th, rv, rhod, C = h.box_fields(oi)
    size = 3
    opts = lgrngn.opts_t()
    opts.coal = False
    opts.adve = opts.sedi = False                # no exchange: this is about the array indexing only
    res = []
    for glob in (False, True):
        out = []
        thg, rvg = th.copy(), rv.copy()
        for r in range(size):
            o, bfr = multi.distmem_opts(oi, r, size)
            o.rng_seed = oi.rng_seed + r
            if not glob:
                o.n_x_bfr, o.n_x_tot = 0, o.nx
                sl = lambda a, ext=0: np.ascontiguousarray(a[bfr:bfr + o.nx + ext])
                a_th, a_rv = sl(thg), sl(rvg)
                args = (a_th, a_rv, sl(rhod))
                kw = dict(Cx=sl(C["Cx"], 1), Cy=sl(C["Cy"]), Cz=sl(C["Cz"]))
            else:
                o.n_x_bfr, o.n_x_tot = bfr, oi.nx
                a_th, a_rv = thg, rvg
                args = (thg, rvg, rhod)
                kw = dict(C)
            pr = make(o)
            pr.init(*args, **kw)
            pr.step_sync(opts, *args, **kw)
            # with adve and sedi off nobody leaves, but a decomposed object still expects the exchange protocol to be completed
            pr.step_async(opts)
            nl, nr = pr.migrate_counts()
            assert nl == nr == 0
            pr.migrate_finish(opts)
            out.append((pr.get_attr("rw2"), pr.get_attr("x")))
            if not glob:
                thg[bfr:bfr + o.nx] = a_th
                rvg[bfr:bfr + o.nx] = a_rv
        res.append((out, thg.copy(), rvg.copy()))
    for (a, b) in zip(res[0][0], res[1][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert not np.array_equal(res[1][1], th)
