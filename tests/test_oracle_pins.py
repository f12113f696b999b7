"""Pins the CPU oracle (oracle/) against the reference's own golden data and known answers.

The reference cannot be built in this image (Boost absent), so these are what anchor the oracle:
  * tests/golden/lgrngn_cond_substepping_refdata.csv -- the reference's
    tests/python/physics/refdata/lgrngn_cond_substepping_refdata.csv (data file, 280 rows); the 56 rows
    with exact_sstp=False are the per-cell substepping path we build.  Tolerances are the reference's own
    (tests/python/physics/lgrngn_cond_substepping_test.py:80-92); the oracle actually lands ~1e-6 from them
    and reproduces the integer-valued diagnostics (sums of multiplicities) exactly.
  * tests/python/physics/lgrngn_cond.py, puddle.py, tests/common/test_common_pvs.cpp, tests/toms748.
"""
import csv
import ctypes
import os

import numpy as np
import pytest

from _harness import oracle_particles, oracle_fastmath_particles, oracle_lib, T_of, p_of, th_dry2std, lognormal_fn
from libcloudphxx_amd import lgrngn

HERE = os.path.dirname(os.path.abspath(__file__))


def _rows():
    with open(os.path.join(HERE, "golden", "lgrngn_cond_substepping_refdata.csv")) as f:
        rows = [r for r in csv.DictReader(f) if r["exact_sstp"] == "False"]
    assert len(rows) == 56
    return rows


def _rows_exact():
    with open(os.path.join(HERE, "golden", "lgrngn_cond_substepping_refdata.csv")) as f:
        rows = [r for r in csv.DictReader(f) if r["exact_sstp"] == "True"]
    assert len(rows) == 224
    return rows


def _row_id(r):
    return "%s-%s-sstp%s-%s%s-act%s" % ("constp" if r["constp"] == "True" else "varp", r["RH_formula"], r["sstp_cond"],
                                         "mix" if r["mixing"] == "True" else "nomix", "-adaptive" if r["adaptive"] == "True" else "",
                                         r["sstp_cond_act"])


def run_substepping_case(make, RH_formula, sstp_cond, constp, step_count=100, exact=False, mixing=True, adaptive=False,
                         sstp_cond_act=1):
    """tests/python/physics/lgrngn_cond_substepping.py:60-240, verbatim procedure"""
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(.61, 0.): lognormal_fn(.04e-6 / 2, 1.4, 60e6), (1.28, 0.): lognormal_fn(4e-6 / 2, 1.2, 10e6)}
    oi.coal_switch = False
    oi.sedi_switch = False
    oi.RH_max = 0.95
    oi.dt = 1
    oi.sd_conc = 1000
    oi.n_sd_max = 1000
    oi.sstp_cond = sstp_cond
    oi.RH_formula = RH_formula
    oi.exact_sstp_cond, oi.sstp_cond_mix, oi.adaptive_sstp_cond, oi.sstp_cond_act = exact, mixing, adaptive, sstp_cond_act
    oi.rc2_T, oi.sstp_cond_adapt_drw2_eps, oi.sstp_cond_adapt_drw2_max = 10, 1e-3, 2      # :62-64
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.coal = False
    opts.RH_max = 1.005
    rhod, th, rv = np.array([1.1]), np.array([305.]), np.array([0.0085])
    rhod_ss, th_ss, rv_ss = np.array([1.]), np.array([300.]), np.array([0.0091])
    p_ss = np.array([p_of(rhod_ss[0], rv_ss[0], T_of(th_ss[0], rhod_ss[0]))])
    if constp:
        th[0] = th_dry2std(th[0], rv[0])
        th_ss[0] = th_dry2std(th_ss[0], rv_ss[0])
        oi.const_p = True
        oi.th_dry = False
    pr = make(oi)
    if constp:
        pr.init(th, rv, rhod, p_ss)
    else:
        pr.init(th, rv, rhod)

    def ss():
        pr.diag_RH()
        return (np.frombuffer(pr.outbuf())[0] - 1) * 100

    def mom_ratio(k):
        pr.diag_wet_rng(0.5e-6, 1)
        pr.diag_wet_mom(k)
        mk = np.frombuffer(pr.outbuf())[0]
        pr.diag_wet_mom(0)
        return mk / np.frombuffer(pr.outbuf())[0]

    def act():
        pr.diag_wet_rng(0.5e-6, 1)
        pr.diag_wet_mom(0)
        return np.frombuffer(pr.outbuf())[0] / 1e3

    def gccn():
        pr.diag_dry_rng(0.5e-6, 1)
        pr.diag_wet_mom(0)
        return np.frombuffer(pr.outbuf())[0] / 1e3

    rhod[0], th[0], rv[0] = rhod_ss[0], th_ss[0], rv_ss[0]
    rv_init, th_init = rv.copy(), th.copy()
    opts.cond = False
    res = {}
    for step in range(step_count):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
        if step == 9:
            res.update(act=act(), mr=mom_ratio(1) * 1e6, sr=mom_ratio(2), tr=mom_ratio(3))
        if step == 0:
            opts.cond = True
    res.update(ss=ss(), th_post_cond=th[0], rv_post_cond=rv[0])
    rv_diff, th_diff = rv_init - rv[0], th_init - th[0]
    rhod[0], th[0], rv[0] = 1.1, 305, 0.0085
    rv_init, th_init = rv.copy(), th.copy()
    for step in range(step_count):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    res.update(th_diff=th[0] - th_init[0] - th_diff[0], rv_diff=rv[0] - rv_init[0] - rv_diff[0],
               act_post_evap=act(), gccn_post_evap=gccn())
    return res


# the reference's own tolerances, lgrngn_cond_substepping_test.py:80-92
REF_TOL = {'ss': ('r', 1.5e-2), 'th_diff': ('a', 1e-5), 'rv_diff': ('a', 1e-6), 'act': ('r', 1.5e-2), 'mr': ('r', 1.5e-2),
           'sr': ('r', 1.5e-2), 'tr': ('r', 1.5e-2), 'act_post_evap': ('r', 1.5e-2), 'gccn_post_evap': ('r', 1.5e-2),
           'th_post_cond': ('r', 1e-4), 'rv_post_cond': ('r', 1e-3)}
# The refdata was produced by a fast-math (-Ofast, CMakeLists.txt:124) build.  The tolerance-terminated root
# finder (eps 2^-15) returns the midpoint of its last bracket, so a build whose iterations stop one step
# earlier/later is biased by O(1e-5) in rw2 per substep; the bias grows with sstp_cond.  Measured here:
#   strict IEEE oracle vs refdata : th_post_cond <= 3.7e-4 K (1.2e-6 rel), th_diff <= 1.3e-5 K, act exact
#   fast-math oracle  vs refdata : th_post_cond <= 3e-6 K  (1e-8 rel),  act exact
# Hence: the strict oracle is held to the reference's own tolerances (th_diff relaxed 1e-5 -> 2e-5 K), and the
# fast-math build of the SAME source is held ~100x tighter.
STRICT_TOL = dict(REF_TOL, th_diff=('a', 2e-5))
TIGHT_FASTMATH = {'th_post_cond': 3e-8, 'rv_post_cond': 3e-7, 'ss': 1e-4, 'mr': 1e-6, 'sr': 1e-6, 'tr': 1e-6}
EXACT = ('act', 'act_post_evap', 'gccn_post_evap')   # sums of integer multiplicities / 1e3


def check_against_row(res, row, tols=STRICT_TOL, tight=None):
    for col in EXACT:
        assert abs(res[col] - float(row[col])) <= 1e-12 * abs(float(row[col])), (col, res[col], row[col])
    for col, (kind, tol) in tols.items():
        ref = float(row[col])
        if kind == 'r':
            assert abs(res[col] - ref) <= tol * abs(ref), (col, res[col], ref)
        else:
            assert abs(res[col] - ref) <= tol, (col, res[col], ref)
    if tight:
        for col, tol in tight.items():
            ref = float(row[col])
            assert abs(res[col] - ref) <= tol * abs(ref), ("tight", col, res[col], ref)


def _run_row(make, row):
    return run_substepping_case(make, lgrngn.RH_formula_t[row["RH_formula"]], int(row["sstp_cond"]), row["constp"] == "True",
                                exact=row["exact_sstp"] == "True", mixing=row["mixing"] == "True", adaptive=row["adaptive"] == "True",
                                sstp_cond_act=int(row["sstp_cond_act"]))


# The 280 + 112 oracle runs (~0.6 s each) are farmed out once per session to a fork pool over the host cores; the
# parametrised tests below only compare the pre-computed result of their row.
_RESULTS = {}


def _pool_job(job):
    kind, row = job
    try:
        return _run_row(oracle_particles if kind == "strict" else oracle_fastmath_particles, row)
    except Exception as e:                                              # delivered to the test that owns the row
        return e


def _result(kind, row):
    if kind not in _RESULTS:
        import multiprocessing as mp
        rows = _rows() + _rows_exact() if kind == "strict" else _rows() + _rows_exact()[::4]
        with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
            out = pool.map(_pool_job, [(kind, r) for r in rows], chunksize=4)
        _RESULTS[kind] = {_row_id(r) + r["exact_sstp"]: o for r, o in zip(rows, out)}
    res = _RESULTS[kind][_row_id(row) + row["exact_sstp"]]
    if isinstance(res, Exception):
        raise res
    return res


@pytest.mark.parametrize("row", _rows(), ids=lambda r: "%s-%s-sstp%s" % ("constp" if r["constp"] == "True" else "varp", r["RH_formula"], r["sstp_cond"]))
def test_cond_substepping_refdata(row):
    check_against_row(_result("strict", row), row)


@pytest.mark.parametrize("row", _rows(), ids=lambda r: "%s-%s-sstp%s" % ("constp" if r["constp"] == "True" else "varp", r["RH_formula"], r["sstp_cond"]))
def test_cond_substepping_refdata_fastmath_build(row):
    check_against_row(_result("fastmath", row), row, REF_TOL, TIGHT_FASTMATH)


# per-particle substepping (exact_sstp_cond; with and without mixing; adaptive number of substeps; sstp_cond_act)
@pytest.mark.parametrize("row", _rows_exact(), ids=_row_id)
def test_perparticle_substepping_refdata(row):
    check_against_row(_result("strict", row), row)


@pytest.mark.parametrize("row", _rows_exact()[::4], ids=_row_id)
def test_perparticle_substepping_refdata_fastmath_build(row):
    check_against_row(_result("fastmath", row), row, REF_TOL, TIGHT_FASTMATH)


# ---- tests/python/physics/lgrngn_cond.py:52-56,131-132,152-187
def run_lgrngn_cond(make, RH_formula, substep_count, constp, opts_dt, step_count=40):
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(.61, 0.): lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
    oi.coal_switch = oi.sedi_switch = False
    oi.RH_max = 0.999
    oi.dt = 1
    oi.sd_conc = 100
    oi.n_sd_max = 100
    oi.sstp_cond = substep_count
    oi.RH_formula = RH_formula
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.coal = False
    if opts_dt > 0:
        oi.variable_dt_switch = True
        step_count = int(step_count * oi.dt / opts_dt)
    opts.dt = opts_dt
    rhod, th, rv = np.array([1.]), np.array([300.]), np.array([0.02])
    p = np.array([p_of(rhod[0], rv[0], T_of(th[0], rhod[0]))])
    rv_init = rv.copy()
    if constp:
        th[0] = th_dry2std(th[0], rv[0])
        oi.const_p = True
        oi.th_dry = False
    th_init = th.copy()
    pr = make(oi)
    if constp:
        pr.init(th, rv, rhod, p)
    else:
        pr.init(th, rv, rhod)

    def ss():
        pr.diag_RH()
        return (np.frombuffer(pr.outbuf())[0] - 1) * 100
    opts.cond = False
    for step in range(step_count):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
        opts.cond = True
    ss_post_cond = ss()
    exp_th = {True: 306.9, False: 307.78}
    exp_rv = {True: 1.628e-2, False: 1.7e-2}
    assert abs(th[0] - exp_th[constp]) < 1e-4 * exp_th[constp]
    assert abs(rv[0] - exp_rv[constp]) < 1e-3 * exp_rv[constp]
    rv_diff = rv_init.copy() - rv[0].copy()
    rv[0] = 0.002
    rv_init = rv.copy()
    for step in range(step_count):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    return ss_post_cond, th[0] - th_init[0], rv[0] - rv_init[0] - rv_diff[0]


@pytest.mark.parametrize("constp", [False, True])
@pytest.mark.parametrize("RH_formula", list(lgrngn.RH_formula_t))
@pytest.mark.parametrize("opts_dt", [-1, 0.5])
def test_lgrngn_cond_known_answers(constp, RH_formula, opts_dt):
    th_diffs = []
    for sstp in (1, 10, 100):
        ss, th_diff, rv_diff = run_lgrngn_cond(oracle_particles, RH_formula, sstp, constp, opts_dt)
        assert abs(ss) < 4.5e-3
        assert abs(rv_diff) < 1e-9
        th_diffs.append(th_diff)
    lim = (1.1e-1, 7.4e-2, 7.3e-2) if constp else (4.2e-2, 4.2e-3, 4.2e-4)
    for d, l in zip(th_diffs, lim):
        assert abs(d) < l


# ---- tests/python/physics/puddle.py:74-81 (expected totals do not depend on the seed: stratified sampling)
def run_puddle(make, seed=1234):
    oi = lgrngn.opts_init_t()
    oi.dry_distros = {(.61, 0.): lognormal_fn(100e-6, 1.4, 1e6)}
    oi.coal_switch = False
    oi.sedi_switch = True
    oi.terminal_velocity = lgrngn.vt_t.beard76
    oi.dt = 1
    oi.nz, oi.nx, oi.dz, oi.dx = 1, 2, 1, 1
    oi.z1, oi.x1 = oi.nz * oi.dz, oi.nx * oi.dx
    oi.rng_seed = seed
    oi.sd_conc = 10000
    oi.n_sd_max = oi.sd_conc * oi.nx * oi.nz
    opts = lgrngn.opts_t()
    opts.adve = opts.cond = opts.coal = False
    opts.sedi = True
    rhod = 1. * np.ones((oi.nx, oi.nz))
    th = 300. * np.ones((oi.nx, oi.nz))
    rv = 0.01 * np.ones((oi.nx, oi.nz))
    pr = make(oi)
    pr.init(th, rv, rhod)
    for it in range(10):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    puddle = pr.diag_puddle()
    pr.diag_all()
    pr.diag_sd_conc()
    tab = np.frombuffer(pr.outbuf()).reshape(oi.nx, oi.nz).copy()
    return puddle, tab, oi


PUDDLE_EXPECTED_PER_CELL = {'HNO3': 0.0, 'NH3': 0.0, 'CO2': 0.0, 'SO2': 0.0, 'H2O2': 0.0, 'O3': 0.0, 'S_VI': 0.0, 'H': 0.0,
                            'liquid_volume': 7.087802417148837e-05, 'dry_volume': 5.630090090571395e-06,
                            'particle_number': 815411.5, 'liquid_number': 815411.5, 'ice_mass': 0.0, 'ice_number': 0.0}


def check_puddle(puddle, tab, oi):
    assert tab[0][0] == 0.
    for a in puddle:
        assert np.isclose(puddle[a], oi.nx * PUDDLE_EXPECTED_PER_CELL[a], atol=0., rtol=1e-4), (a, puddle[a])


@pytest.mark.parametrize("seed", [44, 1234])
def test_puddle_known_totals(seed):
    check_puddle(*run_puddle(oracle_particles, seed))


@pytest.mark.parametrize("native_modes", [False, True])
def test_openmp_build_of_the_oracle_is_bit_identical(native_modes):
    """liblcx_oracle_omp.so (bench.py's cpu_baseline leg, the production-size parity tests) runs the elementwise loops, the stable
    sort, the per-cell counts and sums, the compaction and the boundary selections on all cores the way the reference's OpenMP
    backend does (thrust::omp); a run of a cell is summed by one thread in the serial order and the generator stays serial, so its
    results equal the serial oracle's bit for bit -- including the puddle's global sums and the storage order after SDs were removed"""
    from _harness import oracle_omp_particles, box_opts, box_fields
    oi = box_opts(9, 4, 7, 32)
    if native_modes:                                     # (a distribution without a Python callback: the parallel init_n path)
        oi.dry_distros = {(.61, 0.): lgrngn.lognormal([.02e-6, .075e-6], [1.4, 1.6], [60e6, 40e6])}
    th0, rv0, rhod, C = box_fields(oi)
    out = []
    for make in (oracle_particles, oracle_omp_particles):
        pr = make(oi)
        th, rv = th0.copy(), rv0.copy()
        pr.init(th, rv, rhod, **C)
        rw2 = pr.get_attr("rw2")
        z = pr.get_attr("z")
        big = (np.arange(rw2.size) % 37 == 0) & (z < 1.5 * oi.dz)          # drizzle near the floor: falls out, feeds the puddle
        rw2[big] = (150e-6) ** 2
        pr.set_particles(pr.state_u64("n"), pr.get_attr("rd3"), rw2, pr.get_attr("kappa"), np.full(rw2.size, -1.),
                         pr.get_attr("x"), pr.get_attr("y"), z)
        opts = lgrngn.opts_t()
        for _ in range(3):
            pr.step_sync(opts, th, rv, rhod, **C)
            pr.step_async(opts)
        pud = pr.diag_puddle()
        assert max(pud.values()) > 0 and pr.n_part < rw2.size, "no super-droplet left through the floor"
        out.append((th.copy(), rv.copy(), pr.get_attr("rw2"), pr.get_attr("x"), pr.get_attr("z"), pr.state_real("vt"), pr.get_attr("kappa"),
                    pr.state_u64("n"), pr.state_u64("sorted_id"), pr.state_u64("ijk"),
                    np.array([pud[k] for k in sorted(pud)])))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ the reference's icicle regression output at t = 0
# models/kinematic_2D/tests/paper_GMD_2015/fig_a/refdata/travis_out_lgrngn/travis_timestep0000000000.h5 (+ travis_const.h5): the state
# right after particles_t::init of the 2-D lgrngn set-up as the reference's serial backend produced it (real_t = float) -- reduced to
# domain means per bin by tools/extract_icicle_t0.py (tests/golden/icicle_t0_profiles.npz).  The run drew its aerosol from the float
# flavour of the generator's stream, which no double build reproduces number for number; what IS reproducible from it:
#   * the dry-air density profile and the dry potential temperature: the formula library's hydrostatic::p, theta_std::rhod, theta_dry::std2dry;
#   * the initial spectra as STATISTICS: per dry-radius bin the specific concentration, per wet-radius bin likewise (the equilibrium wet
#     radii of init_wet at RH(z)), the third wet moment per dry bin -- two samples of 2.3e5 super-droplets of the same distribution.
ICICLE_T0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icicle_t0_profiles.npz")


def icicle_t0_setup():
    """kin_cloud_2d_lgrngn.hpp:150-200 / icmw8_case1.hpp:160-230 with travis_calc_lgrngn.cpp's options: 60 x 60 grid points over
    1500 m x 1500 m, cells centred on them (the outermost ones half inside), 64 super-droplets per cell, icicle's bimodal aerosol"""
    d = np.load(ICICLE_T0)
    nx = nz = 60
    dx = 1500. / (nx - 1)
    oi = lgrngn.opts_init_t()
    oi.nx, oi.ny, oi.nz = nx, 0, nz
    oi.dx = oi.dz = dx
    oi.dy = 1.
    oi.x0, oi.z0 = dx / 2, dx / 2
    oi.x1, oi.z1 = (nx - .5) * dx, (nz - .5) * dx
    oi.y1 = 1.
    oi.dt = 1.
    oi.sd_conc = 64
    oi.n_sd_max = 64 * nx * nz
    oi.dry_distros = {(.61, 0.): lgrngn.lognormal([.02e-6, .075e-6], [1.4, 1.6], [60e6, 40e6])}
    oi.kernel = lgrngn.kernel_t.geometric
    oi.terminal_velocity = lgrngn.vt_t.khvorostyanov_spherical
    oi.sstp_cond = oi.sstp_coal = 10
    rhod = np.ascontiguousarray(np.broadcast_to(d["rhod_profile"], (nx, nz)))
    th = np.full((nx, nz), d["th"][0])
    rv = np.full((nx, nz), d["rv"][0])
    return oi, th, rv, rhod, d


def icicle_t0_diagnose(pr, d):
    """what kin_cloud_2d_lgrngn.hpp:40-95 records: domain means of the per-cell diagnostics"""
    out = {"rd": [], "rw": [], "rw3": []}
    for i in range(39):
        pr.diag_dry_rng(d["dry_edges"][i], d["dry_edges"][i + 1]); pr.diag_dry_mom(0)
        out["rd"].append(pr.outbuf_array().mean())
        pr.diag_dry_rng(d["dry_edges"][i], d["dry_edges"][i + 1]); pr.diag_wet_mom(3)
        out["rw3"].append(pr.outbuf_array().mean())
    for i in range(24):
        pr.diag_wet_rng(d["wet_edges"][i], d["wet_edges"][i + 1]); pr.diag_wet_mom(0)
        out["rw"].append(pr.outbuf_array().mean())
    pr.diag_wet_rng(.5e-6, 25e-6); pr.diag_wet_mom(0)
    out["fssp_max"] = pr.outbuf_array().max()
    pr.diag_all(); pr.diag_sd_conc()
    sd = pr.outbuf_array()
    out["sd_conc"] = (sd.min(), sd.max())
    return {k: np.array(v) for k, v in out.items()}


def check_icicle_t0(got, d):
    # 2.3e5 super-droplets uniform in ln(rd) over ~7 e-folds: ~7500 per dry bin of 0.23, ~15000 per wet bin of 0.46; two independent
    # samples differ by sqrt(2 / N) ~ 1.6 % resp. 1.2 %: held to five times that where the spectrum is populated (above 1 % of its peak)
    for key, ref, tol in (("rd", d["rd_mom0_mean"], 0.08), ("rw", d["rw_mom0_mean"], 0.08), ("rw3", d["rw3ofrd_mom3_mean"], 0.10)):
        core = ref > 1e-2 * ref.max()
        assert core.sum() >= 8, key
        rel = got[key][core] / ref[core] - 1
        assert np.abs(rel).max() < tol, (key, rel)
        assert abs(rel.mean()) < 0.02, (key, rel.mean())                         # (no common factor: units, volumes, densities)
        assert np.all((got[key] > 0) == (ref > 0)) or np.abs(np.flatnonzero(got[key] > 0)[[0, -1]] - np.flatnonzero(ref > 0)[[0, -1]]).max() <= 1, key
    assert got["fssp_max"] == 0 or got["fssp_max"] < 2 * d["fssp_mom0_max"][0]
    assert got["sd_conc"][1] == d["sd_conc"][1] == 64 and got["sd_conc"][0] >= 62              # (the reference's t = 0 minimum: 63)


def test_icicle_t0_density_and_theta_from_the_formula_library():
    from libcloudphxx_amd import common
    d = np.load(ICICLE_T0)
    dz = 1500. / 59
    z = np.arange(60) * dz
    rhod = np.array([common.rhod(common.p_hydro(zz, 289., 7.5e-3, 0., 101500.), 289., 7.5e-3) for zz in z])
    np.testing.assert_allclose(rhod, d["rhod_profile"], rtol=1e-6)                # (the reference computed it in float: two pow chains, measured 3.4e-7)
    np.testing.assert_allclose(common.th_std2dry(289., 7.5e-3), d["th"][0], rtol=1e-7)
    assert d["th"][0] == d["th"][1] and d["rv"][0] == d["rv"][1] == np.float32(7.5e-3)


def test_icicle_t0_initial_spectra_oracle():
    oi, th, rv, rhod, d = icicle_t0_setup()
    pr = oracle_particles(oi)
    pr.init(th, rv, rhod)
    check_icicle_t0(icicle_t0_diagnose(pr, d), d)


# ---- the oracle's float flavour (round 5; oracle/Makefile: the same source with real = float, <tgmath.h>, float literals)
def test_icicle_t0_initial_spectra_oracle_float_flavour():
    """the reference's t = 0 output came from ITS float build (fig_a/calc.cpp:36-39): the float flavour of the oracle is held to the same
    statistics as the double one (five sampling standard deviations of 2.3e5 super-droplets, no common factor)"""
    import _harness as h
    oi, th, rv, rhod, d = icicle_t0_setup()
    pr = h.oracle_f32_particles(oi)
    pr.init(th.astype(np.float32), rv.astype(np.float32), rhod.astype(np.float32))
    check_icicle_t0(icicle_t0_diagnose(pr, d), d)


def test_float_flavour_of_the_oracle_tracks_the_double_one():
    """the two flavours on one small box, three full steps each on its own (the double engine's draws serve both: one random stream):
    the same cells throughout, th / rv / wet radii as far apart as 24-bit arithmetic and a root finder at float's tolerance 2^-7 put them"""
    import _harness as h
    from libcloudphxx_amd import lgrngn
    oi = h.box_opts(4, 3, 5, 32)
    th, rv, rhod, C = h.box_fields(oi)
    res = {}
    for nm, mk, dt in (("f64", h.oracle_particles, np.float64), ("f32", h.oracle_f32_particles, np.float32)):
        pr = mk(oi)
        f = [a.astype(dt) for a in (th, rv, rhod)]
        Cs = {k: v.astype(dt) for k, v in C.items()}
        pr.init(f[0], f[1], f[2], **Cs)
        assert pr.get_attr("rw2").dtype == dt
        opts = lgrngn.opts_t()
        opts.coal = False                    # (multiplicities differ by one here and there: rounded from float products)
        for _ in range(3):
            pr.step_sync(opts, f[0], f[1], f[2], **Cs)
            pr.step_async(opts)
        res[nm] = (f[0].astype(np.float64), f[1].astype(np.float64), pr.state_real("rw2"), pr.state_u64("ijk"), pr.state_real("x"))
    a, b = res["f64"], res["f32"]
    assert a[2].size == b[2].size
    assert (a[3] != b[3]).mean() < 2e-3                   # (a droplet within 1e-7 of a cell face)
    np.testing.assert_allclose(b[0], a[0], rtol=2e-5)
    np.testing.assert_allclose(b[1], a[1], rtol=3e-4)
    err = np.abs(b[2] / a[2] - 1)
    assert np.median(err) < 2e-3 and err.max() < 0.1, (np.median(err), err.max())
    np.testing.assert_allclose(b[4], a[4], rtol=0, atol=1e-3)
    with pytest.raises(RuntimeError, match="real_t of 4 bytes"):
        lgrngn.particles_t(oi, np.float64, lib=h.oracle_f32_lib(), prefix="orc_")
