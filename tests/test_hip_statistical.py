"""Statistical checks of the device random-number path (Philox uniforms and the salted-bijection shuffle keys instead of the CPU
stream): the reference's analytic Golovin test, tests/python/physics/coalescence_golovin.py:31-153 (sd_conc variant, RMSD < 1.2e-5;
const_multi variant, RMSD < 3e-5).

How much of the RMSD is the draw (tools/golovin_ensemble.py, 12 seeds each, MI355X): const_multi 1.7e-5 +- 0.7e-5 (max 3.1e-5) with the
shuffle keys of this round, 1.7e-5 +- 0.5e-5 (max 3.0e-5) with round 2's Philox keys (opts_init.dbg_flags & SHUFFLE_PHILOX); sd_conc 1.0e-5 +- 0.2e-5
against 0.9e-5 +- 0.2e-5 -- the two generators are indistinguishable here, and the reference's thresholds sit about two standard
deviations above the mean of either (a seed in forty fails them; the default seed does not)."""
import numpy as np
import pytest
from scipy import special

import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu


def golovin(v, t, n0, v0, b):        # Scott et al 1967, eq. 2.7 (coalescence_golovin.py:47-58)
    x = v / v0
    T = b * n0 * v0 * t
    tau = 1 - np.exp(-T)
    bessel = special.iv(1, 2 * x * np.sqrt(tau))
    result = 0.
    if not np.isinf(bessel):
        result = n0 / v0 * bessel * (1 - tau) * np.exp(-x * (tau + 1)) / x / np.sqrt(tau)
    return 0. if np.isnan(result) else result


def mass_dens(pr, rad, sig0=0.62):
    """host restatement of mass_dens_estim (diagnose_SD_attributes/particles_impl_mass_dens.ipp:7-120) for one 0-D cell"""
    n = pr.state_u64("n").astype(np.float64)
    rw2 = pr.get_attr("rw2")
    sig = sig0 / len(n) ** 0.2
    est = np.sum(n / sig * rw2 ** 1.5 * np.exp(-((np.log(np.sqrt(rw2)) - np.log(rad)) / sig) ** 2 / 2.))
    return est * 4. / 3. * 1e3 * np.sqrt(np.pi / 2.)       # dv = 1/rhod = 1


@pytest.mark.parametrize("init", ["sd_conc", "const_multi"])
@pytest.mark.parametrize("opts_dt", [-1, 400.])
def test_golovin_analytic(opts_dt, init):
    simulation_time = 800
    r_zero, n_zero, b = 30.084e-6, 2 ** 23, 1500.
    v_zero = 4. / 3. * r_zero ** 3 * np.pi

    def expvolumelnr(lnr):
        r = np.exp(lnr)
        return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(- np.power((r / r_zero), 3))
    oi = lgrngn.opts_init_t()
    oi.dt = simulation_time
    oi.sstp_coal = simulation_time
    oi.sedi_switch = False
    oi.dry_distros = {(1e-10, 0.): expvolumelnr}
    oi.kernel = lgrngn.kernel_t.golovin
    oi.terminal_velocity = lgrngn.vt_t.beard77
    oi.kernel_parameters = np.array([b])
    if init == "sd_conc":                              # coalescence_golovin.py:112-120
        oi.sd_conc = 2 ** 14
        oi.n_sd_max = 2 ** 14
    else:
        oi.sd_conc = 0
        oi.sd_const_multi = 1000
        oi.n_sd_max = int(float(n_zero) / oi.sd_const_multi + 10)
    opts = lgrngn.opts_t()
    opts.adve = opts.sedi = opts.cond = False
    opts.dt = opts_dt
    n_step = 1
    if opts_dt > 0:
        oi.variable_dt_switch = True
        n_step = int(simulation_time / opts_dt)
    rhod, th, rv = np.ones(1), 300. * np.ones(1), 0.01 * np.ones(1)
    pr = h.hip_particles(oi)
    pr.init(th, rv, rhod)
    pr.diag_all()
    pr.diag_wet_mom(0)
    n_init = np.frombuffer(pr.outbuf())[0]
    for _ in range(n_step):
        pr.step_sync(opts, th, rv, rhod)
        pr.step_async(opts)
    bins = pow(10, -6 + np.arange(150) / 50.)
    res, ana = np.zeros(bins.size - 1), np.zeros(bins.size - 1)
    for i in range(res.size):
        rad = (bins[i] + bins[i + 1]) / 2.
        res[i] = mass_dens(pr, rad)
        pr.diag_all()
        pr.diag_wet_mass_dens(rad, .62)                    # the device estimator against the host restatement above
        assert abs(np.frombuffer(pr.outbuf())[0] - res[i]) <= 1e-9 * abs(res[i]) + 1e-300
        vol = 4. / 3. * rad ** 3 * np.pi
        ana[i] = golovin(vol, simulation_time, n_init, v_zero, b) * vol * vol * 3000.
    sel = (res > 0) | (ana > 0)
    rmsd = np.sqrt(np.sum((res[sel] - ana[sel]) ** 2) / np.sum(sel))
    assert rmsd < (1.2e-5 if init == "sd_conc" else 3e-5), rmsd          # coalescence_golovin.py:147-153
