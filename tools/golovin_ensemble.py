"""RMSD of the Golovin test (tests/test_hip_statistical.py, const_multi and sd_conc variants) over a range of seeds: how much of the
distance to the analytic solution is the draw.  python tools/golovin_ensemble.py [n_seeds] [philox]
(philox: round 2's shuffle keys, opts_init.dbg_flags & SHUFFLE_PHILOX)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_statistical as ts
import _harness as h
from libcloudphxx_amd import lgrngn

def one(init, opts_dt, seed):
    simulation_time = 800
    r_zero, n_zero, b = 30.084e-6, 2 ** 23, 1500.
    v_zero = 4. / 3. * r_zero ** 3 * np.pi
    def expvolumelnr(lnr):
        r = np.exp(lnr)
        return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(- np.power((r / r_zero), 3))
    oi = lgrngn.opts_init_t()
    oi.dt = simulation_time; oi.sstp_coal = simulation_time; oi.sedi_switch = False
    oi.dry_distros = {(1e-10, 0.): expvolumelnr}
    oi.kernel = lgrngn.kernel_t.golovin; oi.terminal_velocity = lgrngn.vt_t.beard77
    oi.kernel_parameters = np.array([b]); oi.rng_seed = seed
    oi.dbg_flags = int(lgrngn.dbg.SHUFFLE_PHILOX) if PHILOX else 0
    if init == "sd_conc":
        oi.sd_conc = 2 ** 14; oi.n_sd_max = 2 ** 14
    else:
        oi.sd_conc = 0; oi.sd_const_multi = 1000; oi.n_sd_max = int(float(n_zero) / oi.sd_const_multi + 10)
    opts = lgrngn.opts_t(); opts.adve = opts.sedi = opts.cond = False; opts.dt = opts_dt
    n_step = 1
    if opts_dt > 0:
        oi.variable_dt_switch = True; n_step = int(simulation_time / opts_dt)
    rhod, th, rv = np.ones(1), 300. * np.ones(1), 0.01 * np.ones(1)
    pr = h.hip_particles(oi); pr.init(th, rv, rhod)
    pr.diag_all(); pr.diag_wet_mom(0); n_init = np.frombuffer(pr.outbuf())[0]
    for _ in range(n_step):
        pr.step_sync(opts, th, rv, rhod); pr.step_async(opts)
    bins = pow(10, -6 + np.arange(150) / 50.)
    res, ana = np.zeros(bins.size - 1), np.zeros(bins.size - 1)
    for i in range(res.size):
        rad = (bins[i] + bins[i + 1]) / 2.
        res[i] = ts.mass_dens(pr, rad)
        vol = 4. / 3. * rad ** 3 * np.pi
        ana[i] = ts.golovin(vol, simulation_time, n_init, v_zero, b) * vol * vol * 3000.
    sel = (res > 0) | (ana > 0)
    return float(np.sqrt(np.sum((res[sel] - ana[sel]) ** 2) / np.sum(sel)))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
PHILOX = len(sys.argv) > 2 and sys.argv[2] == "philox"
for init in ("const_multi", "sd_conc"):
    for dt in (-1, 400.):
        v = np.array([one(init, dt, 44 + s) for s in range(n)])
        print("philox" if PHILOX else "hashed", init, dt, "mean %.3e sd %.3e min %.3e max %.3e" % (v.mean(), v.std(), v.min(), v.max()), np.round(v * 1e5, 2), flush=True)
