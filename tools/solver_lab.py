"""Offline laboratory for the fast arithmetic's root finder (CPU only, numpy): the droplets and cell fields that the condensation of a
settled bench box sees (the OpenMP oracle run for a few steps on bench.py's fields), the collected growth rate of
csrc/lcx_math.hpp (cond_fun_fast) restated in numpy with its analytic derivative, and candidate solvers counted in growth-rate
EVALUATIONS per droplet and per wave of 64 storage neighbours (a wave is as slow as its slowest droplet).

    python3 tools/solver_lab.py [n_cells_per_dim=32] [steps=8] [workload=stratocumulus|coal-stress]

Nothing here is product or test code; it is how round 5's solver (advance_rw2_lean3_with) was chosen.  Prints, for every variant, the
distribution of evaluations, the mean per droplet, the mean over waves of the per-wave maximum and the largest relative distance of
rw2 from the oracle's answer (TOMS748's bracket midpoint).
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as h            # noqa: E402
import bench                    # noqa: E402
from libcloudphxx_amd import lgrngn   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
workload = sys.argv[3] if len(sys.argv) > 3 else "stratocumulus"
cache = "/tmp/solver_lab_%s_%d_%d.npz" % (workload, n, steps)

# ---- constants (oracle/orc_physics.h)
c_pd, c_pv, c_pw = 1005., 1850., 4218.
R = 8.3144621
R_v, R_d = R / 0.018, R / 0.02897
rho_w, D_0, K_0 = 1e3, 2.26e-5, 2.4e-2
T_tri, l_tri = 273.16, 2.5e6


def snapshot():
    if os.path.exists(cache):
        return dict(np.load(cache))
    oi = bench.make_opts_init(n, n, n, 64, 40., 1, 1, 44, workload)
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(n, n, n, 0, n, np, np.float64)
    sh = (n, n, n)
    th, rv, rhod = [np.ascontiguousarray(np.broadcast_to(a, sh)) for a in (th, rv, rhod)]
    C = dict(Cx=np.ascontiguousarray(np.broadcast_to(Cx, (n + 1, n, n))), Cy=np.ascontiguousarray(np.broadcast_to(Cy, (n, n + 1, n))),
             Cz=np.ascontiguousarray(np.broadcast_to(Cz, (n, n, n + 1))))
    orc = h.oracle_omp_particles(oi)
    orc.init(th, rv, rhod, **C)
    opts = lgrngn.opts_t()
    for s in range(steps):
        orc.step_sync(opts, th, rv, rhod, **C)
        orc.step_async(opts)
    d = dict(rw2=orc.state_real("rw2"), rd3=orc.state_real("rd3"), kpa=orc.state_real("kappa"), vt=orc.state_real("vt"),
             ijk=orc.state_u64("ijk").astype(np.int64))
    orc.step_sync(opts, th, rv, rhod, **C)
    for k in ("T", "RH", "rhod", "eta", "lambda_D", "lambda_K"):
        d[k] = orc.state_real(k)
    # rv as condensation saw it is not kept; RH, T and rhod give rho_v = RH p_vs(T) / (R_v T) for the formula in use (pv_cc)
    d["rv_after"] = orc.state_real("rv")
    d["rw2_ref"] = orc.state_real("rw2")
    d["dt"] = np.float64(oi.dt)
    np.savez(cache, **d)
    return d


S = snapshot()
ijk = S["ijk"]
rw2_old, rd3, kpa, vt = S["rw2"], S["rd3"], S["kpa"], S["vt"]
Tk = S["T"][ijk]; RH = S["RH"][ijk]; rhod = S["rhod"][ijk]; eta = S["eta"][ijk]
lam_D = S["lambda_D"][ijk]; lam_K = S["lambda_K"][ijk]
dt = float(S["dt"])
RH_max = 44.
# rho_v from RH (RH_formula pv_cc: RH = p_v / p_vs, p_v = rho_v R_v T)
p_vs = 611.73 * np.exp((l_tri + (c_pw - c_pv) * T_tri) / R_v * (1. / T_tri - 1. / Tk) - (c_pw - c_pv) / R_v * np.log(Tk / T_tri))
rho_v = RH * p_vs / (R_v * Tk)
lv = l_tri + (c_pv - c_pw) * (Tk - T_tri)
A = 2. * (0.07275 * (1. - 0.002 * (Tk - 291.))) / R_v / Tk / rho_w
Sc = eta / rhod / D_0
Pr = c_pd * eta / K_0
RH_eff = np.minimum(RH, RH_max)
c1 = 2. / (D_0 * rho_v)
c2_rho = 2. * lv * (lv / R_v / Tk - 1.) / (K_0 * RH_eff * Tk)
RH_rho_w = RH_eff * rho_w
c_Re = vt * 2. * rhod / eta
rd3_1mk = rd3 * (1. - kpa)
N = rw2_old.size
print("droplets", N, "cells", S["T"].size, "dt", dt)


class Fun:
    """the collected growth rate on an index subset; counts evaluations"""

    def __init__(self):
        self.evals = np.zeros(N, dtype=np.int64)
        self.devals = np.zeros(N, dtype=np.int64)

    def F(self, idx, x, deriv=False):
        (self.devals if deriv else self.evals)[idx] += 1
        irw = 1. / np.sqrt(x)
        rw = x * irw
        Re = c_Re[idx] * rw
        KnD, KnK = lam_D[idx] * irw, lam_K[idx] * irw
        nD, dD = 1. + KnD, 1. + KnD * (1.71 + 1.33 * KnD)
        nK, dK = 1. + KnK, 1. + KnK * (1.71 + 1.33 * KnK)
        rw3 = x * rw
        na, da = rw3 - rd3[idx], rw3 - rd3_1mk[idx]
        xS, xN = Re * Sc[idx], Re * Pr[idx]
        m = np.where(Re > 1., np.maximum(1., np.abs(Re) ** .077), 1.)
        cS, cN = np.cbrt(1. + xS), np.cbrt(1. + xN)
        Sh, Nu = 1. + cS * m, 1. + cN * m
        klv = np.exp(A[idx] * irw)
        nDSh, nKNu = nD * Sh, nK * Nu
        g = da * RH_eff[idx] - na * klv
        num = g * (nDSh * nKNu)
        q = c1[idx] * dD * nKNu + c2_rho[idx] * dK * nDSh
        den = (da * RH_rho_w[idx]) * q
        Fv = 2. * num / den
        if not deriv:
            return Fv
        # d/dx with x = rw^2:  d irw = -irw^3 / 2, d rw = irw / 2, d rw3 = 1.5 rw
        dirw = -.5 * irw * irw * irw
        drw = .5 * irw
        dKnD, dKnK = lam_D[idx] * dirw, lam_K[idx] * dirw
        dnD, ddD = dKnD, dKnD * (1.71 + 2.66 * KnD)
        dnK, ddK = dKnK, dKnK * (1.71 + 2.66 * KnK)
        dna = dda = 1.5 * rw
        dklv = klv * A[idx] * dirw
        dRe = c_Re[idx] * drw
        # ventilation: d cbrt(1 + Re Sc) = Sc dRe / (3 cS^2); the factor m's derivative for Re > 1: m 0.077 / Re dRe
        dcS = Sc[idx] * dRe / (3. * cS * cS)
        dcN = Pr[idx] * dRe / (3. * cN * cN)
        dm = np.where((Re > 1.) & (np.abs(Re) ** .077 > 1.), m * .077 / np.where(Re > 1., Re, 1.) * dRe, 0.)
        dSh, dNu = dcS * m + cS * dm, dcN * m + cN * dm
        dnDSh, dnKNu = dnD * Sh + nD * dSh, dnK * Nu + nK * dNu
        dg = dda * RH_eff[idx] - dna * klv - na * dklv
        dnum = dg * (nDSh * nKNu) + g * (dnDSh * nKNu + nDSh * dnKNu)
        dq = c1[idx] * (ddD * nKNu + dD * dnKNu) + c2_rho[idx] * (ddK * nDSh + dK * dnDSh)
        dden = (dda * RH_rho_w[idx]) * q + (da * RH_rho_w[idx]) * dq
        dF = 2. * (dnum * den - num * dden) / (den * den)
        return Fv, dF

    def f(self, idx, x):
        return rw2_old[idx] + dt * self.F(idx, x) - x


eps = 2. ** -15
cond_mlt = 2.


def tol_reached(a, b):
    return np.abs(a - b) <= eps * np.minimum(np.abs(a), np.abs(b))


def check_derivative():
    fn = Fun()
    idx = np.arange(0, N, max(1, N // 200000))
    x = rw2_old[idx]
    Fv, dF = fn.F(idx, x, True)
    hh = x * 1e-6
    num = (fn.F(idx, x + hh) - fn.F(idx, x - hh)) / (2 * hh)
    rel = np.abs(num - dF) / np.maximum(np.abs(dF), 1e-300)
    print("derivative check: median rel diff %.2e, 99%% %.2e, max %.2e" % (np.median(rel), np.percentile(rel, 99), rel.max()))


def solve_lean2(guard=False):
    """the product's round-4 solver (lcx_math.hpp advance_rw2_lean2_with); guard: round 5's stopping rule (the iterate's convergence
    believed only while |f| falls to at most half of the previous value on its side)"""
    fn = Fun()
    r = rw2_old.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    drw2 = dt * fn.F(all_i, rw2_old[all_i])
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b)
    mid = ~early & (a == a_un) & tol_reached(a, b)
    r[all_i[mid]] = (a[mid] + b[mid]) / 2
    go = ~early & ~mid
    idx = all_i[go]; a, b, drw2, rd2 = a[go], b[go], drw2[go], rd2[go]
    grows = drw2 > 0
    f_far = fn.f(idx, np.where(grows, b, a))
    fa = np.where(grows, drw2, f_far); fb = np.where(grows, f_far, drw2)
    same = fa * fb > 0
    res = np.where(same, rw2_old[idx] + drw2, np.where(fa == 0, a, b))
    loop = ~same & (fa != 0) & (fb != 0)
    x0, f0, x1, f1 = a.copy(), fa.copy(), b.copy(), fb.copy()
    c = x1 - f1 * (x1 - x0) / (f1 - f0)
    res[loop] = c[loop]
    act = loop.copy()
    for it in range(100):
        if not act.any():
            break
        k = np.nonzero(act)[0]
        fc = fn.f(idx[k], c[k])
        opp = (fc < 0) != (f1[k] < 0)
        fs = np.where(opp, f0[k], f1[k])
        m = 1. - fc / f1[k]
        m = np.where(m > 0, m, .5)
        f0[k] = np.where(opp, f1[k], f0[k] * m)
        x0[k] = np.where(opp, x1[k], x0[k])
        x1[k] = c[k]; f1[k] = fc
        c_new = x1[k] - f1[k] * (x1[k] - x0[k]) / (f1[k] - f0[k])
        res[k] = c_new
        conv = np.abs(c_new - c[k]) <= eps * np.minimum(np.abs(c_new), np.abs(c[k]))
        if guard:
            conv &= np.abs(fc) <= .5 * np.abs(fs)
        done = conv | tol_reached(x0[k], x1[k])
        c[k] = c_new
        act[k[done]] = False
    bad = loop & ~((res > np.minimum(a, b)) & (res < np.maximum(a, b)))
    res[bad] = x1[bad]
    res = np.maximum(res, rd2)
    r[idx] = res
    return r, fn


def solve_lean3(overshoot=.5, accept_euler=True, newton_only_first=True):
    """round 5: Newton probe from the near end instead of the far end's evaluation.
    sequence of iterates: c0 = rw2_old + drw2 (explicit Euler), c1 = rw2_old + drw2 / (1 - dt F') (Newton, linearised implicit);
    |c1 - c0| <= eps min(c0, c1): converged, return c1 (the stopping rule of the loop, applied to the first two iterates);
    else probe f at c1 pushed half a tolerance beyond (so that a good c1 brackets the root together with rw2_old)"""
    fn = Fun()
    r = rw2_old.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    Fv, dF = fn.F(all_i, rw2_old[all_i], True)
    drw2 = dt * Fv
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b)
    mid = ~early & (a == a_un) & tol_reached(a, b)
    r[all_i[mid]] = (a[mid] + b[mid]) / 2
    go = ~early & ~mid
    idx = all_i[go]; a, b, drw2, rd2, dF = a[go], b[go], drw2[go], rd2[go], dF[go]
    x_old = rw2_old[idx]
    s = 1. - dt * dF                       # -f'(rw2_old)
    step = drw2 / s
    c0 = x_old + drw2
    c1 = x_old + step
    inside = (s > 0) & (c1 > a) & (c1 < b)
    res = np.full(idx.size, np.nan)
    state = np.zeros(idx.size, dtype=np.int8)      # 0 probe pending, 1 far pending, 2 bracketed loop, 9 done
    conv0 = inside & (np.abs(c1 - c0) <= eps * np.minimum(np.abs(c0), np.abs(c1))) if accept_euler else np.zeros(idx.size, bool)
    res[conv0] = c1[conv0]; state[conv0] = 9
    grows = drw2 > 0
    far = np.where(grows, b, a)
    # latest point (x1, f1) = the near end; retained end unknown until a sign change is seen
    x1, f1 = x_old.copy(), drw2.copy()
    x0, f0 = far.copy(), np.full(idx.size, np.nan)
    push = overshoot * eps * np.abs(c1) * np.sign(step)
    c = np.where(inside, np.clip(c1 + push, np.minimum(a, b), np.maximum(a, b)), far)
    state[~inside & (state == 0)] = 1
    trips = np.zeros(idx.size, dtype=np.int64)
    for it in range(100):
        act = state < 9
        if not act.any():
            break
        k = np.nonzero(act)[0]
        fc = fn.f(idx[k], c[k])
        trips[k] += 1
        st = state[k]
        opp = (fc < 0) != (f1[k] < 0)
        # --- probe, no sign change against the near end: the far end is evaluated next; the probe point replaces the near end
        pk = k[(st == 0) & ~opp]
        fcp = fc[(st == 0) & ~opp]
        x1[pk] = c[pk]; f1[pk] = fcp; c[pk] = far[pk]; state[pk] = 1
        # --- far end evaluated
        sel = (st == 1)
        fk = k[sel]; fcf = fc[sel]; oppf = opp[sel]
        eu = fk[~oppf]                                    # no sign change on the reference's bracket: explicit Euler
        res[eu] = x_old[eu] + drw2[eu]; state[eu] = 9
        br = fk[oppf]; fcb = fcf[oppf]
        x0[br] = far[br]; f0[br] = fcb                    # bracket (far end, latest same-sign point)
        cn = x1[br] - f1[br] * (x1[br] - x0[br]) / (f1[br] - f0[br])
        c[br] = cn; res[br] = cn; state[br] = 2
        # --- probe with a sign change, or a loop trip: the standard update
        sel = ((st == 0) & opp) | (st == 2)
        lk = k[sel]; fcl = fc[sel]; oppl = opp[sel]; stl = st[sel]
        m = 1. - fcl / f1[lk]
        m = np.where(m > 0, m, .5)
        f0n = np.where(oppl, f1[lk], f0[lk] * m)
        x0n = np.where(oppl, x1[lk], x0[lk])
        f0[lk] = f0n; x0[lk] = x0n
        x1[lk] = c[lk]; f1[lk] = fcl
        c_new = x1[lk] - f1[lk] * (x1[lk] - x0[lk]) / (f1[lk] - f0[lk])
        res[lk] = c_new
        done = (np.abs(c_new - c[lk]) <= eps * np.minimum(np.abs(c_new), np.abs(c[lk]))) | tol_reached(x0[lk], x1[lk])
        c[lk] = c_new
        state[lk] = np.where(done, 9, 2)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    bad = ~((res > lo) & (res < hi)) & ~np.isnan(res)
    res[bad] = x1[bad]
    res = np.maximum(res, rd2)
    r[idx] = res
    return r, fn


def report(name, r, fn, dcost=.45):
    ev = fn.evals + fn.devals
    cost = fn.evals + (1. + dcost) * fn.devals
    W = N // 64 * 64
    wmax = ev[:W].reshape(-1, 64).max(axis=1)
    wcost = cost[:W].reshape(-1, 64).max(axis=1)
    ref = S["rw2_ref"]
    ok = rw2_old > 0
    rel = np.abs(r[ok] / ref[ok] - 1.)
    hist = np.bincount(ev, minlength=10)
    print("%-34s evals/droplet %.3f  wave-max %.3f  (cost units: mean %.3f, wave-max %.3f)  vs oracle: max %.2e, 99.99%% %.2e  hist %s"
          % (name, ev.mean(), wmax.mean(), cost.mean(), wcost.mean(), rel.max(), np.percentile(rel, 99.99), (hist / N).round(4)[:10]))
    return ev


if __name__ == "__main__":
    check_derivative()
    r2, f2 = solve_lean2()
    e2 = report("lean2 (round 4)", r2, f2)
    for ov in (0., .5, 1.):
        for ae in (True, False):
            r3, f3 = solve_lean3(ov, ae)
            e3 = report("lean3 overshoot %.1f accept_euler %d" % (ov, ae), r3, f3)
            rel = np.abs(r3 / r2 - 1.)
            print("      vs lean2: max %.2e  99.99%% %.2e" % (rel.max(), np.percentile(rel, 99.99)))


def describe(ev, lo, hi=99):
    fn = Fun()
    k = np.nonzero((ev >= lo) & (ev <= hi))[0]
    if not k.size:
        return
    Fv, dF = fn.F(k, rw2_old[k], True)
    rw = np.sqrt(rw2_old[k]); rd = np.cbrt(rd3[k])
    q = lambda v: "%.3g/%.3g/%.3g" % tuple(np.percentile(v, [5, 50, 95]))
    print("  evals %d..%d: %d droplets  rw[um] %s  rw/rd %s  RH %s  step/rw2 %s  dtF' %s" %
          (lo, hi, k.size, q(rw * 1e6), q(rw / rd), q(RH[k]), q(dt * Fv / rw2_old[k]), q(dt * dF)))


def solve_lean4(overshoot=.5, max_unbr=100, first_push=None):
    """Newton start, then the secant through the two latest points -- extrapolating while no sign change has been seen (each such
    iterate pushed half a tolerance further, so that a good one brackets the root with its predecessor), the bracketed update of
    lean2 (Anderson-Bjorck) from the first sign change on.  The far end of the reference's bracket is evaluated only when an iterate
    would leave the bracket (or the slope has the wrong sign): no sign change there either -> explicit Euler as the reference."""
    fn = Fun()
    r = rw2_old.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    Fv, dF = fn.F(all_i, rw2_old[all_i], True)
    drw2 = dt * Fv
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b)
    mid = ~early & (a == a_un) & tol_reached(a, b)
    r[all_i[mid]] = (a[mid] + b[mid]) / 2
    go = ~early & ~mid
    idx = all_i[go]; a, b, drw2, rd2, dF = a[go], b[go], drw2[go], rd2[go], dF[go]
    M = idx.size
    x_old = rw2_old[idx]
    grows = drw2 > 0
    far = np.where(grows, b, a)
    dirn = np.where(grows, 1., -1.)
    s = 1. - dt * dF
    c1 = x_old + drw2 / s
    ok = (s > 0) & ((c1 - x_old) * dirn > 0) & ((far - c1) * dirn > 0)
    fp = overshoot if first_push is None else first_push
    c = np.where(ok, c1 + fp * eps * np.abs(c1) * dirn, far)
    c = np.where((far - c) * dirn > 0, c, far)
    x0, f0 = np.full(M, np.nan), np.full(M, np.nan)
    x1, f1 = x_old.copy(), drw2.copy()
    br = np.zeros(M, bool)
    at_far = ~((far - c) * dirn > 0)
    unbr_trips = np.zeros(M, dtype=np.int64)
    res = np.full(M, np.nan)
    act = np.ones(M, bool)
    n_euler = 0
    for it in range(100):
        if not act.any():
            break
        k = np.nonzero(act)[0]
        fc = fn.f(idx[k], c[k])
        opp = (fc < 0) != (f1[k] < 0)
        # the far end without a sign change against the latest point on the near side: explicit Euler
        eu = at_far[k] & ~opp & ~br[k]
        ke = k[eu]
        res[ke] = x_old[ke] + drw2[ke]; act[ke] = False; n_euler += ke.size
        k, fc, opp = k[~eu], fc[~eu], opp[~eu]
        shift = opp | ~br[k]
        m = 1. - fc / f1[k]
        m = np.where(m > 0, m, .5)
        f0[k] = np.where(shift, f1[k], f0[k] * m)
        x0[k] = np.where(shift, x1[k], x0[k])
        x1[k] = c[k]; f1[k] = fc
        br[k] |= opp
        c_new = x1[k] - f1[k] * (x1[k] - x0[k]) / (f1[k] - f0[k])
        res[k] = c_new
        done = (np.abs(c_new - c[k]) <= eps * np.minimum(np.abs(c_new), np.abs(c[k]))) | (br[k] & tol_reached(x0[k], x1[k]))
        # not bracketed yet: push the extrapolated iterate; leave for the far end when it is outside the bracket / moves backwards
        unb = ~br[k]
        unbr_trips[k[unb]] += 1
        cn = np.where(unb, c_new + overshoot * eps * np.abs(c_new) * dirn[k], c_new)
        gofar = unb & (~(((cn - x1[k]) * dirn[k] > 0) & ((far[k] - cn) * dirn[k] > 0)) | (unbr_trips[k] >= max_unbr))
        cn = np.where(gofar, far[k], cn)
        at_far[k] = gofar
        c[k] = cn
        act[k[done & ~gofar]] = False
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    bad = ~((res > lo) & (res < hi))
    res[bad] = x1[bad]
    res = np.maximum(res, rd2)
    r[idx] = res
    print("      (explicit-Euler answers: %d, still active after 100: %d)" % (n_euler, act.sum()))
    return r, fn


def safe_mask(idx, x_old, drw2, a, b, a_un):
    """the far end's sign follows from monotone bounds (see classify_safe for the check of each rule)"""
    irw = 1. / np.sqrt(x_old); rw3 = x_old * x_old * irw
    na, da = rw3 - rd3[idx], rw3 - rd3_1mk[idx]
    klv = np.exp(A[idx] * irw)
    RHe = RH_eff[idx]
    g0 = da * RHe - na * klv
    g1 = da * RHe - na
    base = (da > 0) & (c_Re[idx] >= 0) & (rd3_1mk[idx] < rd3[idx])
    grows = drw2 > 0
    # G1: F(b) < 2 F(a) from monotone bounds (a_w and beta Sh / r grow resp. fall with r, klv >= 1), AND f monotone on the bracket --
    #     dt F' <= (b/a) drw2 [da klv A irw^3 / (2 g0) + (g1/g0) / a] < 1 (the Kelvin term's slope on the falling branch bounded by
    #     its value at the near end, the transfer factor's by H / x)
    U = b * drw2 * (.5 * da * klv * A[idx] * irw + g1) < g0 * x_old * x_old
    G1 = grows & (g1 * b < 2. * g0 * a) & U
    G2 = grows & (RHe <= 1.) & (b < 2. * a)
    unc = a == a_un
    E1 = ~grows & unc & (na > 0) & (3. * x_old * x_old * (da - na) > A[idx] * na * da)
    sfr = (x_old - a) / x_old
    # E3 (falling branch, evaporating): the Kelvin term's growth over the bracket bounded (Bernoulli), and f monotone:
    #     dt F' <= |drw2| da klv' A irw'^3 / (2 |g0|) with the primed values at the far end <= 1.5 x the near end's for sfr <= 1/8
    E3 = ~grows & unc & (sfr <= .125) & (na > 0) & (na * klv * sfr * (klv - 1.) < -g0) & (.75 * -drw2 * da * klv * A[idx] * irw < -g0 * x_old)
    E4 = np.zeros_like(grows)
    return base & (G1 | G2 | E1 | E3 | E4)


def approx_dF(idx, x, Fv):
    """the derivative the kernel can afford: ventilation factors held constant, single precision"""
    f = np.float32
    irw = 1. / np.sqrt(x); rw = x * irw; rw3 = x * rw
    na, da = rw3 - rd3[idx], rw3 - rd3_1mk[idx]
    klv = np.exp(A[idx] * irw)
    g = da * RH_eff[idx] - na * klv
    Re = c_Re[idx] * rw
    m = np.where(Re > 1., np.maximum(1., np.abs(Re) ** .077), 1.)
    Sh, Nu = 1. + np.cbrt(1. + Re * Sc[idx]) * m, 1. + np.cbrt(1. + Re * Pr[idx]) * m
    KnD, KnK = (lam_D[idx] * irw).astype(f), (lam_K[idx] * irw).astype(f)
    nD, dD = f(1) + KnD, f(1) + KnD * (f(1.71) + f(1.33) * KnD)
    nK, dK = f(1) + KnK, f(1) + KnK * (f(1.71) + f(1.33) * KnK)
    tD = (c1[idx]).astype(f) * dD / (nD * Sh.astype(f))
    tK = (c2_rho[idx]).astype(f) * dK / (nK * Nu.astype(f))
    i2x = (.5 / x).astype(f)
    bD = KnD * ((f(1.71) + f(2.66) * KnD) / dD - f(1) / nD)
    bK = KnK * ((f(1.71) + f(2.66) * KnK) / dK - f(1) / nK)
    mix = i2x * (tD * bD + tK * bK) / (tD + tK)
    dg = (1.5 * rw * (RH_eff[idx] - klv)).astype(f) + (na * klv * A[idx] * irw).astype(f) * i2x
    rel = dg / g.astype(f) - (1.5 * rw / da).astype(f) + mix
    return (Fv.astype(f) * rel).astype(np.float64)


def solve_lean5(overshoot=.5, safe_bound=True, newton_first=True, dekker=False, cheap_deriv=False, need_br=False, max_unbr=100):
    """lean4 with the reference's far-end rule kept where it can matter.  After the first evaluation (with derivative):
      SAFE   a growing droplet whose far end provably has the opposite sign -- F(b) < 2 F(a) follows from monotone bounds:
             (da RH - na) b < 2 (da RH - na klv) a  [beta(Kn), Sh, Nu and a_w grow with r at most like r; klv >= 1] -- takes the Newton
             probe path of lean4 without evaluating the far end;
      else   the far end is evaluated as in lean2 (no sign change: explicit Euler), and the FIRST iterate is the Newton point from the
             near end instead of the secant through the bracket ends (stiff haze: the far end is hundreds of radii away)."""
    fn = Fun()
    r = rw2_old.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    Fv, dF = fn.F(all_i, rw2_old[all_i], True)
    drw2 = dt * Fv
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b)
    mid = ~early & (a == a_un) & tol_reached(a, b)
    r[all_i[mid]] = (a[mid] + b[mid]) / 2
    go = ~early & ~mid
    idx = all_i[go]; a, b, drw2, rd2, dF, a_un = a[go], b[go], drw2[go], rd2[go], dF[go], a_un[go]
    M = idx.size
    x_old = rw2_old[idx]
    if cheap_deriv:
        dFx = dF
        dF = approx_dF(idx, x_old, drw2 / dt)
        rel = np.abs(dt * dF - dt * dFx) / np.maximum(1., np.abs(dt * dFx))
        print("      cheap derivative: |dt dF - exact| / max(1, |dt dF|): median %.1e  99%% %.1e  max %.1e" % (np.median(rel), np.percentile(rel, 99), rel.max()))
    grows = drw2 > 0
    far = np.where(grows, b, a)
    dirn = np.where(grows, 1., -1.)
    s = 1. - dt * dF
    c1 = x_old + drw2 / s
    ok = (s > 0) & ((c1 - x_old) * dirn > 0) & ((far - c1) * dirn > 0)
    # the bound
    irw = 1. / np.sqrt(x_old); rw3 = x_old * x_old * irw
    na, da = rw3 - rd3[idx], rw3 - rd3_1mk[idx]
    klv = np.exp(A[idx] * irw)
    g0 = da * RH_eff[idx] - na * klv
    g1 = da * RH_eff[idx] - na
    safe = grows & (g1 * b < 2. * g0 * a) & (da > 0) & ok if safe_bound else np.zeros(M, bool)
    if safe_bound == "full":
        safe = safe_mask(idx, x_old, drw2, a, b, a_un) & ok
    print("      safe (far end not evaluated): %.4f of the droplets that iterate; growing %.4f" % (safe.mean(), grows.mean()))
    x0, f0 = far.copy(), np.full(M, np.nan)
    x1, f1 = x_old.copy(), drw2.copy()
    br = np.zeros(M, bool)
    res = np.full(M, np.nan)
    act = np.ones(M, bool)
    # unsafe droplets: far end first
    u = np.nonzero(~safe)[0]
    ff = fn.f(idx[u], far[u])
    same = (ff < 0) == (drw2[u] < 0)
    res[u[same]] = x_old[u[same]] + drw2[u[same]]; act[u[same]] = False
    ub = u[~same]
    f0[ub] = ff[~same]; br[ub] = True
    sec = x1 - f1 * (x1 - x0) / (f1 - f0)
    c = np.where(safe, c1 + overshoot * eps * np.abs(c1) * dirn, np.where(ok & newton_first, c1, sec))
    c = np.where((far - c) * dirn > 0, c, far)
    at_far = ~((far - c) * dirn > 0)
    n_euler = int(same.sum())
    # dekker: remember the previous same-side point for a secant through the two latest iterates
    xp, fp_ = np.full(M, np.nan), np.full(M, np.nan)
    trips = np.zeros(M, dtype=np.int64)
    for it in range(100):
        if not act.any():
            break
        k = np.nonzero(act)[0]
        fc = fn.f(idx[k], c[k])
        opp = (fc < 0) != (f1[k] < 0)
        eu = at_far[k] & ~opp & ~br[k]
        ke = k[eu]
        res[ke] = x_old[ke] + drw2[ke]; act[ke] = False; n_euler += ke.size
        k, fc, opp = k[~eu], fc[~eu], opp[~eu]
        shift = opp | ~br[k]
        m = 1. - fc / f1[k]
        m = np.where(m > 0, m, .5)
        xp[k] = np.where(shift, np.nan, x1[k]); fp_[k] = np.where(shift, np.nan, f1[k])
        f0[k] = np.where(shift, f1[k], f0[k] * m)
        x0[k] = np.where(shift, x1[k], x0[k])
        x1[k] = c[k]; f1[k] = fc
        br[k] |= opp
        c_new = x1[k] - f1[k] * (x1[k] - x0[k]) / (f1[k] - f0[k])
        if dekker:
            # same side twice inside a bracket: the secant through the two latest iterates when it lands strictly between the latest
            # iterate and the retained end
            cs = x1[k] - f1[k] * (x1[k] - xp[k]) / (f1[k] - fp_[k])
            use = ~np.isnan(cs) & ((cs - x1[k]) * (x0[k] - cs) > 0)
            c_new = np.where(use, cs, c_new)
        res[k] = c_new
        done = (np.abs(c_new - c[k]) <= eps * np.minimum(np.abs(c_new), np.abs(c[k]))) | (br[k] & tol_reached(x0[k], x1[k]))
        if need_br:
            done &= br[k]
        unb = ~br[k]
        trips[k[unb]] += 1
        cn = np.where(unb, c_new + overshoot * eps * np.abs(c_new) * dirn[k], c_new)
        gofar = unb & (~(((cn - x1[k]) * dirn[k] > 0) & ((far[k] - cn) * dirn[k] > 0)) | (trips[k] >= max_unbr))
        cn = np.where(gofar, far[k], cn)
        at_far[k] = gofar
        c[k] = cn
        act[k[done & ~gofar]] = False
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    bad = ~((res > lo) & (res < hi))
    res[bad] = x1[bad]
    res = np.maximum(res, rd2)
    r[idx] = res
    print("      (explicit-Euler answers: %d, still active after 100: %d)" % (n_euler, act.sum()))
    return r, fn


def classify_safe():
    """which droplets' far-end sign follows from monotone bounds; checks the claim and that f is monotone on their brackets"""
    fn = Fun()
    all_i = np.nonzero(rw2_old > 0)[0]
    x = rw2_old[all_i]
    drw2 = dt * fn.F(all_i, x)
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = x + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = x + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b) | ((a == a_un) & tol_reached(a, b))
    safe = safe_mask(all_i, x, drw2, a, b, a_un) & ~early
    print("iterating %d, safe %.4f" % ((~early).sum(), safe.sum() / (~early).sum()))
    k = all_i[safe]
    grows = drw2[safe] > 0
    far = np.where(grows, b[safe], a[safe])
    ff = fn.f(k, far)
    wrong = (ff < 0) == (drw2[safe] < 0)
    print("   far-end sign as claimed: %d violations of %d" % (wrong.sum(), k.size))
    prev = drw2[safe].copy()
    nonmono = np.zeros(k.size, bool)
    for j in range(1, 33):
        xx = x[safe] + (far - x[safe]) * (j / 32.)
        cur = fn.f(k, xx)
        nonmono |= np.where(grows, cur > prev, cur < prev) if False else ((cur - prev) * np.where(grows, 1., -1.) > 0)
        prev = cur
    print("   f not monotone along the bracket (33 samples): %d" % nonmono.sum())
    return all_i, safe
