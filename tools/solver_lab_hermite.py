"""Round 6 laboratory (CPU, numpy; see tools/solver_lab.py for the set-up): the lean solver with the CONFIRMING evaluation of its first
secant point replaced by the slope at the near end.

Today (lean2, lcx_math.hpp): f at the near end (rw2_old: f0 = drw2), f at the far end of the reference's bracket, the secant point c,
then f(c) and the next secant point c' -- for 72 % of the droplets only to find |c' - c| <= eps c.  Here the near end's evaluation also
yields f'(near) (the growth rate's analytic derivative, what tools/solver_lab.py approx_dF prices at a quarter of an evaluation in single
precision); with (x0, f0, f0'), (x1, f1) the quadratic through the two points with the given slope has its root c_h next to c (one Newton
step on that quadratic from c); |c_h - c| <= eps min(c_h, c) is the loop's own stopping rule applied to the second-order and the
third-order estimate, and c_h is the answer.  Otherwise the loop goes on from c_h.

    python3 tools/solver_lab_hermite.py [n] [steps] [workload]
"""
import sys
import numpy as np
import solver_lab as L
from solver_lab import Fun, N, rw2_old, rd3, dt, eps, cond_mlt, tol_reached, S


WLIM = 0.1
KRHO = 0.


def solve_lean2h(cheap=True, accept_scale=1.0, small_only=True):
    fn = Fun()
    r = rw2_old.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    Fv, dF = fn.F(all_i, rw2_old[all_i], True)
    fn.devals[all_i] -= 1; fn.evals[all_i] += 1          # counted as an evaluation + the slope's price (report's dcost) below
    slope_n = np.zeros(N)
    if cheap:
        dF = L.approx_dF(all_i, rw2_old[all_i], Fv)
    drw2 = dt * Fv
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b)
    clamp_ok = (rd3[all_i] - L.rd3_1mk[all_i]) * L.RH_eff[all_i] > 1e-12 * rd3[all_i]
    mid = ~early & tol_reached(a, b) & ((a == a_un) | clamp_ok)
    r[all_i[mid]] = (a[mid] + b[mid]) / 2
    go = ~early & ~mid
    idx = all_i[go]; a, b, drw2, rd2, dF = a[go], b[go], drw2[go], rd2[go], dF[go]
    x_old = rw2_old[idx]
    grows = drw2 > 0
    far = np.where(grows, b, a)
    f_far = fn.f(idx, far)
    fa = np.where(grows, drw2, f_far); fb = np.where(grows, f_far, drw2)
    same = fa * fb > 0
    res = np.where(same, x_old + drw2, np.where(fa == 0, a, b))
    loop = ~same & (fa != 0) & (fb != 0)
    x0, f0, x1, f1 = a.copy(), fa.copy(), b.copy(), fb.copy()
    c = x1 - f1 * (x1 - x0) / (f1 - f0)
    # the quadratic through (near, drw2) with slope s = dt F' - 1 and (far, f_far): one Newton step on it from the secant point
    near_is_a = (x_old == a)                               # (a clamped lower end is not the near end: no slope there -> plain path)
    near_ok = np.where(grows, near_is_a, True)
    h = far - x_old
    s = dt * dF - 1.
    kap = 2. * (f_far - drw2 - s * h) / (h * h)
    d = c - x_old
    q = drw2 + d * (s + .5 * kap * d)
    qp = s + kap * d
    ch = c - q / qp
    inside = (ch > np.minimum(x0, x1)) & (ch < np.maximum(x0, x1)) & (qp * s > 0)
    elig = loop & near_ok & inside & (np.abs(h) <= WLIM * x_old)      # (a bracket much narrower than the droplet: the quadratic is a fair model)
    if small_only:                                         # the slope holds the ventilation factors constant: droplets below ~8 um
        xs = L.c_Re[idx] * np.sqrt(x_old) * np.maximum(L.Sc[idx], L.Pr[idx])
        elig &= np.abs(xs) < 2. ** -8
    rho = np.abs(kap * h / s)                              # the slope's relative change over the bracket
    acc = elig & (np.abs(ch - c) * (1. + KRHO * rho) <= accept_scale * eps * np.minimum(np.abs(ch), np.abs(c)))
    res[loop] = c[loop]
    res[acc] = ch[acc]
    c = np.where(elig, ch, c)
    act = loop & ~acc
    naccepted = acc.sum()
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
        conv = (np.abs(c_new - c[k]) <= eps * np.minimum(np.abs(c_new), np.abs(c[k]))) & (np.abs(fc) <= .5 * np.abs(fs))
        done = conv | tol_reached(x0[k], x1[k])
        c[k] = c_new
        act[k[done]] = False
    bad = loop & ~((res > np.minimum(a, b)) & (res < np.maximum(a, b)))
    res[bad] = x1[bad]
    res = np.maximum(res, rd2)
    r[idx] = res
    print("   accepted without a confirming evaluation: %.4f of the droplets in the loop (%d of %d), eligible %.4f" % (naccepted / max(1, loop.sum()), naccepted, loop.sum(), elig.sum() / max(1, loop.sum())))
    return r, fn


if __name__ == "__main__":
    L.check_derivative()
    r2, f2 = L.solve_lean2(guard=True)
    e2 = L.report("lean2 + guard (round 5)", r2, f2)
    # the true root, for an error that is not the reference's own bracket width: lean2 run to a tolerance of 2^-40
    eps_save = L.eps
    for cheap in (False, True):
        for sc in (1.0, 0.5, 0.25):
            rh, fh = solve_lean2h(cheap, sc)
            L.report("lean2h cheap %d accept %.2f eps" % (cheap, sc), rh, fh, dcost=.25)
            rel = np.abs(rh / r2 - 1.)
            print("      vs lean2: max %.2e  99.99%% %.2e  99%% %.2e" % (rel.max(), np.percentile(rel, 99.99), np.percentile(rel, 99)))
