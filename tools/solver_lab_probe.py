"""solver_lab.py's droplets through `lean6`: ONE Newton probe where the far end's sign is proven (safe_mask) and the probe confirms itself,
everything else (unproven, unconfirmed) handed to the bracketed solver in full (lean2 with the guard) -- the form a workgroup fold or a wave
queue can run: phase 1 at full lanes (near end + classification + slope + probe), phase 2 dense.
    python3 tools/solver_lab_probe.py [n=32] [steps=24] [workload]
Prints the share of droplets that phase 1 settles, the folded droplets per workgroup of 256 storage neighbours, the error against the
oracle's TOMS748 answers and against lean2's."""
import sys
import numpy as np
import solver_lab as L

N, eps, dt, cond_mlt = L.N, L.eps, L.dt, L.cond_mlt
rw2_old, rd3 = L.rw2_old, L.rd3


def solve_lean6(overshoot=.5, guard=.5, cheap=True, full_mask=True):
    fn = L.Fun()
    r2, f2 = L.solve_lean2(True)                       # phase 2's answers (and the fallback for everybody)
    r = r2.copy()
    all_i = np.nonzero(rw2_old > 0)[0]
    Fv, dF = fn.F(all_i, rw2_old[all_i], True)
    drw2 = dt * Fv
    rd2 = np.cbrt(rd3[all_i]) ** 2
    a_un = rw2_old[all_i] + np.minimum(0., cond_mlt * drw2)
    a = np.maximum(rd2, a_un)
    b = rw2_old[all_i] + np.maximum(0., cond_mlt * drw2)
    early = (drw2 == 0) | (a == b) | ((a == a_un) & L.tol_reached(a, b))
    go = ~early
    idx = all_i[go]; a, b, drw2, dF, a_un = a[go], b[go], drw2[go], dF[go], a_un[go]
    x_old = rw2_old[idx]
    if cheap:
        dF = L.approx_dF(idx, x_old, drw2 / dt)
    grows = drw2 > 0
    far = np.where(grows, b, a)
    dirn = np.where(grows, 1., -1.)
    s = 1. - dt * dF
    c1 = x_old + drw2 / s
    ok = (s > 0) & ((c1 - x_old) * dirn > 0) & ((far - c1) * dirn > 0)
    if full_mask:
        safe = L.safe_mask(idx, x_old, drw2, a, b, a_un) & ok
    else:
        irw = 1. / np.sqrt(x_old); rw3 = x_old * x_old * irw
        na, da = rw3 - rd3[idx], rw3 - L.rd3_1mk[idx]
        klv = np.exp(L.A[idx] * irw)
        g0 = da * L.RH_eff[idx] - na * klv; g1 = da * L.RH_eff[idx] - na
        safe = grows & (g1 * b < 2. * g0 * a) & (da > 0) & ok
    c = c1 + overshoot * eps * np.abs(c1) * dirn
    c = np.where((far - c) * dirn > 0, c, c1)
    k = np.nonzero(safe)[0]
    fc = fn.f(idx[k], c[k])
    f_old = drw2[k]
    c_new = c[k] - fc * (c[k] - x_old[k]) / (fc - f_old)
    acc = (np.abs(c_new - c[k]) <= eps * np.minimum(np.abs(c_new), np.abs(c[k]))) & (np.abs(fc) <= guard * np.abs(f_old))
    acc &= (c_new > np.minimum(a[k], b[k])) & (c_new < np.maximum(a[k], b[k]))
    ka = k[acc]
    r[idx[ka]] = np.maximum(c_new[acc], np.cbrt(rd3[idx[ka]]) ** 2)
    settled = np.zeros(N, bool)
    settled[all_i[early]] = True
    settled[idx[ka]] = True
    settled[rw2_old <= 0] = True
    it = go.sum()
    print("iterating %.4f of all; of those: safe %.4f, accepted after the probe %.4f (%.4f of the safe)" %
          (it / N, safe.sum() / it, ka.size / it, ka.size / max(1, safe.sum())))
    W = N // 256 * 256
    fold = (~settled[:W]).reshape(-1, 256).sum(axis=1)
    print("folded per workgroup of 256: mean %.1f  median %d  90%% %d  99%% %d  max %d;  P(> 64) %.4f  P(> 128) %.4f" %
          (fold.mean(), np.median(fold), np.percentile(fold, 90), np.percentile(fold, 99), fold.max(), (fold > 64).mean(), (fold > 128).mean()))
    fw = (~settled[:W]).reshape(-1, 64).sum(axis=1)
    print("unsettled per wave of 64: mean %.2f, P(0) %.4f" % (fw.mean(), (fw == 0).mean()))
    ref = L.S["rw2_ref"]
    okk = rw2_old > 0
    rel = np.abs(r[okk] / ref[okk] - 1.)
    rel2 = np.abs(r[okk] / r2[okk] - 1.)
    print("vs oracle: max %.2e  99.99%% %.2e  median %.2e;   vs lean2: max %.2e  99.99%% %.2e" %
          (rel.max(), np.percentile(rel, 99.99), np.median(rel), rel2.max(), np.percentile(rel2, 99.99)))
    rel0 = np.abs(r2[okk] / ref[okk] - 1.)
    print("lean2 vs oracle: max %.2e  99.99%% %.2e  median %.2e" % (rel0.max(), np.percentile(rel0, 99.99), np.median(rel0)))
    # evaluations of phase 2 per folded droplet (lean2's count) and their per-wave maximum when packed densely in storage order
    e2 = f2.evals
    un = np.nonzero(~settled)[0]
    ev = e2[un]
    Wd = un.size // 64 * 64
    print("phase 2: %.4f of all droplets, %.2f evaluations each, %.2f for the slowest of 64 packed" %
          (un.size / N, ev.mean(), ev[:Wd].reshape(-1, 64).max(axis=1).mean() if Wd else 0.))
    return r


if __name__ == "__main__":
    for kw in (dict(), dict(overshoot=0.), dict(cheap=False), dict(full_mask=False), dict(guard=1.)):
        print("---", kw)
        solve_lean6(**kw)
