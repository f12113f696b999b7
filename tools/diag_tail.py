#!/usr/bin/env python3
"""Diagnostics of the condensation tail: which droplets leave the substep tolerance against the oracle, in
both arithmetic modes, at the sizes where it shows (C5 at 16^3 x 512 and the 2^25-droplet box), and the per-formula spread of the
terminal velocities.  Writes gpurun_out/diag_tail_*.npz / .json.  Test infrastructure: loads the oracle as the checker.

    python tools/diag_tail.py [c5] [big] [vt]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as h          # noqa: E402
import bench                  # noqa: E402
from libcloudphxx_amd import lgrngn   # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def step_pair(orc, hip, opts, th, rv, rhod, C):
    tho, rvo, thh, rvh = th.copy(), rv.copy(), th.copy(), rv.copy()
    orc.step_sync(opts, tho, rvo, rhod, **C)
    hip.step_sync(opts, thh, rvh, rhod, **C)
    h.push_coal_replay(orc, hip, orc.opts_init.sstp_coal)
    orc.step_async(opts)
    hip.step_async(opts)
    return (tho, rvo), (thh, rvh)


def run_case(tag, oi, fields, steps, make_oracle):
    res = {}
    for strict in (True, False):
        oi.strict_fp = strict
        t0 = time.time()
        orc, hip = h.make_pair(oi, fields, make_oracle=make_oracle)
        th, rv, rhod, C = fields
        opts = lgrngn.opts_t()
        per_step = []
        for it in range(steps):
            pre = {k: orc.state_real(k) for k in ("rw2", "rd3", "kappa", "vt")}
            pre_n, pre_ijk = orc.state_u64("n"), orc.state_u64("ijk")
            (tho, rvo), (thh, rvh) = step_pair(orc, hip, opts, th, rv, rhod, C)
            ok_n = bool(np.array_equal(hip.state_u64("n"), orc.state_u64("n")))
            ok_sid = bool(np.array_equal(hip.state_u64("sorted_id"), orc.state_u64("sorted_id")))
            ok_ijk = bool(np.array_equal(hip.state_u64("ijk"), orc.state_u64("ijk")))
            rh, ro = hip.get_attr("rw2"), orc.get_attr("rw2")
            err = np.abs(rh / ro - 1)
            bad = np.nonzero(err > 1e-4)[0]
            eth, erv = np.abs(thh / tho - 1), np.abs(rvh / rvo - 1)
            rec = {"n_part": int(ro.size), "n_exact": ok_n, "sorted_id_exact": ok_sid, "ijk_exact": ok_ijk, "n_bad_1e-4": int(bad.size), "n_bad_2e-4": int((err > 2e-4).sum()),
                   "n_bad_1e-6": int((err > 1e-6).sum()), "n_bad_1e-9": int((err > 1e-9).sum()),
                   "median": float(np.median(err)), "max": float(err.max()), "th_max": float(eth.max()), "rv_max": float(erv.max()),
                   "th_bad_1e-7": int((eth > 1e-7).sum()), "rv_bad_1e-6": int((erv > 1e-6).sum())}
            per_step.append(rec)
            print(tag, "strict" if strict else "fast", "step", it, json.dumps(rec), flush=True)
            if bad.size:
                # ids after the step are the ids before it only while nobody died in between (coalescence removes a few): the
                # pre-step state is matched through rd3, unique to the last bit for almost all droplets and untouched by condensation
                pre_rd3 = pre["rd3"]
                order = np.argsort(pre_rd3, kind="stable")
                src = order[np.clip(np.searchsorted(pre_rd3[order], orc.state_real("rd3")[bad]), 0, order.size - 1)]
                np.savez(os.path.join(OUT, "diag_tail_%s_%s_step%d.npz" % (tag, "strict" if strict else "fast", it)),
                         bad=bad, rw2_hip=rh[bad], rw2_orc=ro[bad], rd3_post=orc.state_real("rd3")[bad], kpa_post=orc.state_real("kappa")[bad],
                         src=src, pre_rw2=pre["rw2"][src], pre_rd3=pre_rd3[src], pre_kpa=pre["kappa"][src], pre_vt=pre["vt"][src],
                         pre_n=pre_n[src], pre_ijk=pre_ijk[src], th=th.ravel()[pre_ijk[src]], rv=rv.ravel()[pre_ijk[src]],
                         rhod=rhod.ravel()[pre_ijk[src]], rw2_now=ro[bad])
            h.copy_state(orc, hip)
        res["strict" if strict else "fast"] = {"steps": per_step, "wall_s": time.time() - t0}
        del orc, hip
    json.dump(res, open(os.path.join(OUT, "diag_tail_%s.json" % tag), "w"), indent=1)


def case_c5():
    n = 16
    oi = h.box_opts(n, n, n, 512, kernel=lgrngn.kernel_t.hall_pinsky_stratocumulus)
    run_case("c5_16", oi, h.box_fields(oi), 2, None)


def case_big():
    nx, ny, nz = 128, 128, 32
    oi = bench.make_opts_init(nx, ny, nz, 64, 40., 1, 1, 44)
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(nx, ny, nz, 0, nx, np, np.float64)
    run_case("big_2p25", oi, (th, rv, rhod, {"Cx": Cx, "Cy": Cy, "Cz": Cz}), 2, h.oracle_omp_particles)


def case_vt():
    out = {}
    for vt in (lgrngn.vt_t.beard76, lgrngn.vt_t.beard77, lgrngn.vt_t.beard77fast, lgrngn.vt_t.khvorostyanov_spherical, lgrngn.vt_t.khvorostyanov_nonspherical):
        for strict in (True, False):
            oi = h.box_opts(4, 3, 5, 2048, terminal_velocity=vt, strict_fp=strict)
            orc, hip = h.make_pair(oi, h.box_fields(oi))
            n = orc.n_part
            rw2 = np.exp(np.linspace(np.log(0.5e-6), np.log(3e-3), n)) ** 2
            g = orc.state_real
            args = (orc.state_u64("n"), g("rd3"), rw2, g("kappa"), g("vt"), g("x"), g("y"), g("z"))
            orc.set_particles(*args)
            hip.set_particles(*args)
            orc.stage("hskpng_vterm_all")
            hip.stage("hskpng_vterm_all")
            vo, vh = orc.state_real("vt"), hip.state_real("vt")
            e = np.abs(vh / vo - 1)
            k = int(np.argmax(e))
            out["%s/%s" % (vt.name, "strict" if strict else "fast")] = {"max": float(e.max()), "at_r": float(np.sqrt(rw2[k])), "n_gt_1e-12": int((e > 1e-12).sum()),
                                                                       "n_gt_1e-13": int((e > 1e-13).sum()), "n": n}
            print(vt.name, strict, out["%s/%s" % (vt.name, "strict" if strict else "fast")], flush=True)
    json.dump(out, open(os.path.join(OUT, "diag_vt.json"), "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["vt", "c5", "big"]
    if "vt" in what:
        case_vt()
    if "c5" in what:
        case_c5()
    if "big" in what:
        case_big()
