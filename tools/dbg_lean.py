import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as h
from libcloudphxx_amd import lgrngn
oi = h.box_opts(12, 10, 14, 64, sstp_cond=2, strict_fp=False)
fields = h.box_fields(oi)
th, rv, rhod, C = fields
res = []
for flags in (0, int(lgrngn.dbg.COND_LEAN_R3)):
    oi.dbg_flags = flags
    hip = h.hip_particles(oi)
    hip.init(th, rv, rhod, **C)
    rw2 = hip.get_attr("rw2")
    rw2[::7] = (60e-6) ** 2
    hip.set_particles(hip.state_u64("n"), hip.get_attr("rd3"), rw2, hip.get_attr("kappa"), np.full(rw2.size, -1.),
                      hip.get_attr("x"), hip.get_attr("y"), hip.get_attr("z"))
    opts = lgrngn.opts_t()
    opts.coal = opts.adve = opts.sedi = False
    thh, rvh = th.copy(), rv.copy()
    out = []
    for it in range(3):
        hip.step_sync(opts, thh, rvh, rhod, **C)
        hip.step_async(opts)
        out.append((hip.get_attr("rw2").copy(), thh.copy(), rvh.copy(), hip.get_attr("rd3"), hip.state_u64("ijk"), hip.state_real("vt")))
    res.append((out, rw2))
for it in range(3):
    a, b = res[0][0][it], res[1][0][it]
    bad = np.nonzero(a[0] != b[0])[0]
    print("step", it, "rw2 mismatches", bad.size, "of", a[0].size, "th equal", np.array_equal(a[1], b[1]), "rv equal", np.array_equal(a[2], b[2]))
    for i in bad[:8]:
        prev = res[0][1][i] if it == 0 else res[0][0][it - 1][0][i]
        print("   i", i, "rw2_old %.17g" % prev, "new %.17g r3 %.17g" % (a[0][i], b[0][i]), "rel", abs(a[0][i] / b[0][i] - 1), "rd3 %.6g" % a[3][i], "cell", a[4][i], "vt", a[5][i])
    if bad.size:
        break
