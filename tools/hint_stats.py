"""How well does a droplet's root-finder iteration count of one step predict the next one's?  (k_cond_lean deals a workgroup's droplets
to its waves by it.)  python tools/hint_stats.py [n]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libcloudphxx_amd import lgrngn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
oi = bench.make_opts_init(n, n, n, 64, 40., 1, 1, 44)
oi.strict_fp = False
oi.reorder_every = -1 if len(sys.argv) > 2 else 0
th, rv, rhod, Cx, Cy, Cz = bench.make_fields(n, n, n, 0, n, np, np.float64)
pr = lgrngn.factory(lgrngn.backend_t.HIP, oi)
pr.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz)
opts = lgrngn.opts_t()
prev = None
for it in range(14):
    pr.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
    hint = pr.state_u64("raw_cond_hint").astype(np.int64)
    ijk = pr.state_u64("raw_ijk")
    alive = ijk != 0xFFFFFFFF
    pr.step_async(opts)
    if it >= 2:
        hh = np.bincount(np.minimum(hint[alive], 8), minlength=9) / alive.sum()
        msg = "step %2d  k hist %s" % (it, np.round(hh, 3))
        if prev is not None and prev.size == hint.size:
            both = alive
            same = (prev[both] == hint[both]).mean()
            fast_prev, fast_now = prev[both] <= 1, hint[both] <= 1
            msg += "  same %.3f  P(fast now | fast before) %.4f  P(slow now | slow before) %.3f  fast share %.3f" % (
                same, (fast_now & fast_prev).sum() / max(fast_prev.sum(), 1), (~fast_now & ~fast_prev).sum() / max((~fast_prev).sum(), 1), fast_now.mean())
            # waves of 64 consecutive storage slots: max k per wave, as stored and after the workgroup's deal (256 slots)
            m = hint.size // 256 * 256
            k = np.where(alive[:m], hint[:m], 0).reshape(-1, 4, 64)
            plain = k.max(axis=2).mean()
            pk = np.where(alive[:m], prev[:m], 0).reshape(-1, 256)
            order = np.argsort(pk > 1, axis=1, kind="stable")
            dealt = np.take_along_axis(k.reshape(-1, 256), order, axis=1).reshape(-1, 4, 64).max(axis=2).mean()
            ideal = np.sort(k.reshape(-1, 256), axis=1).reshape(-1, 4, 64).max(axis=2).mean()
            msg += "  mean k %.2f  wave max: plain %.2f dealt %.2f (ideal sort %.2f)" % (k.mean(), plain, dealt, ideal)
        print(msg, flush=True)
    prev = hint
