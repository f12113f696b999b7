"""How far th and rv of the strict arithmetic are from the oracle after a condensation step (the boxes of tests/test_hip_parity.py::test_cond_step
and of the drizzle test): the numbers behind _harness.cond_bars.  Run on the GPU box: python tools/strict_bar_probe.py"""
import os
import sys

import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _harness as h                                            # noqa: E402
from libcloudphxx_amd import lgrngn                             # noqa: E402
import test_hip_parity as tp                                    # noqa: E402

for mode in ("strict", "toms", "fast"):
    for sstp in (1, 4):
        kw, _ = tp._arith(mode)
        oi = h.box_opts(4, 4, 6, 64, sstp_cond=sstp, **kw)
        fields = h.box_fields(oi)
        orc, hip = h.make_pair(oi, fields)
        opts = lgrngn.opts_t()
        opts.coal = opts.adve = opts.sedi = False
        worst = [0., 0., 0.]
        for it in range(3):
            (tho, rvo), (thh, rvh) = tp.step_pair(orc, hip, opts, fields)
            ro, rh = orc.get_attr("rw2"), hip.get_attr("rw2")
            worst = [max(worst[0], np.abs(thh / tho - 1).max()), max(worst[1], np.abs(rvh / rvo - 1).max()), max(worst[2], np.median(np.abs(rh / ro - 1)))]
            h.copy_state(orc, hip)
        print("%-6s sstp %d   th %.2e   rv %.2e   median rw2 %.2e" % (mode, sstp, *worst))
