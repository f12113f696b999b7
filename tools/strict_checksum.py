"""Checksums of a few strict-arithmetic steps (th, rv, rw2, x, n): run before and after a change that must not alter a bit of the
strict path, compare the two outputs.   python3 tools/strict_checksum.py   (needs the GPU)"""
import sys, hashlib
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _harness as h
from libcloudphxx_amd import lgrngn
for dims, sstp in (((12, 6, 10), 2), ((8, 0, 9), 3)):
    oi = h.box_opts(*dims, 64, sstp_cond=sstp)          # strict_fp default
    th, rv, rhod, C = h.box_fields(oi)
    pr = h.hip_particles(oi)
    pr.init(th, rv, rhod, **C)
    rw2 = pr.get_attr("rw2"); rw2[::9] = (30e-6) ** 2
    pr.set_particles(pr.state_u64("n"), pr.get_attr("rd3"), rw2, pr.get_attr("kappa"), np.full(rw2.size, -1.), pr.get_attr("x"), pr.get_attr("y") if dims[1] else None, pr.get_attr("z"))
    opts = lgrngn.opts_t()
    for _ in range(4):
        pr.step_sync(opts, th, rv, rhod, **C); pr.step_async(opts)
    m = hashlib.sha256()
    for a in (th, rv, pr.get_attr("rw2"), pr.get_attr("x"), pr.state_u64("n")):
        m.update(np.ascontiguousarray(a).tobytes())
    print(dims, m.hexdigest()[:16], pr.n_part)
