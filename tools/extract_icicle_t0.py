#!/usr/bin/env python3
"""Writes tests/golden/icicle_t0_profiles.npz from the reference-held output of its own icicle regression run
(models/kinematic_2D/tests/paper_GMD_2015/fig_a/refdata/travis_out_lgrngn/travis_timestep0000000000.h5 and travis_const.h5; compared by
the reference at fig_a/CMakeLists.txt:113-126): the state right after particles_t::init of the 2-D lgrngn set-up (60 x 60 grid points,
64 super-droplets per cell, serial backend, real_t = float) as icicle diagnoses it -- per dry-radius bin the specific concentration,
per wet-radius bin likewise, the third wet moment per dry bin -- reduced to DOMAIN MEANS per bin (the fixture is data, 3 KB), plus the
dry-air density profile of const.h5.  Build container only (reads /root/reference through tools/h5min.py).

    python3 tools/extract_icicle_t0.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import h5min  # noqa: E402

REF = "/root/reference/models/kinematic_2D/tests/paper_GMD_2015/fig_a/refdata/travis_out_lgrngn/"


def edges_as_icicle_reads_them(exp0, step, count):
    """bins.hpp:4-20 -> travis_calc_lgrngn.cpp:19-33: float(edge) written with the stream's default six significant digits, parsed again"""
    return np.array([float("%g" % np.float32(1e-6 * 10 ** (exp0 + i * step))) for i in range(count)])


def main():
    f = h5min.File(REF + "travis_timestep0000000000.h5")
    c = h5min.File(REF + "travis_const.h5")
    dry_edges, wet_edges = edges_as_icicle_reads_them(-3, .1, 40), edges_as_icicle_reads_them(-3, .2, 25)
    rd = np.stack([f["rd_rng%03d_mom0" % i].astype(np.float64) for i in range(39)])            # [bin, x, z]
    rw = np.stack([f["rw_rng%03d_mom0" % (i + 2)].astype(np.float64) for i in range(24)])      # (rng 0, 1: the FSSP and rain ranges)
    rw3 = np.stack([f["rw3ofrd_rng%03d_mom3" % i].astype(np.float64) for i in range(39)])
    out = dict(dry_edges=dry_edges, wet_edges=wet_edges,
               rd_mom0_mean=rd.mean(axis=(1, 2)), rw_mom0_mean=rw.mean(axis=(1, 2)), rw3ofrd_mom3_mean=rw3.mean(axis=(1, 2)),
               rd_mom0_level_mean=rd.mean(axis=1), rw_mom0_level_mean=rw.mean(axis=1),
               fssp_mom0_max=np.array([f["rw_rng000_mom0"].max()]), rain_mom0_max=np.array([f["rw_rng001_mom0"].max()]),
               rhod_profile=c["G"][0].astype(np.float64), th=np.array([f["th"].min(), f["th"].max()], dtype=np.float64),
               rv=np.array([f["rv"].min(), f["rv"].max()], dtype=np.float64), sd_conc=np.array([f["sd_conc"].min(), f["sd_conc"].max()], dtype=np.float64),
               shape=np.array(f["th"].shape))
    p = os.path.join(ROOT, "tests", "golden", "icicle_t0_profiles.npz")
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
