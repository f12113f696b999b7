"""bench.py's contract with the driver: ONE JSON line on stdout with the agreed keys (the roofline and cpu_baseline objects included),
and the host-side helpers of its CPU leg.  The GPU test runs the real script on a small box."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def test_cpu_baseline_leg_on_a_small_sample():
    import bench
    a = types.SimpleNamespace(cpu_sample_n=6, sd_conc=16, dx=40., sstp_cond=1, sstp_coal=1, cpu_sample_steps=1, workload="stratocumulus", dt=1., kernel=None)
    r = bench.cpu_baseline(a)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in r, k
    assert r["kind"] == "port" and r["unit"] == "super-droplets/s" and r["value"] > 0
    assert 1 <= r["cores"] <= (os.cpu_count() or 1) and r["cores"] <= r["cpu_quota"]
    assert "6^3" in r["sample"]


def test_cpu_quota_is_a_positive_count_within_the_affinity_mask():
    import _harness as h
    q = h.cpu_quota()
    assert 1 <= q <= len(os.sched_getaffinity(0))


def test_synthetic_fields_have_the_shapes_the_api_wants():
    import bench
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(5, 4, 3, 0, 5, np, np.float64)
    shapes = [np.broadcast_to(t, s).shape for t, s in zip((th, rv, rhod, Cx, Cy, Cz),
                                                           [(5, 4, 3)] * 3 + [(6, 4, 3), (5, 5, 3), (5, 4, 4)])]
    assert shapes == [(5, 4, 3)] * 3 + [(6, 4, 3), (5, 5, 3), (5, 4, 4)]
    assert np.abs(np.asarray(Cx)).max() <= 0.3 + 1e-12
    oi = bench.make_opts_init(5, 4, 3, 64, 40., 1, 1, 44)
    assert (oi.nx, oi.ny, oi.nz, oi.sd_conc) == (5, 4, 3, 64) and oi.n_sd_max >= 5 * 4 * 3 * 64


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--self-ring"]])
def test_bench_prints_one_json_line_with_the_agreed_keys(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", "16", "--steps", "4", "--warmup", "1", "--cpu-sample-n", "6", "--cpu-sample-steps", "1",
           "--strict-leg-steps", "2", "--leg-steps", "2", "--stage-steps", "2"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines                      # (libraries that print on their own -- RCCL's banner -- go to stderr)
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "super-droplets/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["super_droplets"] - 1) < 0.05
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if not extra:
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert "strict_fp" in d and d["strict_fp"]["ms_per_step"] > 0
        # the other legs: the API's default arithmetic (round 5: fast, the reference's TOMS748 iterates) with device and with host arrays
        # (what an unchanged caller passes), the headline with host arrays, a coalescence that collides, and 512 per cell (C5)
        for leg in ("api_default", "api_default_host_arrays", "host_arrays", "coal_stress", "c5"):
            assert d[leg]["ms_per_step"] > 0, leg
        assert d["coal_stress"]["collided_pairs_per_step"] > 0 and d["coal_stress"]["coal_ms"] > 0
        assert abs(d["c5"]["value"] * d["c5"]["ms_per_step"] * 1e-3 / (16 ** 3 * 512) - 1) < 0.05


@pytest.mark.gpu
def test_bench_coal_stress_workload_reports_collisions():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "coal-stress", "--n", "16", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--no-strict-leg", "--no-toms-leg", "--no-host-leg", "--stage-steps", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.split("\n") if l.strip()][0])
    c = d["coal_stress"]
    # a per cent of the candidate pairs collide per second in this spectrum (SURVEY 8d, C5 variant ii)
    assert c["collided_pairs_per_step"] > 0 and 1e-4 < c["share_of_candidate_pairs"] < 0.2, c
    assert d["stage_roofline"]["coal"]["bytes_per_sd"] > 2 * 4 + 4 + 8 + 3 * 8
