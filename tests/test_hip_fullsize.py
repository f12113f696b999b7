"""BASELINE-size runs (3-D 128^3 cells x 64 super-droplets = 1.34e8 SDs, BASELINE.json configs[2]) checked through
size-independent properties: sortedness and stability of the permutation, consistency of the CSR offsets with a host
histogram, conservation of water between vapour and droplets, conservation of dry volume under coalescence,
super-droplet bookkeeping across precipitation."""
import numpy as np
import pytest

import bench
import _harness as h
from libcloudphxx_amd import lgrngn

pytestmark = pytest.mark.gpu
N, SD = 128, 64


@pytest.fixture(scope="module", params=[True, False], ids=["strict_fp", "fast_fp"])
def big(request):
    """strict_fp = False is exactly what bench.py runs (the benchmarked configuration is a tested configuration)"""
    oi = bench.make_opts_init(N, N, N, SD, 40., 1, 1, 44)
    oi.strict_fp = request.param
    oi.cond_solver = 0                 # (bench.py's main sets both from its arguments; left alone they are whatever opts_init_t() starts from)
    fields = bench.make_fields(N, N, N, 0, N, np, np.float64)
    pr = lgrngn.factory(lgrngn.backend_t.HIP, oi)
    th, rv, rhod, Cx, Cy, Cz = fields
    pr.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz)
    return pr, oi, fields


def test_init_counts(big):
    pr, oi, _ = big
    assert pr.n_part == N ** 3 * SD
    pr.diag_all()
    pr.diag_sd_conc()
    out = pr.outbuf_array()
    assert out.min() == SD and out.max() == SD


def check_sorted(pr):
    sid = pr.state_u64("sorted_id")
    sijk = pr.state_u64("sorted_ijk")
    ijk = pr.state_u64("ijk")
    cs = pr.state_u64("cell_start")
    n = pr.n_part
    assert len(sid) == n
    d = np.diff(sijk.astype(np.int64))
    assert d.min() >= 0                                        # grouped by cell, ascending
    assert np.array_equal(ijk[sid], sijk)                      # the permutation really is an argsort of ijk
    assert np.all(np.diff(sid.astype(np.int64))[d == 0] > 0)   # stable: ids ascend inside a cell
    chk = np.zeros(n, dtype=bool)
    chk[sid] = True
    assert chk.all()                                           # a permutation: every id exactly once
    assert np.array_equal(np.diff(cs.astype(np.int64)), np.bincount(ijk.astype(np.int64), minlength=pr.n_cell))


def test_full_step_properties(big):
    pr, oi, fields = big
    th, rv, rhod, Cx, Cy, Cz = [f.copy() for f in fields]
    check_sorted(pr)
    opts = lgrngn.opts_t()

    def m3():
        pr.diag_all()
        pr.diag_wet_mom(3)
        return pr.outbuf_array().reshape(rv.shape)

    def dry_total():
        pr.diag_all()
        pr.diag_dry_mom(3)
        return np.sum(pr.outbuf_array() * (rhod.ravel() * oi.dx ** 3))   # specific moment * rhod * dv

    # --- condensation only: water is conserved cell by cell
    rv0 = rv.copy()
    before = m3()
    o = lgrngn.opts_t()
    o.coal = o.adve = o.sedi = False
    pr.step_sync(o, th, rv, rhod, Cx, Cy, Cz)
    # (fast_fp: what bench.py's headline runs -- read back from the object)
    h.assert_mode(pr, *((True, 0, "strict") if oi.strict_fp else (False, 0, ("lean", "lean_sorted"))))
    pr.step_async(o)
    after = m3()
    np.testing.assert_allclose(rv - rv0, -(after - before) * 4. / 3 * np.pi * 1e3, rtol=1e-8, atol=1e-15)
    assert np.abs(rv - rv0).max() > 0
    # --- coalescence only: total dry volume is conserved, SD count can only drop
    d0, n0 = dry_total(), pr.n_part
    o = lgrngn.opts_t()
    o.cond = o.adve = o.sedi = False
    pr.step_sync(o, th, rv, rhod, Cx, Cy, Cz)
    pr.step_async(o)
    assert pr.n_part <= n0
    np.testing.assert_allclose(dry_total(), d0, rtol=1e-10)
    check_sorted(pr)
    # --- everything on: bookkeeping of precipitated / removed SDs
    n_before = pr.n_part
    p_before = pr.diag_puddle()
    pr.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
    pr.step_async(opts)
    check_sorted(pr)
    pr.diag_all()
    pr.diag_sd_conc()
    assert pr.outbuf_array().sum() == pr.n_part <= n_before
    p_after = pr.diag_puddle()
    assert p_after["particle_number"] >= p_before["particle_number"]
    for a in ("x", "y", "z"):
        v = pr.get_attr(a)
        assert v.min() >= 0 and v.max() < N * oi.dx


def test_production_shuffled_order_at_full_size():
    """The PRODUCTION path at BASELINE size (what bench.py times: fast arithmetic, the re-sort's scatter carried by the condensation
    kernel, the in-cell ranking by buckets on its side stream, Philox + bijection keys): after each step_sync the sorted order is read as
    it is ("raw_sorted_id"), and the shuffle keys that the step's coalescence then consumed are read back (lcx_rng_dump).  Properties that
    do not depend on the size: the order is a permutation of the living super-droplets, grouped by ascending cell, the CSR offsets are the
    cells' histogram, and inside every cell the keys ASCEND STRICTLY -- i.e. the order is the reference's shuffle-sort of these keys
    (hskpng_sort.ipp:28-47), for all 1.3e8 of them; the keys of consecutive steps differ (a fresh shuffle per step)."""
    oi = bench.make_opts_init(N, N, N, SD, 40., 1, 1, 44)
    oi.strict_fp = False
    oi.cond_solver = 0
    oi.dbg_flags = int(lgrngn.dbg.TAG)                          # (lcx_rng_dump records what coalescence consumes)
    th, rv, rhod, Cx, Cy, Cz = bench.make_fields(N, N, N, 0, N, np, np.float64)
    pr = lgrngn.factory(lgrngn.backend_t.HIP, oi)
    pr.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz)
    opts = lgrngn.opts_t()
    dead = 0xFFFFFFFF
    last_keys = None
    for it in range(4):
        pr.step_sync(opts, th, rv, rhod, Cx, Cy, Cz)
        h.assert_mode(pr, False, 0, "lean", no_dbg=("NO_DEFERRED_SORT", "COND_SORTED_ORDER", "RANK_BY_COUNTING", "SHUFFLE_PHILOX"))
        order = pr.state_u64("raw_sorted_id").astype(np.int64)
        ijk = pr.state_u64("raw_ijk")
        alive = ijk != dead
        n = pr.n_part
        assert order.size == n == int(alive.sum())
        seen = np.zeros(ijk.size, dtype=bool)
        seen[order] = True
        assert seen.sum() == n and not seen[~alive].any()          # every living super-droplet exactly once, no dead one
        cells = ijk[order].astype(np.int64)
        d = np.diff(cells)
        assert d.min() >= 0                                        # grouped by ascending cell
        cs = np.concatenate([[0], np.cumsum(np.bincount(cells, minlength=N ** 3))])
        pr.step_async(opts)
        keys = pr.rng_dump(0, 1).astype(np.int64)                  # un by storage id, as this step's coalescence drew them
        k = keys[order]
        if it == 0:                                                # (the sort behind init ranks by id; this step's coalescence shuffles for itself)
            assert np.all(np.diff(order)[d == 0] > 0)
        else:                                                      # (from then on the re-sort at the end of a step ranks by the NEXT coalescence's keys)
            assert np.all(np.diff(k)[d == 0] > 0)                  # strictly ascending keys inside every cell
        assert np.unique(np.diff(cs)).size > 1 or it == 0          # (cells differ in occupancy once droplets have moved)
        if last_keys is not None and last_keys.size == keys.size:
            assert (last_keys != keys).mean() > 0.99               # a fresh shuffle every step
        last_keys = keys
