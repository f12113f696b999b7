"""CPU checks of the oracle's test hooks for the reverse replay (tests/test_hip_reverse_replay.py runs them against the device):
orc_rng_replay_push (the oracle consumes a queued random stream instead of drawing from its engine), the persistent tag
(LCX_DBG_TAG) and orc_set_state_real."""
import numpy as np
import pytest

import _harness as h
from libcloudphxx_amd import lgrngn


def colliding(oi_kw=None):
    oi = h.box_opts(4, 3, 5, 32, dx=10., **(oi_kw or {}))
    oi.dt = 5.
    oi.dry_distros = {(1e-10, 0.): lgrngn.expvolume(30.084e-6, 2 ** 23)}
    oi.kernel = lgrngn.kernel_t.hall_davis_no_waals
    oi.dbg_flags = int(lgrngn.dbg.TAG)
    return oi


def test_oracle_fed_its_own_stream_repeats_itself_and_tags_follow_the_droplets():
    oi = colliding()
    th, rv, rhod, C = h.box_fields(oi)
    a, b = h.oracle_particles(oi), h.oracle_particles(oi)
    for p in (a, b):
        p.init(th.copy(), rv.copy(), rhod.copy(), **C)
    assert np.array_equal(a.state_real("tag"), np.arange(a.n_part))
    opts = lgrngn.opts_t()
    opts.cond = False
    n0 = a.n_part
    for it in range(8):
        fa, fb = [th.copy(), rv.copy()], [th.copy(), rv.copy()]
        a.step_sync(opts, fa[0], fa[1], rhod, **C)
        b.step_sync(opts, fb[0], fb[1], rhod, **C)
        # what a's engine will draw in this step_async: the shuffle keys, then the uniforms of the candidate pairs
        un, u01 = h.oracle_rng_preview(a, [(1, a.n_part), (0, a.n_part)])
        b.rng_replay_push(1, un)
        b.rng_replay_push(0, u01)
        a.step_async(opts)
        b.step_async(opts)
        assert b.rng_replay_pending() == 0
        for nm in ("n", "ijk", "sorted_id"):
            assert np.array_equal(a.state_u64(nm), b.state_u64(nm)), (it, nm)
        for nm in ("rw2", "rd3", "x", "z", "tag"):
            assert np.array_equal(a.state_real(nm), b.state_real(nm)), (it, nm)
    # super-droplets were used up / fell out: the tags are those of the survivors, still ascending (stable compaction), no longer 0 .. n-1
    tag = a.state_real("tag")
    assert a.n_part < n0 and np.all(np.diff(tag) > 0) and tag[-1] > a.n_part - 1


def test_oracle_replay_queue_and_set_state_real_refuse_what_does_not_fit():
    oi = colliding()
    th, rv, rhod, C = h.box_fields(oi)
    a = h.oracle_particles(oi)
    a.init(th.copy(), rv.copy(), rhod.copy(), **C)
    with pytest.raises(RuntimeError):
        a.rng_replay_push(7, np.zeros(3))
    with pytest.raises(RuntimeError):
        a.set_state_real("rw2", np.zeros(a.n_part + 1))
    with pytest.raises(RuntimeError):
        a.set_state_real("nonsense", np.zeros(a.n_part))
    rw2 = a.state_real("rw2") * 1.5
    a.set_state_real("rw2", rw2)
    assert np.array_equal(a.state_real("rw2"), rw2)
    a.set_state_real("th", th.ravel() + 1.)
    assert np.array_equal(a.state_real("th"), th.ravel() + 1.)


def test_builtin_exponential_spectrum_equals_the_python_function_of_the_references_golovin_test():
    """lcx_distro_t with n_modes = -1 (lgrngn.expvolume) against the function the reference's test passes
    (tests/python/physics/coalescence_golovin.py:41-44), both through the oracle's initialisation"""
    r_zero, n_zero = 30.084e-6, 2 ** 23

    def expvolumelnr(lnr):
        r = np.exp(lnr)
        return n_zero * 3. * np.power(r, 3) / np.power(r_zero, 3) * np.exp(- np.power((r / r_zero), 3))
    res = []
    for fun in (lgrngn.expvolume(r_zero, n_zero), expvolumelnr):
        oi = colliding()
        oi.dry_distros = {(1e-10, 0.): fun}
        th, rv, rhod, C = h.box_fields(oi)
        p = h.oracle_particles(oi)
        p.init(th.copy(), rv.copy(), rhod.copy(), **C)
        res.append((p.state_u64("n"), p.state_real("rd3")))
    assert np.array_equal(res[0][1], res[1][1])
    # (numpy's power / exp against C's pow / exp: the integer multiplicities may differ by one where the product sits on a half)
    d = np.abs(res[0][0].astype(np.int64) - res[1][0].astype(np.int64))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
