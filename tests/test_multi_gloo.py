"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the slab decomposition + neighbour exchange protocol
(libcloudphxx_amd/multi.py) with the oracle as the per-rank engine."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_distmem_opts_follow_reference_split():
    from libcloudphxx_amd import lgrngn, multi
    # src/detail/distmem_opts.hpp:10-16: `nx / size + .5` with an INTEGER division = floor(nx/size) to all but the last rank,
    # the remainder to the last
    assert [multi.get_dev_nx(5, r, 2) for r in range(2)] == [2, 3]
    assert [multi.get_dev_nx(6, r, 4) for r in range(4)] == [1, 1, 1, 3]
    assert [multi.get_dev_nx(9, r, 6) for r in range(6)] == [1, 1, 1, 1, 1, 4]
    assert [multi.get_dev_nx(128, r, 8) for r in range(8)] == [16] * 8
    assert [multi.get_dev_nx(7, r, 3) for r in range(3)] == [2, 2, 3]
    oi = lgrngn.opts_init_t()
    oi.nx, oi.dx, oi.x1, oi.n_sd_max = 7, 2., 14., 700
    o1, bfr = multi.distmem_opts(oi, 2, 3)
    assert (o1.nx, bfr, o1.x0, o1.x1, o1.n_sd_max) == (3, 4, 0., 6., 700 // 3 + 1)
    assert o1.bcond_lft == o1.bcond_rgt == multi.BCOND_DISTMEM
    oi.open_side_walls = True
    o0, _ = multi.distmem_opts(oi, 0, 3)
    assert (o0.bcond_lft, o0.bcond_rgt) == (multi.BCOND_OPEN, multi.BCOND_DISTMEM)


@pytest.mark.parametrize("world,nx,Cx,scheme,sd_conc", [(2, 6, 1., "euler", 8), (2, 5, -1., "euler", 8), (3, 7, 1., "euler", 8), (2, 8, 1., "pred_corr", 8),
                                                        (3, 10, -1., "pred_corr", 8), (2, 6, 1., "euler", 80), (3, 7, -1., "euler", 80),
                                                        (1, 5, 1., "euler", 80), (1, 6, -1., "pred_corr", 8)])
def test_ring_round_trip_is_bit_identical(world, nx, Cx, scheme, sd_conc, tmp_path):
    """sd_conc = 80: 320 super-droplets cross each face per step, more than the one-tile first part the ranks start with -- the
    protocol's second batch (libcloudphxx_amd/multi.py) runs in the first step and the announced capacity covers the later ones"""
    port = free_port()
    res = str(tmp_path / "r%d.npy")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_multi_worker.py"), str(r), str(world), str(port), str(nx), "4", str(Cx), res, scheme, str(sd_conc)])
             for r in range(world)]
    from test_hip_spmd import wait_all
    codes = wait_all(procs, 240)
    assert codes == [0] * world, codes
    for r in range(world):
        d = np.load(res % r)
        assert np.array_equal(d[0], d[1])
        assert d[0][0].sum() > 0
