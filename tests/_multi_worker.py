"""Worker for the multi-process tests: the 1-D slab decomposition and neighbour exchange of
libcloudphxx_amd.multi driven with the CPU oracle as the per-rank particle engine (gloo backend).
Restates tests/mpi/mpi_adve_test.cpp:196-255: after nx_total steps of C = +-1 advection every SD has
travelled once around the periodic ring and every per-cell diagnostic must be bit-identical."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def run(rank, world, port, nx, nz, Cx_val, result_path, scheme="euler", sd_conc=8):
    import torch.distributed as dist
    import _harness as h
    from libcloudphxx_amd import lgrngn, multi
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        oi = lgrngn.opts_init_t()
        oi.dry_distros = {(.61, 0.): h.lognormal_fn(.04e-6 / 2, 1.4, 60e6)}
        oi.coal_switch = oi.sedi_switch = False
        oi.dt = 1
        oi.nx, oi.nz, oi.dx, oi.dz = nx, nz, 1, 1
        oi.x1, oi.z1 = nx * oi.dx, nz * oi.dz
        oi.sd_conc = sd_conc
        oi.n_sd_max = sd_conc * nx * nz * 2
        oi.adve_scheme = lgrngn.as_t[scheme]         # pred_corr: the Courant halo is exchanged between the ranks as well
        oi.rng_seed = 44 + rank                      # mpi_adve_test.cpp:95 seeds every rank differently
        # one rank: a ring of one (its neighbours are the rank itself) -- the protocol with every message delivered to the sender
        prt = multi.particles_multi_t(oi, np.float64, make_particles=h.oracle_particles, self_ring=(world == 1))
        nxl = prt.opts_init.nx
        assert sum(multi.get_dev_nx(nx, r, world) for r in range(world)) == nx
        th, rv, rhod = 300. * np.ones((nxl, nz)), .01 * np.ones((nxl, nz)), np.ones((nxl, nz))
        Cx, Cz = Cx_val * np.ones((nxl + 1, nz)), np.zeros((nxl, nz + 1))
        prt.init(th, rv, rhod, Cx=Cx, Cz=Cz)
        opts = lgrngn.opts_t()
        opts.cond = opts.coal = opts.sedi = False

        def diags():
            out = []
            prt.diag_all(); prt.diag_sd_conc(); out.append(prt.outbuf_array())
            for fn, k in ((prt.diag_dry_mom, 1), (prt.diag_wet_mom, 1), (prt.diag_kappa_mom, 1)):
                prt.diag_all(); fn(k); out.append(prt.outbuf_array())
            return np.stack(out)
        before = diags()
        n_before = prt.n_part
        moved = 0
        for step in range(nx):
            prt.step_sync(opts, th, rv, rhod, Cx, None, Cz)
            prt.step_async(opts)
        after = diags()
        np.save(result_path % rank, np.stack([before, after]))
        ok = np.array_equal(before, after) and prt.n_part == n_before and prt.bytes_moved > 0
        # a whole x-plane crosses each face per step: with more than a tile (256) of super-droplets per plane the first message exceeds
        # its agreed first part and the protocol's second batch must have run (once: the sender then announces a larger first part)
        if sd_conc * nz > 256:
            ok = ok and prt.second_rounds >= 1 and prt.second_rounds <= 2
        else:
            ok = ok and prt.second_rounds == 0
        sys.exit(0 if ok else 3)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run(int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]), a[7], a[8] if len(a) > 8 else "euler", int(a[9]) if len(a) > 9 else 8)
