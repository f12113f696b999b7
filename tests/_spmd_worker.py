"""Worker of tests/test_hip_spmd.py: one rank of libcloudphxx_amd.multi.particles_multi_t with the HIP ENGINE (the product) under
torch.distributed -- gloo with the host-staged transport, so that several ranks can share the one GPU of the test box; the protocol
(device-side counts, one message per direction, one host synchronisation per step in the engine) is the one an RCCL run takes.

    python _spmd_worker.py <mode> <rank> <world> <port> <result-path-pattern> [transport]
mode "ring":  tests/mpi/mpi_adve_test.cpp:196-255 -- nx steps at Courant number 1 take every super-droplet once around the ring;
mode "steps": three full steps (condensation, coalescence, advection, sedimentation) of a 3-D box, the slab's state saved for the
              comparison with the native multi-device object's slabs;
mode "uneven": nx not divisible by the number of ranks, Courant number close to 1 (the thin slabs send nearly a whole plane to the
              thick one) -- conservation and state saved for the comparison with the native object."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def box(nx, ny, nz, sd_conc, seed, **kw):
    import _harness as h
    from libcloudphxx_amd import lgrngn
    oi = h.box_opts(nx, ny, nz, sd_conc, dx=20., **kw)
    oi.dry_distros = {(.61, 0.): lgrngn.lognormal([.02e-6, .075e-6], [1.4, 1.6], [60e6, 40e6])}
    oi.n_sd_max = sd_conc * nx * max(ny, 1) * nz * 3
    oi.rng_seed = seed
    return oi


def slab_state(p):
    st = {k: p.state_u64(k) for k in ("n", "ijk", "sorted_id")}
    st.update({k: p.get_attr(k) for k in ("rd3", "rw2", "x", "z")})
    st["n_part"] = np.array([p.n_part])
    return st


def run(mode, rank, world, port, res, transport):
    if os.environ.get("LCX_SPMD_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=True)
    import torch
    import torch.distributed as dist
    import _harness as h
    from libcloudphxx_amd import lgrngn, multi
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # RCCL refuses two ranks on one device: on the one-GPU test box it carries the messages of ONE rank whose neighbours are the rank
    # itself (self_ring) -- every send and receive of the protocol is a real RCCL operation on the engine's stream
    self_ring = mode.startswith("self")
    if self_ring:
        assert world == 1
        mode = mode[4:]
    dist.init_process_group("nccl" if transport == "rccl" else "gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            **({"device_id": dev} if transport == "rccl" else {}))
    try:
        if mode == "ring":
            nx, nz = 8, 4
            oi = lgrngn.opts_init_t()
            oi.dry_distros = {(.61, 0.): lgrngn.lognormal([.02e-6], [1.4], [60e6])}
            oi.coal_switch = oi.sedi_switch = False
            oi.dt = 1
            oi.nx, oi.nz, oi.dx, oi.dz = nx, nz, 1, 1
            oi.x1, oi.z1 = nx, nz
            oi.sd_conc = 80                         # 320 per plane: the first message exceeds its one-tile first part
            oi.n_sd_max = 80 * nx * nz * 2
            oi.adve_scheme = lgrngn.as_t.euler
            oi.rng_seed = 44 + rank
            oi.dev_id = 0
            prt = multi.particles_multi_t(oi, np.float64, device=dev, transport=transport, self_ring=self_ring)
            assert prt.transport == transport
            nxl = prt.opts_init.nx
            th, rv, rhod = 300. * np.ones((nxl, nz)), .01 * np.ones((nxl, nz)), np.ones((nxl, nz))
            Cx, Cz = np.ones((nxl + 1, nz)), np.zeros((nxl, nz + 1))
            prt.init(th, rv, rhod, Cx=Cx, Cz=Cz)
            opts = lgrngn.opts_t()
            opts.cond = opts.coal = opts.sedi = False

            def diags():
                out = []
                prt.diag_all(); prt.diag_sd_conc(); out.append(prt.outbuf_array())
                for fn, k in ((prt.diag_dry_mom, 1), (prt.diag_wet_mom, 1), (prt.diag_kappa_mom, 1)):
                    prt.diag_all(); fn(k); out.append(prt.outbuf_array())
                return np.stack(out)
            before, n_before = diags(), prt.n_part
            for step in range(nx):
                prt.step_sync(opts, th, rv, rhod, Cx, None, Cz)
                prt.step_async(opts)
                if os.environ.get("LCX_SPMD_TRACE"):
                    print("rank", rank, "step", step, "n_part", prt.n_part, "second_rounds", prt.second_rounds, flush=True)
            after = diags()
            np.save(res % rank, np.stack([before, after]))
            ok = np.array_equal(before, after) and prt.n_part == n_before and prt.bytes_moved > 0 and 1 <= prt.second_rounds <= 2
            sys.exit(0 if ok else 3)
        full = mode in ("steps", "fsteps")               # ("fsteps": fast arithmetic -- the slabs leave their re-sort to the next condensation kernel)
        nx, ny, nz = (8, 3, 4) if full else (7, 0, 5)
        oi = box(nx, ny, nz, 24, 44 + rank, coal_switch=full, strict_fp=(mode != "fsteps"))
        th, rv, rhod, C = h.box_fields(oi)
        if mode == "uneven":
            C["Cx"] = 0.95 * np.ones_like(C["Cx"])
        oi.dev_id = 0
        prt = multi.particles_multi_t(oi, np.float64, device=dev, transport=transport, global_arrays=True, self_ring=self_ring)
        prt.init(th, rv, rhod, **C)
        opts = lgrngn.opts_t()
        opts.coal = full and not self_ring
        n_tot0 = None
        for it in range(3 if full else 5):
            prt.step_sync(opts, th, rv, rhod, **C)       # (global arrays: every rank writes its planes of its own copy)
            prt.step_async(opts)
        st = slab_state(prt.prt)
        b, n = prt.n_x_bfr, prt.opts_init.nx
        st["th"], st["rv"] = th[b:b + n], rv[b:b + n]
        st["host_syncs"] = np.array([prt.host_syncs])
        st["bytes_moved"] = np.array([prt.bytes_moved])
        st["y"] = prt.get_attr("y") if ny else np.zeros(0)
        np.savez(res % rank, **st)
        sys.exit(0)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run(a[1], int(a[2]), int(a[3]), int(a[4]), a[5], a[6] if len(a) > 6 else "host")
