"""1-D domain decomposition across the GPUs of one node: one process per GPU, neighbour exchange of
migrating super-droplets with torch.distributed point-to-point operations (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

This is the SPMD flavour (one PROCESS per GPU, e.g. under torch.distributed.run or next to an MPI host model); the reference's
multi_CUDA object -- ONE process, all GPUs of the node -- is native in the C library (lcx_create_multi, csrc/lcx_multi.hpp) and is
what factory(multi_CUDA | multi_HIP) returns.  It replaces the reference's multi_CUDA backend (src/particles_multi_gpu_*.ipp,
src/impl_multi_gpu/particles_multi_gpu_impl_step_async_and_copy.ipp:28-206: one std::thread per GPU,
cudaMemcpyPeerAsync of two packed buffers, five thread barriers per step) and its MPI twin
(src/impl/distributed_memory/particles_impl_mpi_exchange.ipp:20-330).  There is no collective on the data
path: each rank talks to its left and right neighbour only (periodic ring, or open ends with
open_side_walls), one message pair per direction per step:  counts first, then the attribute-major packed
records produced on the device by lcx_migrate_pack (include/lcx.h).

Slab sizes follow detail::get_dev_nx / distmem_opts (src/detail/distmem_opts.hpp:10-52).
"""
import copy

import numpy as np

from . import lgrngn

BCOND_SHAREDMEM, BCOND_DISTMEM, BCOND_OPEN = 0, 1, 3      # src/detail/bcond.hpp


def get_dev_nx(nx, rank, size):
    """distmem_opts.hpp:10-16.  `opts_init.nx / size + .5` is an INTEGER division there (both operands are int), so every
    rank but the last gets floor(nx / size) planes and the last one the remainder."""
    per = nx // size
    n = per if rank < size - 1 else nx - rank * per
    if n <= 0:
        raise RuntimeError("libcloudph++: number of devices exceeds nx")
    return n


def distmem_opts(opts_init, rank, size):
    """Per-rank copy of opts_init for the slab owned by `rank` (distmem_opts.hpp:20-52).
    Returns (local opts_init, n_x_bfr)."""
    oi = copy.copy(opts_init)
    oi.dry_distros = dict(opts_init.dry_distros)
    n_x_bfr = rank * get_dev_nx(opts_init.nx, 0, size)
    oi.nx = get_dev_nx(opts_init.nx, rank, size)
    if rank != 0:
        oi.x0 = 0.
    if rank != size - 1:
        oi.x1 = oi.nx * oi.dx
    else:
        oi.x1 = opts_init.x1 - n_x_bfr * opts_init.dx
    oi.n_sd_max = opts_init.n_sd_max // size + 1
    oi.n_x_tot = opts_init.nx
    oi.n_x_bfr = n_x_bfr
    if size > 1:
        if not opts_init.open_side_walls:
            oi.bcond_lft = oi.bcond_rgt = BCOND_DISTMEM
        else:
            oi.bcond_lft = BCOND_OPEN if rank == 0 else BCOND_DISTMEM
            oi.bcond_rgt = BCOND_OPEN if rank == size - 1 else BCOND_DISTMEM
    return oi, n_x_bfr


class particles_multi_t:
    """SPMD flavour of particles_t<real_t, multi_CUDA>: every rank constructs it with the GLOBAL opts_init;
    arrays passed to init/step_sync are the rank's LOCAL slabs (x-planes [n_x_bfr, n_x_bfr + nx_local)) unless
    global_arrays=True, in which case the library indexes the global arrays with the n_x_bfr offset exactly like
    the reference does (initialization/particles_impl_init_e2l.ipp:44-46).

    make_particles(opts_init_local) -> particles object (defaults to the HIP backend); make_buffer(nbytes) ->
    (object keeping the buffer alive, address, torch tensor view) lets the CPU tests run the same protocol on gloo.
    """

    def __init__(self, opts_init, real_t=np.float64, make_particles=None, device=None, global_arrays=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            raise RuntimeError("libcloudph++: multi_HIP needs torch.distributed to be initialised (one process per GPU)")
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        if opts_init.nx == 0:
            raise RuntimeError("libcloudph++: multi_CUDA backend works only for nx>0")       # particles_multi_gpu_impl.ipp:49
        if self.size > opts_init.nx:
            raise RuntimeError("libcloudph++: number of devices exceeds nx")                  # :62
        self.real_t = np.dtype(real_t)
        self.glob_opts_init = opts_init
        oi, self.n_x_bfr = distmem_opts(opts_init, self.rank, self.size)
        if not global_arrays:
            oi.n_x_bfr = 0
            oi.n_x_tot = oi.nx        # slab-local arrays: a Courant halo wraps inside the slab, the exchange below overwrites it
        self.opts_init = oi
        if device is None and make_particles is None:
            # the HIP engine works on device buffers: with the default engine the rank's current GPU is the device (a host tensor's
            # address handed to the pack / unpack kernels would fault)
            if not torch.cuda.is_available():
                raise RuntimeError("libcloudph++: multi_HIP needs a GPU per process (pass make_particles / device for another engine)")
            device = torch.device("cuda", torch.cuda.current_device())
            oi.dev_id = device.index
        self.device = device
        if make_particles is None:
            make_particles = lambda o: lgrngn.particles_t(o, real_t)
        self.prt = make_particles(oi)
        self.on_gpu = device is not None and str(device).startswith("cuda")
        periodic = not opts_init.open_side_walls
        self.lft = (self.rank - 1) % self.size if (periodic or self.rank > 0) else None
        self.rgt = (self.rank + 1) % self.size if (periodic or self.rank < self.size - 1) else None
        if self.size == 1:
            self.lft = self.rgt = None
        # neighbours' domain edges in THEIR local frames (xchng_domains.ipp:23-52)
        self.lft_x1 = get_dev_nx(opts_init.nx, self.lft, self.size) * opts_init.dx if self.lft is not None else -1.
        if self.lft is not None and self.lft == self.size - 1:
            self.lft_x1 = opts_init.x1 - self.lft * get_dev_nx(opts_init.nx, 0, self.size) * opts_init.dx
        self.rgt_x0 = (opts_init.x0 if self.rgt == 0 else 0.) if self.rgt is not None else -1.
        self.bytes_moved = 0

    # ---- fan-outs (particles_multi_gpu_step.ipp:16-56, particles_multi_gpu_diag.ipp)
    def __getattr__(self, name):
        return getattr(self.prt, name)

    def init(self, *a, **kw):
        self.prt.init(*a, **kw)

    def sync_in(self, *a, **kw):
        self.prt.sync_in(*a, **kw)
        self._exchange_courant_halo()

    def step_sync(self, *a, **kw):
        self.prt.step_sync(*a, **kw)
        self._exchange_courant_halo()

    def _exchange_courant_halo(self):
        """pred_corr advection reads Courant numbers up to two x-planes outside the slab: every rank sends the planes next
        to its edges to the neighbours (particles_impl_xchng_courants.ipp:15-160), three small messages per side"""
        if self.size == 1:
            return
        torch, dist = self.torch, self.dist
        isz = self.real_t.itemsize
        for which in (0, 1, 2):
            cnt = self.prt.courant_halo_count(which)
            if not cnt:
                continue
            out_l, out_r, in_l, in_r = (self._buf(cnt * isz) for _ in range(4))
            ops = []
            if self.lft is not None:
                self.prt.courant_halo_pack(which, 0, out_l.data_ptr())
                ops.append(dist.P2POp(dist.isend, out_l, self.lft))
            if self.rgt is not None:
                self.prt.courant_halo_pack(which, 1, out_r.data_ptr())
                ops.append(dist.P2POp(dist.isend, out_r, self.rgt))
            if self.rgt is not None:
                ops.append(dist.P2POp(dist.irecv, in_r, self.rgt))
            if self.lft is not None:
                ops.append(dist.P2POp(dist.irecv, in_l, self.lft))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if self.on_gpu:
                torch.cuda.current_stream().synchronize()
            if self.lft is not None:
                self.prt.courant_halo_unpack(which, 0, in_l.data_ptr())
            if self.rgt is not None:
                self.prt.courant_halo_unpack(which, 1, in_r.data_ptr())

    def _buf(self, nbytes):
        t = self.torch.empty(max(int(nbytes), 8), dtype=self.torch.uint8, device=self.device if self.on_gpu else "cpu")
        return t

    def step_async(self, opts):
        """local step_async, then the neighbour exchange and post_copy
        (impl_multi_gpu/..._step_async_and_copy.ipp:28-206 without the thread barriers)"""
        torch, dist = self.torch, self.dist
        self.prt.step_async(opts)
        if self.size == 1:
            return
        n_lft, n_rgt = self.prt.migrate_counts()
        rec = self.prt.migrate_record_bytes()
        dev = self.device if self.on_gpu else "cpu"
        # 0) pack on the device (x re-based to the receiver's frame): needs only this rank's own counts, so it is issued
        #    before the count exchange
        out_l, out_r = self._buf(n_lft * rec), self._buf(n_rgt * rec)
        if self.lft is not None and n_lft:
            self.prt.migrate_pack(0, self.lft_x1, out_l.data_ptr(), out_l.numel())
        if self.rgt is not None and n_rgt:
            self.prt.migrate_pack(1, self.rgt_x0, out_r.data_ptr(), out_r.numel())
        # 1) counts.  send order (left, right); receive order (from right, from left): with two ranks both
        #    messages travel between the same pair and are matched in posting order.
        cnt_out = [torch.tensor([n_lft], dtype=torch.int64, device=dev), torch.tensor([n_rgt], dtype=torch.int64, device=dev)]
        cnt_in = [torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]   # [from right, from left]
        ops = []
        if self.lft is not None:
            ops.append(dist.P2POp(dist.isend, cnt_out[0], self.lft))
        if self.rgt is not None:
            ops.append(dist.P2POp(dist.isend, cnt_out[1], self.rgt))
        if self.rgt is not None:
            ops.append(dist.P2POp(dist.irecv, cnt_in[0], self.rgt))
        if self.lft is not None:
            ops.append(dist.P2POp(dist.irecv, cnt_in[1], self.lft))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        in_rgt, in_lft = int(cnt_in[0].item()), int(cnt_in[1].item())
        # 2) exchange the payloads
        buf_r, buf_l = self._buf(in_rgt * rec), self._buf(in_lft * rec)
        ops = []
        if self.lft is not None and n_lft:
            ops.append(dist.P2POp(dist.isend, out_l[:n_lft * rec], self.lft))
        if self.rgt is not None and n_rgt:
            ops.append(dist.P2POp(dist.isend, out_r[:n_rgt * rec], self.rgt))
        if self.rgt is not None and in_rgt:
            ops.append(dist.P2POp(dist.irecv, buf_r[:in_rgt * rec], self.rgt))
        if self.lft is not None and in_lft:
            ops.append(dist.P2POp(dist.irecv, buf_l[:in_lft * rec], self.lft))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if self.on_gpu:
                torch.cuda.current_stream().synchronize()
        self.bytes_moved += (n_lft + n_rgt) * rec
        # 3) append immigrants (left neighbour's first, as the reference unpacks lft then rgt), drop emigrants, re-index
        if in_lft:
            self.prt.migrate_unpack(buf_l.data_ptr(), in_lft)
        if in_rgt:
            self.prt.migrate_unpack(buf_r.data_ptr(), in_rgt)
        self.prt.migrate_finish(opts)

    def diag_puddle(self):
        """sum over ranks (particles_multi_gpu_diag.ipp:246-268)"""
        torch, dist = self.torch, self.dist
        loc = self.prt.diag_puddle()
        t = torch.tensor([loc[k] for k in lgrngn.output_names], dtype=torch.float64, device=self.device if self.on_gpu else "cpu")
        dist.all_reduce(t)
        return {k: float(v) for k, v in zip(lgrngn.output_names, t.cpu())}
