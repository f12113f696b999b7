"""1-D domain decomposition across the GPUs of one node, ONE PROCESS PER GPU: neighbour exchange of migrating super-droplets with
torch.distributed point-to-point operations (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests and for several
ranks on one GPU).

The reference's multi_CUDA object -- ONE process, all GPUs of the node -- is native in the C library (lcx_create_multi,
csrc/lcx_multi.hpp) and is what factory(multi_CUDA | multi_HIP) returns.  This module is its SPMD twin (e.g. under
torch.distributed.run, or next to an MPI host model) and replaces the reference's MPI flavour
(src/impl/distributed_memory/particles_impl_mpi_exchange.ipp:20-330) as well as the per-step choreography of
src/impl_multi_gpu/particles_multi_gpu_impl_step_async_and_copy.ipp:28-206 (one std::thread per GPU, cudaMemcpyPeerAsync of two
packed buffers, five thread barriers).  There is no collective on the data path: each rank talks to its left and right neighbour
only (periodic ring, or open ends with open_side_walls).

Protocol of a step (include/lcx.h, lcx_exch_*; the engine keeps the emigrant lists and their counts on the device):

  step_async                      coalescence ... advection, boundary, re-index of those that stay           (queued)
  exch_pack                       emigrants -> two outboxes, header = {count, overflow, first-part size of the NEXT message}  (queued)
  exch_sort_interior              (queued once the batch below is under way) scan / scatter / rank of the cells no immigrant can reach
  ONE batch of isend / irecv      the header and the first `cap` records of each message, `cap` agreed one step ahead through the
                                  header (the sender sizes it from its previous counts), so that no count travels ahead of the
                                  payload and the host never waits for one; ordered against the engine's stream, not the host
  exch_unpack                     both inboxes -> storage, histogram                                                     (queued)
  exch_finish                     the step's ONE host synchronisation (32 bytes of counts), then scan / scatter / rank
  (rarely) a second batch         only when a message held more records than its first part: the remaining tiles, unpack, finish

Slab sizes follow detail::get_dev_nx / distmem_opts (src/detail/distmem_opts.hpp:10-52).
"""
import copy

import numpy as np

from . import lgrngn

BCOND_SHAREDMEM, BCOND_DISTMEM, BCOND_OPEN = 0, 1, 3      # src/detail/bcond.hpp
EXCH_HDR, EXCH_TILE = 256, 256                            # csrc/lcx_kernels.hpp


def get_dev_nx(nx, rank, size):
    """distmem_opts.hpp:10-16.  `opts_init.nx / size + .5` is an INTEGER division there (both operands are int), so every
    rank but the last gets floor(nx / size) planes and the last one the remainder."""
    per = nx // size
    n = per if rank < size - 1 else nx - rank * per
    if n <= 0:
        raise RuntimeError("libcloudph++: number of devices exceeds nx")
    return n


def distmem_opts(opts_init, rank, size, self_ring=False):
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
    if size > 1 or self_ring:
        if not opts_init.open_side_walls:
            oi.bcond_lft = oi.bcond_rgt = BCOND_DISTMEM
        else:
            oi.bcond_lft = BCOND_OPEN if rank == 0 else BCOND_DISTMEM
            oi.bcond_rgt = BCOND_OPEN if rank == size - 1 else BCOND_DISTMEM
    return oi, n_x_bfr


class _DevMem:
    """a device allocation owned by the engine, seen by torch through the CUDA array interface"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _tiles(n_rec):
    return (int(n_rec) + EXCH_TILE - 1) // EXCH_TILE


class particles_multi_t:
    """SPMD flavour of particles_t<real_t, multi_CUDA>: every rank constructs it with the GLOBAL opts_init;
    arrays passed to init/step_sync are the rank's LOCAL slabs (x-planes [n_x_bfr, n_x_bfr + nx_local)) unless
    global_arrays=True, in which case the library indexes the global arrays with the n_x_bfr offset exactly like
    the reference does (initialization/particles_impl_init_e2l.ipp:44-46).

    make_particles(opts_init_local) -> particles object (defaults to the HIP backend).
    transport: "rccl"  -- device buffers handed to torch.distributed as they are (backend nccl = RCCL over xGMI), ordered against the
                          engine's stream: no host synchronisation of its own; needs one GPU per rank;
               "host"  -- the used part of each message staged through host memory (gloo): several ranks on one GPU, CPU engines;
               None    -- "rccl" when the engine runs on a GPU and the process group has a device backend, else "host".
    """

    def __init__(self, opts_init, real_t=np.float64, make_particles=None, device=None, global_arrays=False, transport=None, self_ring=False):
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
        # self_ring: ONE rank whose left and right neighbour is the rank itself (periodic side walls): every message of the protocol
        # travels through the transport to the sender's own inbox.  What a one-GPU box can run of the RCCL path (RCCL refuses two
        # ranks on one device): tests/test_hip_spmd.py drives it; a production run has no use for it.
        self.self_ring = bool(self_ring) and self.size == 1
        if self.self_ring and opts_init.open_side_walls:
            raise RuntimeError("libcloudph++: a ring of one rank needs periodic side walls")
        oi, self.n_x_bfr = distmem_opts(opts_init, self.rank, self.size, self.self_ring)
        if not global_arrays:
            oi.n_x_bfr = 0
            oi.n_x_tot = oi.nx        # slab-local arrays: a Courant halo wraps inside the slab, the exchange below overwrites it
        self.opts_init = oi
        if device is None and make_particles is None:
            # the HIP engine works on device buffers: with the default engine the rank's current GPU is the device
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
        if self.size == 1 and not self.self_ring:
            self.lft = self.rgt = None
        self.exchanging = self.size > 1 or self.self_ring
        # neighbours' domain edges in THEIR local frames (xchng_domains.ipp:23-52)
        self.lft_x1 = get_dev_nx(opts_init.nx, self.lft, self.size) * opts_init.dx if self.lft is not None else -1.
        if self.lft is not None and self.lft == self.size - 1:
            self.lft_x1 = opts_init.x1 - self.lft * get_dev_nx(opts_init.nx, 0, self.size) * opts_init.dx
        self.rgt_x0 = (opts_init.x0 if self.rgt == 0 else 0.) if self.rgt is not None else -1.
        self.bytes_moved = 0
        self.second_rounds = 0
        self.host_syncs = 0            # host synchronisations this object caused beyond the engine's one per step
        if transport is None:
            has_dev_backend = False
            try:
                has_dev_backend = "nccl" in str(dist.get_backend()).lower()
            except Exception:
                pass
            transport = "rccl" if (self.on_gpu and has_dev_backend) else "host"
        if transport not in ("rccl", "host"):
            raise RuntimeError("libcloudph++: transport must be 'rccl' or 'host'")
        if transport == "rccl" and not self.on_gpu:
            raise RuntimeError("libcloudph++: the rccl transport needs the engine's buffers on a GPU")
        self.transport = transport
        self._halo_bufs = {}
        if self.exchanging:
            self._setup_exchange()

    # ---- message buffers: owned by the engine, wrapped once
    def _setup_exchange(self):
        torch = self.torch
        p = self.prt
        self.cap_rec = p.exch_enable(get_dev_nx(self.glob_opts_init.nx, 0, self.size))       # one capacity for all ranks
        self.tile_bytes = (p.exch_message_bytes(EXCH_TILE) - EXCH_HDR)
        nbytes = p.exch_message_bytes(self.cap_rec)
        ptrs = p.exch_buffers()
        if self.on_gpu:
            self.box = [torch.as_tensor(_DevMem(q, nbytes), device=self.device) for q in ptrs]
            if self.transport == "host":
                self.stage = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
            s = p.stream()
            self.ext_stream = torch.cuda.ExternalStream(s, device=self.device) if s else None
        else:
            import ctypes
            self.box = [torch.from_numpy(np.ctypeslib.as_array((ctypes.c_ubyte * nbytes).from_address(q))) for q in ptrs]
            self.stage = self.box
            self.ext_stream = None
        # first-part capacities in records (multiples of a tile): [to/from the left, to/from the right].  The start value is the same
        # on every rank by construction; from then on the sender announces the next one in its header
        cap0 = min(self.cap_rec, max(EXCH_TILE, _tiles(self.cap_rec // 8) * EXCH_TILE))
        self.send_cap = [cap0, cap0]
        self.recv_cap = [cap0, cap0]
        self.last_out = [0, 0]

    def _next_cap(self, side):
        """first-part size of the NEXT message to `side`: a quarter above the last count, at least one tile, at most the inbox"""
        want = _tiles(self.last_out[side] * 5 // 4 + 1) * EXCH_TILE
        return int(min(self.cap_rec, max(EXCH_TILE, want)))

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

    # ---- transports: one batch of point-to-point operations; segs = [(tensor view to send | None, peer), ...] in the fixed order
    #      send left, send right, receive from right, receive from left (with two ranks both messages travel between the same pair and
    #      are matched in posting order)
    def _p2p(self, send_l, send_r, recv_r, recv_l, between=None):
        """between: called after the operations have been STARTED and before they are waited for (work that overlaps the transfer)"""
        dist = self.dist
        if self.self_ring and self.transport == "host":
            # (gloo has no connection from a rank to itself: what goes out to the left comes in from the right)
            if between is not None:
                between()
            if send_l is not None:
                recv_r.copy_(send_l)
            if send_r is not None:
                recv_l.copy_(send_r)
            return
        ops = []
        if send_l is not None:
            ops.append(dist.P2POp(dist.isend, send_l, self.lft))
        if send_r is not None:
            ops.append(dist.P2POp(dist.isend, send_r, self.rgt))
        if recv_r is not None:
            ops.append(dist.P2POp(dist.irecv, recv_r, self.rgt))
        if recv_l is not None:
            ops.append(dist.P2POp(dist.irecv, recv_l, self.lft))
        works = dist.batch_isend_irecv(ops) if ops else []
        if between is not None:
            between()
        for w in works:
            w.wait()                       # (device tensors: the CURRENT STREAM waits, not the host)

    def _ship(self, ranges, between=None):
        """ranges: [(lo, hi) bytes of outbox-left to send | None, outbox-right | None, inbox-from-right to fill | None, inbox-from-left | None]
        between: engine work that does not depend on the messages, queued once they are under way"""
        torch = self.torch
        out_l, out_r, in_l, in_r = self.box
        for t, r in zip((out_l, out_r, in_r, in_l), ranges):
            assert r is None or 0 <= r[0] <= r[1] <= t.numel(), (r, t.numel())      # (a slice beyond the box would be clamped silently)
        views = lambda t, r: None if r is None else t[r[0]:r[1]]
        if self.transport == "rccl":
            if self.ext_stream is not None:
                with torch.cuda.stream(self.ext_stream):       # torch's "current stream" = the engine's: RCCL orders itself against it
                    self._p2p(views(out_l, ranges[0]), views(out_r, ranges[1]), views(in_r, ranges[2]), views(in_l, ranges[3]), between)
            else:
                self._p2p(views(out_l, ranges[0]), views(out_r, ranges[1]), views(in_r, ranges[2]), views(in_l, ranges[3]), between)
            return
        if between is not None:
            between()                      # (host-staged: nothing overlaps, the work is simply done first)
        # host-staged: the engine's queue drains, the used bytes cross the host
        st = self.stage
        if self.on_gpu:
            if self.ext_stream is not None:
                self.ext_stream.synchronize()
            self.host_syncs += 1
            for k in (0, 1):
                if ranges[k] is not None:
                    st[k][ranges[k][0]:ranges[k][1]].copy_(self.box[k][ranges[k][0]:ranges[k][1]])
            torch.cuda.synchronize(self.device)
        self._p2p(views(st[0], ranges[0]), views(st[1], ranges[1]), views(st[3], ranges[2]), views(st[2], ranges[3]))
        if self.on_gpu:
            for k, r in ((3, ranges[2]), (2, ranges[3])):
                if r is not None:
                    self.box[k][r[0]:r[1]].copy_(st[k][r[0]:r[1]])
            torch.cuda.synchronize(self.device)

    def _msg_range(self, rec_lo, rec_hi, with_header):
        """bytes of a message that hold the tiles of records [rec_lo, rec_hi) (rec_lo a multiple of the tile), optionally from the header on"""
        lo = 0 if with_header else EXCH_HDR + (rec_lo // EXCH_TILE) * self.tile_bytes
        return (lo, EXCH_HDR + _tiles(rec_hi) * self.tile_bytes)

    def step_async(self, opts):
        """local step_async, then the neighbour exchange and post_copy
        (impl_multi_gpu/..._step_async_and_copy.ipp:28-206 without the thread barriers)"""
        p = self.prt
        p.step_async(opts)
        if not self.exchanging:
            return
        hl, hr = self.lft is not None, self.rgt is not None
        nxt = [self._next_cap(0), self._next_cap(1)]
        p.exch_pack(hl, self.lft_x1, hr, self.rgt_x0, nxt[0], nxt[1])
        sc, rc = self.send_cap, self.recv_cap
        # the re-sort of the slab's interior needs nothing from the neighbours: queued while the messages travel
        self._ship([self._msg_range(0, sc[0], True) if hl else None, self._msg_range(0, sc[1], True) if hr else None,
                    self._msg_range(0, rc[1], True) if hr else None, self._msg_range(0, rc[0], True) if hl else None], between=p.exch_sort_interior)
        p.exch_unpack(hl, hr, rc[0], rc[1])
        done, rec = p.exch_finish(opts)
        out, inc = [rec[1], rec[2]], [rec[3], rec[4]]
        # a message that outgrew the RECEIVER's inbox: the receiver raises in its exch_finish (flag 1 of its record); the sender must not
        # go on to a second part that no receive will ever match (it would wait for the process group's time-out instead of saying why)
        for k, has in ((0, hl), (1, hr)):
            if has and out[k] > self.cap_rec:
                raise RuntimeError("libcloudph++: more super-droplets crossed a slab face in one step than the exchange buffer holds (%d records); "
                                   "raise opts_init.n_sd_max" % self.cap_rec)
        # the rare second part: a message that held more than its agreed first part (both ends see it in their own record)
        more_out = [hl and out[0] > sc[0], hr and out[1] > sc[1]]
        more_in = [hl and inc[0] > rc[0], hr and inc[1] > rc[1]]
        if any(more_out) or any(more_in):
            self.second_rounds += 1
            self._ship([self._msg_range(sc[0], out[0], False) if more_out[0] else None, self._msg_range(sc[1], out[1], False) if more_out[1] else None,
                        self._msg_range(rc[1], inc[1], False) if more_in[1] else None, self._msg_range(rc[0], inc[0], False) if more_in[0] else None])
        if not done:
            p.exch_unpack(hl, hr)
            done, rec = p.exch_finish(opts)
            if not done:
                raise RuntimeError("libcloudph++: neighbour exchange incomplete after its second part")
        rec_bytes = self.tile_bytes // EXCH_TILE
        self.bytes_moved += (out[0] + out[1]) * rec_bytes
        self.last_out = out
        self.send_cap = nxt
        self.recv_cap = [rec[9] if hl else rc[0], rec[10] if hr else rc[1]]

    # ---- Courant halo of pred_corr
    def _halo_buf(self, key, nbytes, where):
        t = self._halo_bufs.get(key)
        if t is None or t.numel() < nbytes:
            t = self.torch.empty(max(int(nbytes), 8), dtype=self.torch.uint8, device=where)
            self._halo_bufs[key] = t
        return t

    def _exchange_courant_halo(self):
        """pred_corr advection reads Courant numbers up to two x-planes outside the slab: every rank sends the planes next
        to its edges to the neighbours (particles_impl_xchng_courants.ipp:15-160), three small messages per side"""
        if not self.exchanging:
            return
        torch = self.torch
        isz = self.real_t.itemsize
        staged = self.on_gpu and self.transport == "host"
        hl, hr = self.lft is not None, self.rgt is not None
        for which in (0, 1, 2):
            cnt = self.prt.courant_halo_count(which)
            if not cnt:
                continue
            nb = cnt * isz
            # what the engine packs into / unpacks from (its own memory space), and what torch.distributed ships: out_l, out_r, in_l, in_r
            eng = [self._halo_buf((which, "e", k), nb, self.device if self.on_gpu else "cpu") for k in range(4)]
            wire = eng if not staged else [self._halo_buf((which, "w", k), nb, "cpu") for k in range(4)]
            if hl:
                self.prt.courant_halo_pack(which, 0, eng[0].data_ptr())        # (synchronous: the planes are in the buffer on return)
            if hr:
                self.prt.courant_halo_pack(which, 1, eng[1].data_ptr())
            if staged:
                wire[0].copy_(eng[0]); wire[1].copy_(eng[1])
                torch.cuda.synchronize(self.device)
            self._p2p(wire[0][:nb] if hl else None, wire[1][:nb] if hr else None, wire[3][:nb] if hr else None, wire[2][:nb] if hl else None)
            if self.on_gpu and not staged:
                torch.cuda.current_stream().synchronize()                      # the engine unpacks on its own stream
            if staged:
                eng[2].copy_(wire[2]); eng[3].copy_(wire[3])
                torch.cuda.synchronize(self.device)
            if hl:
                self.prt.courant_halo_unpack(which, 0, eng[2].data_ptr())
            if hr:
                self.prt.courant_halo_unpack(which, 1, eng[3].data_ptr())

    def diag_puddle(self):
        """sum over ranks (particles_multi_gpu_diag.ipp:246-268)"""
        torch, dist = self.torch, self.dist
        loc = self.prt.diag_puddle()
        dev_ = self.device if (self.on_gpu and self.transport == "rccl") else "cpu"
        t = torch.tensor([loc[k] for k in lgrngn.output_names], dtype=torch.float64, device=dev_)
        dist.all_reduce(t)
        return {k: float(v) for k, v in zip(lgrngn.output_names, t.cpu())}
