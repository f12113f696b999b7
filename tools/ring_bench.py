"""Emulates the N-rank decomposed run of bench.py on ONE GPU: N slab objects stepped one after the other in one process,
migrants moved with the same pack / unpack / finish primitives as libcloudphxx_amd/multi.py (device buffers, no RCCL).
The mean time per slab per step is what one rank of an N-GPU run spends outside the transport itself.
    python tools/ring_bench.py --slabs 8 [--steps 10 --warmup 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from libcloudphxx_amd import lgrngn, multi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slabs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--sd-conc", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, size = args.n, args.slabs
    oi_g = bench.make_opts_init(n, n, n, args.sd_conc, 40., 1, 1, 44)
    oi_g.strict_fp = False

    class TorchXP:
        @staticmethod
        def arange(m, dtype=None):
            return torch.arange(m, dtype=torch.float64, device=dev)
        sin, cos, exp, log = staticmethod(torch.sin), staticmethod(torch.cos), staticmethod(torch.exp), staticmethod(torch.log)

    prts, fields, ois = [], [], []
    for r in range(size):
        oi, bfr = multi.distmem_opts(oi_g, r, size)
        oi.n_x_bfr = 0
        oi.rng_seed = 44 + r
        p = lgrngn.factory(lgrngn.backend_t.HIP, oi, np.float64)
        nx = oi.nx
        f = bench.make_fields(nx, n, n, bfr, n, TorchXP, torch.float64)
        shapes = [(nx, n, n)] * 3 + [(nx + 1, n, n), (nx, n + 1, n), (nx, n, n + 1)]
        f = [t.expand(s).contiguous() for t, s in zip(f, shapes)]
        arrs = [lgrngn.DeviceArray(t.data_ptr(), t.shape) for t in f]
        p.init(arrs[0], arrs[1], arrs[2], Cx=arrs[3], Cy=arrs[4], Cz=arrs[5])
        prts.append(p); fields.append((f, arrs)); ois.append(oi)
    opts = lgrngn.opts_t()
    rec = prts[0].migrate_record_bytes()
    moved = 0

    def step():
        nonlocal moved
        for p, (_, a) in zip(prts, fields):
            p.step_sync(opts, *a)
            p.step_async(opts)
        packs = []
        for r, p in enumerate(prts):
            nl, nr = p.migrate_counts()
            lft, rgt = (r - 1) % size, (r + 1) % size
            bl = torch.empty(max(nl * rec, 8), dtype=torch.uint8, device=dev)
            br = torch.empty(max(nr * rec, 8), dtype=torch.uint8, device=dev)
            if nl: p.migrate_pack(0, ois[lft].x1, bl.data_ptr(), bl.numel())
            if nr: p.migrate_pack(1, ois[rgt].x0, br.data_ptr(), br.numel())
            packs.append((nl, bl, nr, br))
            moved += nl + nr
        for r, p in enumerate(prts):
            lft, rgt = (r - 1) % size, (r + 1) % size
            if packs[lft][2]: p.migrate_unpack(packs[lft][3].data_ptr(), packs[lft][2])
            if packs[rgt][0]: p.migrate_unpack(packs[rgt][1].data_ptr(), packs[rgt][0])
        for p in prts:
            p.migrate_finish(opts)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    moved = 0
    for p in prts:
        p.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = prts[0].timings()
    print(json.dumps({"slabs": size, "ms_per_step_all_slabs": dt / args.steps * 1e3, "ms_per_slab": dt / args.steps / size * 1e3,
                      "super_droplets": int(sum(p.n_part for p in prts)), "migrants_per_step": moved / args.steps,
                      "bytes_per_direction_per_rank": moved / args.steps / size / 2 * rec,
                      "slab0_stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in tm.items()}}))


if __name__ == "__main__":
    main()
