import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from libcloudphxx_amd import lgrngn
n, sd = 128, 512
oi = bench.make_opts_init(n, n, n, sd, 40., 1, 1, 44)
oi.strict_fp = False
dev = torch.device("cuda", 0)
tdtype = torch.float64
class TorchXP:
    @staticmethod
    def arange(m, dtype=None): return torch.arange(m, dtype=tdtype, device=dev)
    sin, cos, exp, log = staticmethod(torch.sin), staticmethod(torch.cos), staticmethod(torch.exp), staticmethod(torch.log)
f = bench.make_fields(n, n, n, 0, n, TorchXP, tdtype)
shapes = [(n, n, n)] * 3 + [(n + 1, n, n), (n, n + 1, n), (n, n, n + 1)]
ft = [t.expand(sh).contiguous() for t, sh in zip(f, shapes)]
th, rv, rhod, Cx, Cy, Cz = [lgrngn.DeviceArray(t.data_ptr(), t.shape) for t in ft]
torch.cuda.synchronize()
def say(*a):
    print(*a, flush=True)
pr = lgrngn.factory(lgrngn.backend_t.HIP, oi)
say("created")
pr.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz)
say("init done", pr.n_part)
opts = lgrngn.opts_t()
import ctypes
lib = pr._lib
pr.sync_in(th, rv, rhod, Cx, Cy, Cz); lib.lcx_dev_sync(); say("sync_in")
pr.step_cond(opts, th, rv); lib.lcx_dev_sync(); say("step_cond")
os.environ["AMD_LOG_LEVEL"] = "0"
pr.set_profiling(True)
try:
    pr.step_async(opts); lib.lcx_dev_sync(); say("step_async", pr.n_part)
except Exception as e:
    say("FAILED", e)
say("ok")
