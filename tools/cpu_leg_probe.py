"""Stage timers of the CPU oracle's OpenMP build on the bench workload (what bench.py's cpu_baseline leg times), to see which stages
still run on one thread:   ORC_TIMERS=1 [OMP_NUM_THREADS=k] python3 tools/cpu_leg_probe.py <cells per dimension> <steps>"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, _harness as h
from libcloudphxx_amd import lgrngn
n = int(sys.argv[1]); steps = int(sys.argv[2])
oi = bench.make_opts_init(n, n, n, 64, 40., 1, 1, 44)
th, rv, rhod, Cx, Cy, Cz = bench.make_fields(n, n, n, 0, n, np, np.float64)
pr = h.oracle_omp_particles(oi)
lib = h.oracle_omp_lib()
ti=time.perf_counter(); pr.init(th, rv, rhod, Cx=Cx, Cy=Cy, Cz=Cz); print("init s", time.perf_counter()-ti)
opts = lgrngn.opts_t()
pr.step_sync(opts, th, rv, rhod, Cx, Cy, Cz); pr.step_async(opts)
lib.orc_timers_dump()
t0 = time.perf_counter(); done = 0
for _ in range(steps):
    pr.step_sync(opts, th, rv, rhod, Cx, Cy, Cz); pr.step_async(opts); done += pr.n_part
dt = time.perf_counter() - t0
print("threads", lib.orc_num_threads(), "SD/s %.3e" % (done / dt), "s/step %.3f" % (dt / steps))
lib.orc_timers_dump()
