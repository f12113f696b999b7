"""one C2 run (bench.c2_leg) for a kernel trace:  rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/c2_trace.py [steps] [cond_solver] [dbg,flags]"""
import os
import sys
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from libcloudphxx_amd import lgrngn
kw = {"cond_solver": int(sys.argv[2])} if len(sys.argv) > 2 else {}
if len(sys.argv) > 3:
    kw["dbg_flags"] = sum(int(getattr(lgrngn.dbg, f)) for f in sys.argv[3].split(","))
r = bench.c2_leg(lgrngn, torch, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, **kw)
print(json.dumps(r))
