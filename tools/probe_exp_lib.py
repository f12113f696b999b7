"""exp_lib (the device library's exp with its coefficients in constant memory, lcx_math.hpp) against the library's own exp: the same bits
(math probe 9 against 3).   python3 tools/probe_exp_lib.py   (needs the GPU)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
from libcloudphxx_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(1)
x = np.concatenate([rng.uniform(-745, 710, 4000000), rng.uniform(-2, 2, 4000000), 10. ** rng.uniform(-300, 2.9, 1000000),
                    [0., -0., 1., -1., 709.78, 709.79, 1024., 1025., -1074., -1075., -1076., np.inf, -np.inf, np.nan, 1e-320]])
def probe(which):
    y = np.empty_like(x)
    rc = lib.lcx_math_probe(C.c_int(which), x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(x.size))
    assert rc == 0
    return y
a, b = probe(3), probe(9)
same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
print("exp_lib vs exp: %d arguments, %d differ" % (x.size, int((~same).sum())))
if not same.all():
    i = np.nonzero(~same)[0][:5]
    print(x[i], a[i], b[i])
    sys.exit(1)
