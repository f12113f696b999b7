"""libcloudphxx_amd -- MI355X-native super-droplet (lgrngn) microphysics backend.

Only the Lagrangian hot path of libcloudph++ lives here: HIP kernels + C ABI in csrc/
(include/lcx.h), and host-side mirrors of the reference interface (lgrngn.py for Python callers,
include/libcloudphxx_amd/ for C++ callers).
"""
from . import lgrngn  # noqa: F401
