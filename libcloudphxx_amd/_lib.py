"""Locates and loads the product shared library (HIP kernels + C ABI).

The library is built in-tree by __graft_entry__.build() / libcloudphxx_amd/build.py.  There is
NO fallback: if the HIP library is missing or cannot be loaded, importing the backend fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblcx_hip.so")
_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libcloudphxx_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        os.environ.setdefault("LCX_DATA_DIR", os.path.join(_HERE, "data"))
        _cached = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    return _cached
