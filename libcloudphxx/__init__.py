"""Drop-in name of the reference's Python module: `from libcloudphxx import lgrngn, common` resolves to the MI355X backend
(ref: bindings/python/lib.cpp:40-52,212-214).  `lgrngn.factory(lgrngn.backend_t.CUDA | HIP | multi_CUDA | multi_HIP, opts_init)`
returns the HIP implementation; the CPU backends (serial, OpenMP) are not compiled in and raise, as in a reference build
without them (ref: src/lib.cpp:21-38).  bulk schemes (blk_1m, blk_2m) are outside this library."""
from libcloudphxx_amd import common, lgrngn  # noqa: F401

git_revision = "libcloudphxx_amd"
