"""Runs ONE of the reference's own Python test scripts, unmodified, from /root/reference (build container only) against the
CPU oracle through the same Python mirror the product uses: `libcloudphxx` resolves to the repo's drop-in package with
lgrngn.factory() redirected to the oracle library for every backend.  Test infrastructure (oracle = checker)."""
import os
import runpy
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
sys.path.insert(0, os.path.dirname(here))
import _harness as h                      # noqa: E402
import libcloudphxx                       # noqa: E402
from libcloudphxx_amd import lgrngn, common  # noqa: E402


def factory(backend, opts_init, real_t=None):
    return h.oracle_particles(opts_init)


lgrngn.factory = factory
script = sys.argv[1]
as_pytest = "--pytest" in sys.argv[2:]
sys.argv = [script] + [a for a in sys.argv[2:] if a != "--pytest"]
os.chdir(os.environ.get("LCX_REF_RUN_DIR", "/tmp"))
if as_pytest:                         # the reference runs this one with `python -m pytest` (tests/python/unit/CMakeLists.txt)
    import pytest
    sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", os.getcwd(), script]))
runpy.run_path(script, run_name="__main__")
