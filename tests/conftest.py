import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Test-infrastructure convenience only: a checkout whose in-tree libfennec_hip.so was not built
    # yet (it is git-ignored) gets it built here with hipcc, exactly as __graft_entry__.build() does.
    # The product never builds or falls back by itself: fennec_amd.load_library() raises when the
    # library is missing.
    lib = os.path.join(ROOT, "fennec_amd", "libfennec_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "fennec_amd", "csrc")])


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/ -- test infrastructure; compiled on demand with gcc)."""
    from oracle import oracle
    oracle.build()
    return oracle
