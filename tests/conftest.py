import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import seesaw_oracle
    seesaw_oracle.c_lib()
    return seesaw_oracle


@pytest.fixture()
def lab_build():
    """tests that flip kernel variants (ssw_tune_*) or drive single kernels (ssw_debug_*) run on libseesaw_hip_debug.so --
    the same sources compiled with -DSSW_DEBUG_HOOKS (include/seesaw_hip_debug.h); everything else in the suite runs on the
    product library, which has no such switch.  Handles are created and closed inside the test."""
    from seesaw_amd import _lib
    with _lib.debug_hooks() as lib:
        yield lib


def free_port() -> int:
    """a TCP port the OS reports free right now (bind to port 0 on 127.0.0.1): rendezvous ports derived from the pid could
    collide between concurrent runs of the suite"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])
