import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def hm(pkg):
    return pkg.lib()


@pytest.fixture(scope="session")
def hm_hooks(pkg):
    """libheif_mi355x_test.so: the shipping library's objects + csrc/test_hooks.cpp (hm_debug_set, hm_debug_kernel_regs).  A library
    of its own beside `hm` in this process (own knobs, own device pool): what is set here does not reach `hm`."""
    import ctypes
    pkg.lib()  # (torch's HIP runtime first)
    L = ctypes.CDLL(pkg.capi.TEST_LIB_PATH)
    L.hm_status_string.restype = ctypes.c_char_p
    L.hm_last_error.restype = ctypes.c_char_p
    L.hm_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.hm_debug_set.restype = ctypes.c_int
    pkg.capi.bind_decode(L)
    return L


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.load()
