import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SAMPLES = os.path.join(GOLDEN, "samples")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def wm():
    """The product package (ctypes mirror of the C ABI); builds the native code if missing."""
    mod = importlib.import_module("rtl-wmbus_amd")
    if not os.path.exists(mod.LIB_PATH) or not os.path.exists(mod.SYNTH_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def samples():
    import numpy as np
    return {
        "samples2": np.fromfile(os.path.join(SAMPLES, "rtlsdr_868.950M_1M6_samples2.cu8"), np.uint8),
        "issue48": np.fromfile(os.path.join(SAMPLES, "rtlsdr_868.625M_2M4_issue48.cu8"), np.uint8),
    }


@pytest.fixture(scope="session")
def libm_is_glibc_235():
    """True when this host's atan2f gives glibc 2.35's answers (tests/golden/atan2f_kat.bin, made by make_atan2f_kat.py).
    The exact path restates that generation; the oracle and the reference call the host's libm.  Where they differ the
    REFERENCE prints other soft symbols, and comparisons against the host's libm / the oracle's taps say nothing about
    the kernels: such tests skip with this reason instead of failing."""
    import ctypes
    import numpy as np
    kat = np.fromfile(os.path.join(GOLDEN, "atan2f_kat.bin"), "<u4").reshape(-1, 3)
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    y, x = kat[:, 0].copy().view(np.float32), kat[:, 1].copy().view(np.float32)
    got = np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y[::8], x[::8])], np.float32).view(np.uint32)
    return bool(np.array_equal(got, kat[::8, 2]))
