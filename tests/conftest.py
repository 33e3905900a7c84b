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
