import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "image-feature-extraction_amd"


# The library's default evaluates the solver's acos/cos with float polynomials
# (IFE_OPT_TRIG_MODE=2: inside north_star's 1e-5 bar, not bit-faithful).  Most tests here
# compare with the oracle far more tightly than that bar (sorted sample columns, histogram
# counts and tool outputs bit for bit), so the test session -- and every tool it spawns --
# starts contexts in mode 0, the double evaluation the oracle uses; the tests of the default
# mode select it explicitly (`ctx_fast`, tests/test_gpu_trig_default.py).
os.environ["IFE_TRIG_MODE"] = "0"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ife():
    """The product package (ctypes binding of csrc/libife_hip.so)."""
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synthetic")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle: the checker, never the thing under test on the GPU path."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.set_threads(min(8, os.cpu_count() or 1))
    return pyoracle


@pytest.fixture(scope="session")
def ctx(ife):
    """A device context; GPU tests fail (not skip) when the HIP library is missing."""
    c = ife.Context(0)
    yield c
    c.close()


@pytest.fixture()
def ctx_fast(ife, ctx):
    """The session context switched to the library's default trig mode for one test."""
    ctx.set_option(ife.OPT_TRIG_MODE, 2)
    yield ctx
    ctx.set_option(ife.OPT_TRIG_MODE, 0)
