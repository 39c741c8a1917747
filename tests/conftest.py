import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "image-feature-extraction_amd"


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
