import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "hands-on-point-cloud-processing_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def pcr():
    """The product: Python mirror over the C-ABI of libpcr_hip.so (fails loudly if not built)."""
    return importlib.import_module(PKG)


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


@pytest.fixture(scope="session")
def golden():
    return load_golden
