import importlib
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
def fx():
    """The product package (directory name has a hyphen, so import it by name)."""
    return importlib.import_module("feature-extractor_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import fx_oracle
    fx_oracle.lib()
    return fx_oracle


@pytest.fixture(scope="session")
def gpu_fx(fx):
    """The package, after checking that the HIP library really is usable here; a -m gpu run on a
    box without the extension or without a gfx950 device must fail, not skip."""
    fx.load_library(build_if_missing=False)
    a = fx.BatchAnalyser(1, 1024)
    a.close()
    return fx
