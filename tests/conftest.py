import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import sdm_pkg
    mod = sdm_pkg.load()
    # build the HIP libraries if the tree has none yet (no-op when lib/*.so is newer than the sources);
    # a missing library is never papered over: load_library() raises and every test fails loudly
    if not os.path.exists(mod.lib_path()):
        mod.build_mod.build_all(verbose=True)
    return mod


@pytest.fixture(scope="session")
def oracle():
    from pm_oracle import Oracle
    return Oracle("strict")


@pytest.fixture(scope="session")
def gpu_ok(pkg):
    lib = pkg.load_library()
    if lib.sdm_device_count() < 1:
        pytest.fail("-m gpu tests need a visible HIP device (no CPU fallback exists)")
    return True
