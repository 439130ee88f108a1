import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Build libadmm_hip.so and the oracle once per session (no-op when fresh)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def lib(built):
    import admm_library_amd as pkg
    return pkg.load_library()


@pytest.fixture(scope="session")
def gpu(lib):
    """GPU tests must run on the HIP path: fail loudly, never skip, if it is absent."""
    import admm_library_amd as pkg
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X box")
    return True
