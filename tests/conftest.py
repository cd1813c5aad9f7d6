import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import _gpis_pkg
    return _gpis_pkg.load_package()


@pytest.fixture(scope="session")
def ob():
    import oracle_bindings
    oracle_bindings.build()
    return oracle_bindings


@pytest.fixture(autouse=True)
def _clean_kernel_selection_env():
    """The library reads its kernel-selection overrides from the environment at call time; a test that sets
    one must not leak it into the next."""
    keys = ("GPIS_MARCH", "GPIS_WAVE_TAIL", "GPIS_PATHS_SORT", "GPIS_PATHS_PRESORT")
    yield
    for k in keys:
        os.environ.pop(k, None)
