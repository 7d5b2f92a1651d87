import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """A zipc_hip context on cuda:0.  Fails (does not skip) when the HIP library
    or the GPU is missing: GPU tests must never pass on a fallback."""
    import zipc_amd

    return zipc_amd.default_context(0)
