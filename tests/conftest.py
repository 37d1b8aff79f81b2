import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """The built C-ABI library (compiled on demand; cross-compiles without a GPU)."""
    from gymwipe_amd import _native
    _native.build()
    return _native.lib()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ct_oracle
    ct_oracle.build()
    return ct_oracle.lib()
