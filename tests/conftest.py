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
    """The CPU oracle (test infrastructure). Builds oracle/libocn_oracle.so on first use."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ocn():
    """The product package on the MI355X; fails loudly when the HIP extension is missing."""
    import oldoceananigans_jl_amd as ocn
    return ocn


@pytest.fixture(scope="session")
def arch(ocn):
    return ocn.GPU(0)
