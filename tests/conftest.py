import os
import sys

import pytest

# torch bundles its own ROCm user-space (libamdhip64.so.7, hipfft, rccl ...) under the same sonames as /opt/rocm. Whoever is
# loaded first wins for the whole process, and torch cannot initialise its GPU runtime on top of the system one -- so in
# any process that uses torch on the GPU (the distributed path), torch must be imported BEFORE libocn_mi355x.so is loaded.
if os.environ.get("OCN_TEST_NO_TORCH") != "1":
    import torch  # noqa: F401,E402

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
