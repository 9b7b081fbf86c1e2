import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import _pkg
    return _pkg.load()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def binding(pkg):
    from vitcpp_amd import binding as B
    if not os.path.exists(B.LIB_PATH):
        B.build()
    return B


@pytest.fixture(scope="session")
def torch_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible: the HIP path has no CPU fallback")
    return torch
