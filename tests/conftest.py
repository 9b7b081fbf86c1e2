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


def boundary_rows(binding, path, n, dtype=0, **options):
    """Image ids an n-image forward of this model is checked on: both ends of the batch and the images on either side of every
    sub-batch stream boundary of the context the engine actually builds (vitx_ctx_split) -- r03 verdict: the hard-coded 109 / 110
    had gone stale when the split moved to 103 + 153."""
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=n, dtype=dtype, **options)
    parts, ids = ctx.split(n), ctx.boundary_rows(n)
    ctx.close(); model.close()
    assert sum(parts) == n and all(p > 0 for p in parts)
    off = 0
    for sz in parts[:-1]:
        off += sz
        assert off - 1 in ids and off in ids
    return ids
