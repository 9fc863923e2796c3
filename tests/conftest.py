import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One yakhip context on cuda:0 bound to torch's current stream.  Fails loudly when the HIP
    library is missing or no GPU is visible -- there is no fallback path to test."""
    import torch

    from yak_amd import capi

    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    capi.load()
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
