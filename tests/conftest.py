import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # sweeps (scripts/gpu_cross_checks.sh): MMX_TEST_ROUTE=prefer_wide | fused | wide | explicit_jacobian sends every problem the
    # tests create through that route (a test-side variable: the library itself reads no environment on the solve path)
    route = os.environ.get("MMX_TEST_ROUTE")
    if route:
        from momentum_amd import capi

        capi.default_route = route


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/mmx_oracle.hpp)."""
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch
