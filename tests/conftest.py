import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the HIP library and the oracle's C twin exist (both build without a GPU)."""
    lib = os.path.join(ROOT, "lotus_amd", "liblotus_hip.so")
    olib = os.path.join(ROOT, "oracle", "liblvs_oracle.so")
    tlib = os.path.join(ROOT, "oracle", "liblvs_blas_twin.so")
    if not (os.path.exists(lib) and os.path.exists(olib) and os.path.exists(tlib)):
        import __graft_entry__ as g

        g.build()
    yield


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip_backend():
    if not has_gpu():
        pytest.skip("no GPU")
    from lotus_amd.backend import HipBackend

    return HipBackend("cuda:0")
