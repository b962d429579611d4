import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "kmer-sets-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip: the
    # product path has no CPU fallback.  Without -m gpu the gpu tests are simply
    # deselected by the marker expression the driver passes.
    pass


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.fail("this test needs a GPU (torch.cuda.is_available() is False)")
    import torch

    return torch.device("cuda:0")
