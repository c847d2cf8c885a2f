import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def require_gpu():
    # -m gpu tests must FAIL (not skip) if the GPU or the HIP library is missing on a GPU box;
    # they are only deselected by marker on CPU-only runs.
    assert _has_gpu(), "a gfx950 GPU is required for -m gpu tests"
