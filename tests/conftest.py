import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["checker", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """The host-logic tests run twice: on the CPU checker backend (no GPU needed) and, on the GPU
    box, on the shipped HIP backend."""
    if request.param == "checker":
        from checker_backend import CheckerBackend
        return CheckerBackend()
    from image_classification_amd.backend import default_backend
    return default_backend()
