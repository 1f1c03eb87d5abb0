import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """-m gpu tests need a real device: on a CPU-only box a plain `pytest` skips them instead of failing"""
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device here)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the -m gpu tests; fails loudly when the HIP path is unavailable."""
    from strling_amd import api
    c = api.Context(0)
    yield c
    c.close()
