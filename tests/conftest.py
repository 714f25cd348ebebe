import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ultrasonic-communication_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if they are collected on a box without
    # a GPU (plain `pytest tests/`), skip instead of failing on uc_create.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def uc_tuning(monkeypatch):
    """The library reads its experiment switches (UC_GRID, UC_*_GROUP, UC_STATIC_DEAL, ...) only under UC_TUNING=1."""
    monkeypatch.setenv("UC_TUNING", "1")
