import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    # /dev/kfd is what a ROCm process needs; asking torch would initialise the runtime for nothing
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (the GPU tier runs under gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
