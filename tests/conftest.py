import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "vision-transformers-pytorch_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / round-end driver)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly on a GPU box when the HIP extension is missing,
    # and must be deselected (by -m "not gpu") on CPU; never silently skipped.
    pass
