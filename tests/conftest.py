import os
import sys

import pytest

os.environ.setdefault("P3V_GEMM_BIG_ROWS", "512")   # M >= 1024: rows [0,512) on the 256x256-tile GEMM, the rest on 128x128 (read once by the library, csrc/p3v_runtime.hip; tests change it through ops.set_tuning)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU or without the HIP library."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
