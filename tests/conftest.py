"""pytest configuration: registers the `gpu` marker and puts the host-side package on sys.path.

`-m "not gpu"` runs everywhere (oracle vs golden vectors, host logic, ABI export check, gloo
world_size-2 tests); `-m gpu` needs a real MI355X and goes through the C-ABI of libvlfb_hip.so.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib")
for p in (LIB, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
