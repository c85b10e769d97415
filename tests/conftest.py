import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multifield-adaptive-retrieval_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):      # tests/: sibling test modules and tests/helpers import each other by name
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        from mfar import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: fail loudly rather than silently skipping (the product has no CPU fallback).
    # Plain runs on a CPU box skip the gpu-marked tests.
    if _gpu_present():
        return
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
