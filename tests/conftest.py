import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle and the native library exist (cheap no-op when already built)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "jsplayer_amd", "libjspgen.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jsplayer_amd", "gen")])
    if not os.path.exists(os.path.join(ROOT, "jsplayer_amd", "libjsplayer_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jsplayer_amd", "csrc"), "-j8"])
