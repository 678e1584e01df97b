import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BINVOX_DIR = os.path.join(ROOT, "binvox")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["chair", "bunny", "table", "suzanne", "teapot"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def fixtures_vox():
    """The five shipped 64^3 binvox fixtures as float32 [5,64,64,64,1] (oracle reader)."""
    from oracle.io_phong import read_binvox
    return np.stack([read_binvox(os.path.join(BINVOX_DIR, n + ".binvox")).astype(np.float32)[..., None]
                     for n in FIXTURES])


def demo_pose(az=250.0, el=60.0, r=3.3):
    return np.array([az * np.pi / 180.0, (90 - el) * np.pi / 180.0, 3.3 / r], np.float32)
