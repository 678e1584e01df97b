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


GEMM_MODES = ("split", "f32", "split16")        # rendernet_amd.ops.GEMM_MODES, the product default first


@pytest.fixture(params=GEMM_MODES)
def gemm_mode(request, monkeypatch):
    """Runs the test once per multiply-stage mode of the wide convs (rendernet_amd.ops.gemm_mode): "split" = the product default
    (bf16x3 operands, fp32 accumulate), "f32" = exact-fp32 MFMA everywhere (the fallback, RN_WINO_GEMM=f32), "split16" = the opt-in
    fp16x2 fast mode.  Whole modules opt in with `pytest.mark.usefixtures("gemm_mode")`: every net-level -m gpu test (configs 2, 3, 5,
    inverse rendering, the CLIs) is green in all three, at the same bars."""
    from rendernet_amd import ops
    monkeypatch.setattr(ops._MODE, "mode", request.param, raising=False)      # = `with ops.gemm_mode(...)` around the test
    return request.param
