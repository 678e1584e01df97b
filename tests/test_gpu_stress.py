"""BASELINE config 5 (high-res stress: 128^3 voxels -> 256^3 -> 1024x1024; 16.3 TMAC per frame, 3.79 GB of
weights).  The oracle needs minutes per frame on CPU, so the check is (a) a committed golden crop of one
frame produced by tests/golden/make_golden.py stress, and (b) a size-independent property: frames of a
batch are rendered independently (batch of 2 == two batches of 1, bit for bit).  -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, demo_pose

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]      # every test once per multiply-stage mode (conftest.py)


@pytest.fixture(scope="module")
def stress_renderer():
    from rendernet_amd.shader import Renderer, stress_spec, init_shader_weights
    spec = stress_spec(1)
    return Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True))


def _chair128(fixtures_vox):
    v = fixtures_vox[0, ..., 0]
    return v.repeat(2, 0).repeat(2, 1).repeat(2, 2)[None, ..., None]


def test_stress_frame_matches_golden(stress_renderer, fixtures_vox):
    path = os.path.join(GOLDEN_DIR, "stress_chair_demo_pose.npz")
    if not os.path.exists(path):
        pytest.skip("stress golden not generated")
    g = np.load(path)
    out = stress_renderer.render(_chair128(fixtures_vox), demo_pose()[None]).cpu().numpy()
    assert out.shape == (1, 1024, 1024, 1)
    crop = out[0, 448:576, 448:576, 0]
    assert np.abs(crop - g["output_crop"]).max() <= 1e-3
    lg = np.log(crop.astype(np.float64) / (1 - crop.astype(np.float64)))
    assert np.abs(lg - g["logits_crop"]).max() <= 1e-3 * np.abs(g["logits_crop"]).max() + 1e-5


def test_stress_batch_independence(stress_renderer, fixtures_vox, gemm_mode):
    """Frames of a batch are rendered independently: bit for bit in the exact and the bf16x3 modes (every launch plan sums a frame's
    products in the same order whatever the batch); split16 scales every tensor by ITS maximum, which a batch-mate can raise -- there
    the frame moves by fp32 rounding only."""
    vox = np.concatenate([_chair128(fixtures_vox), _chair128(fixtures_vox)[:, ::-1].copy()])
    poses = np.stack([demo_pose(), demo_pose(40, 35, 3.0)])
    both = stress_renderer.render(vox, poses)
    one = stress_renderer.render(vox[1:2], poses[1:2])
    if gemm_mode == "split16":
        assert float((both[1:2] - one).abs().max()) <= 1e-5
    else:
        assert torch.equal(both[1:2], one)
    assert bool(torch.isfinite(both).all()) and float(both.std()) > 0


def test_stress_batch8_matches_golden(stress_renderer):
    """BASELINE config 5 at its configured batch of 8 (the frames of `bench.py --mode stress`): one call, every frame's
    centre crop against the committed oracle output (tests/golden/make_golden.py stress8)."""
    path = os.path.join(GOLDEN_DIR, "stress_bench_frames.npz")
    if not os.path.exists(path):
        pytest.skip("stress batch golden not generated")
    from bench import synthetic_batch              # conftest.py puts the repo root on sys.path
    g = np.load(path)
    vox, poses = synthetic_batch(8, 2)
    out = stress_renderer.render(vox, poses).cpu().numpy()
    assert out.shape == (8, 1024, 1024, 1)
    n = 0
    for i in range(8):
        if "output_%d" % i not in g.files:
            continue
        crop = out[i, 448:576, 448:576, 0]
        assert np.abs(crop - g["output_%d" % i]).max() <= 1e-3, i
        lg = np.log(crop.astype(np.float64) / (1 - crop.astype(np.float64)))
        want = g["logits_%d" % i]
        assert np.abs(lg - want).max() <= 1e-3 * np.abs(want).max() + 1e-5, i
        n += 1
    assert n >= 4
