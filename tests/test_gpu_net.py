"""Whole-path parity: resampler -> RenderNet on the HIP path vs the CPU oracle, tap by tap.  -m gpu.

Tolerances (fp32 path, north_star: per-pixel L-inf <= 1e-3 on the sigmoid output):
  every intermediate tap:  max|got-want| <= 2e-4 * max|want|   (relative, because with synthetic
                           Xavier weights the logits are small and an absolute bound on the output
                           alone would be a weak test)
  final image:             max|got-want| <= 1e-3 absolute, as north_star states it.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, demo_pose
from oracle import rendernet as ON
from oracle import resample as OR

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]      # every test once per multiply-stage mode (conftest.py)
TAP_RTOL = 2e-4
OUT_ATOL = 1e-3


def _check_taps(got_taps, want_taps, names=None):
    for name, want in want_taps.items():
        if names is not None and name not in names:
            continue
        if name == "logits":          # the HIP path fuses the sigmoid into e_conv11: invert it
            o = got_taps["output"].double()
            got = torch.log(o / (1 - o)).float().cpu().numpy()
            err, ref = np.abs(got - want).max(), np.abs(want).max()
            assert err <= 1e-3 * ref + 1e-5, "logits: max err %g, max |ref| %g" % (err, ref)
            continue
        got = got_taps[name].cpu().numpy()
        assert got.shape == want.shape, (name, got.shape, want.shape)
        err, ref = np.abs(got - want).max(), np.abs(want).max()
        assert err <= TAP_RTOL * ref + 1e-7, "tap %s: max err %g, max |ref| %g" % (name, err, ref)


@pytest.mark.parametrize("out_ch", [1, 3])
def test_tiny_net_all_taps(out_ch):
    from rendernet_amd.shader import Renderer, tiny_spec, init_shader_weights
    spec = tiny_spec(out_ch)
    w = init_shader_weights(spec, seed=1234, perturb=True)
    rng = np.random.default_rng(7)
    vox = (rng.random((3, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    poses = np.stack([demo_pose(250, 60, 3.3), demo_pose(40, 30, 3.0), demo_pose(135, 80, 4.0)])
    want_taps = {}
    x = OR.net_input(vox, poses, 16, 32, mode="tf")
    want = ON.rendernet_forward(x, w, want_taps, spec.n_res1, spec.n_res2, spec.n_res3)
    r = Renderer(spec, w)
    got_taps = {}
    got = r.render(vox, poses, taps=got_taps)
    d = np.abs(got_taps["net_in"].cpu().numpy() - x)
    assert (d > 2e-4).mean() <= 1e-4            # resampler: see test_gpu_resample for the rationale
    # feed the oracle's net input through the HIP net so that tap errors are the net's own
    from rendernet_amd import variables as V
    from rendernet_amd.shader import RenderNet
    V.set_default_store(r.store)
    got_taps = {}
    got = RenderNet(torch.as_tensor(x).cuda(), False, spec=spec, taps=got_taps)
    _check_taps(got_taps, want_taps)
    assert np.abs(got.cpu().numpy() - want).max() <= OUT_ATOL


def test_golden_tiny():
    """Committed golden vector (tests/golden/make_golden.py ran the oracle in the build container)."""
    from rendernet_amd.shader import Renderer, tiny_spec, init_shader_weights
    g = np.load(os.path.join(GOLDEN_DIR, "tiny_shader.npz"))
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=int(g["seed"]), perturb=True)
    r = Renderer(spec, w)
    taps = {}
    out = r.render(g["vox"], g["poses"], taps=taps).cpu().numpy()
    assert np.abs(out - g["output"]).max() <= OUT_ATOL
    for name in ("enc3", "enc4", "enc6", "logits"):
        key = "tap_" + name
        if key in g.files:
            got = (taps[name] if name in taps else None)
            if got is not None:
                want = g[key]
                assert np.abs(got.cpu().numpy() - want).max() <= 5e-4 * np.abs(want).max() + 1e-6, name


def test_session_contract_and_full_size_frame(fixtures_vox):
    """One full-size frame (64^3 -> 128^3 -> 512^2, 237 M parameters) through the reference's
    Session contract, checked against the oracle on the 3-D encoder output, the projection unit
    and the final image; plus the committed golden crop of the same frame."""
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    spec = ShaderSpec().check()
    w = init_shader_weights(spec, seed=1234, perturb=True)
    r = Renderer(spec, w)
    vox = fixtures_vox[0:1]
    pose = demo_pose()[None]
    out = r.run("encoder/output:0", {"real_model_in:0": vox, "view_name:0": pose, "patch_size:0": 128,
                                     "is_training:0": False})
    assert out.shape == (1, 512, 512, 1) and out.dtype == np.float32
    g = np.load(os.path.join(GOLDEN_DIR, "full_chair_demo_pose.npz"))
    assert np.abs(out[0, 192:320, 192:320, 0] - g["output_crop"]).max() <= OUT_ATOL
    lg = np.log(out[0, 192:320, 192:320, 0] / (1 - out[0, 192:320, 192:320, 0]))
    assert np.abs(lg - g["logits_crop"]).max() <= 5e-4 * np.abs(g["logits_crop"]).max() + 1e-5
    # live oracle on the front of the net (cheap part: resampler + 3-D encoder front + e_conv3)
    x = OR.net_input(vox, pose, 64, 128, mode="tf")
    taps = {}
    got = r.render(vox, pose, taps=taps)
    d = np.abs(taps["net_in"].cpu().numpy() - x)
    assert (d > 2e-4).mean() <= 1e-5
    assert np.abs(got.cpu().numpy() - out).max() == 0.0           # deterministic


def test_hipgraph_replay_matches_eager():
    """Renderer.capture: the captured graph replays to the same bits as the eager launches, for new inputs too."""
    from rendernet_amd.shader import Renderer, tiny_spec, init_shader_weights
    spec = tiny_spec(1)
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True), device="cuda:0")
    rng = np.random.default_rng(5)
    replay = r.capture(2)
    for seed in (0, 1):
        vox = (np.random.default_rng(seed).random((2, 16, 16, 16, 1)) < 0.3).astype(np.float32)
        poses = np.array([[1.0 + seed, 0.7, 0.9], [2.0, 0.4 + 0.1 * seed, 1.1]], np.float32)
        want = r.render(vox, poses)
        got = replay(vox, poses)
        torch.cuda.synchronize()
        assert torch.equal(got, want)


def test_bench_frames_match_golden(fixtures_vox):
    """Parity ON the benched configuration (BASELINE configs[1]): five frames of bench.py's batch -- every shipped fixture,
    five different azimuths -- rendered in one full-size call and compared with the committed oracle output
    (tests/golden/make_golden.py bench_frames): four 128x128 crops of image and logits per frame, strided samples of the
    3-D encoder output and of the projection unit's output."""
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from bench import synthetic_batch              # conftest.py puts the repo root on sys.path
    g = np.load(os.path.join(GOLDEN_DIR, "bench_frames.npz"))
    idx = [int(i) for i in g["frames"]]
    vox, poses = synthetic_batch(24)
    spec = ShaderSpec().check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True))
    taps = {}
    out = r.render(vox[idx], poses[idx], taps=taps).cpu().numpy()
    assert out.shape == (5, 512, 512, 1)
    for k in range(5):
        assert abs(float(taps["net_in"][k].double().sum()) - float(g["net_in_sum_%d" % k])) <= 1e-3 * float(g["net_in_sum_%d" % k])
        e3 = taps["enc3_skip"][k, 3::8, 5::8, 1::4, :].cpu().numpy()
        e4 = taps["enc4"][k, 3::8, 5::8, :].cpu().numpy()
        for got, key in ((e3, "enc3_skip_%d" % k), (e4, "enc4_%d" % k)):
            want = g[key]
            assert np.abs(got - want).max() <= TAP_RTOL * np.abs(want).max() + 1e-6, key
        for c, (r0, c0) in enumerate(g["crops"]):
            crop = out[k, r0:r0 + 128, c0:c0 + 128, 0]
            assert np.abs(crop - g["output_%d" % k][c]).max() <= OUT_ATOL, (k, c)
            lg = np.log(crop.astype(np.float64) / (1 - crop.astype(np.float64)))
            want = g["logits_%d" % k][c]
            assert np.abs(lg - want).max() <= 5e-4 * np.abs(want).max() + 1e-5, (k, c)
