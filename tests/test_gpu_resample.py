"""Parity of the HIP resampler (through the C ABI) against oracle/resample.py.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import demo_pose
from oracle import resample as OR

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(a).cuda()


def _poses(n, seed=0):
    rng = np.random.default_rng(seed)
    p = np.stack([rng.uniform(0, 2 * np.pi, n), rng.uniform(0.2, 2.6, n), rng.uniform(0.75, 1.3, n)], 1)
    return p.astype(np.float32)


def test_affine_bit_exact_raw_layout(fixtures_vox):
    """rn_resample_affine_fwd evaluates the same operations in the same order as the oracle's
    'ordered' mode: bit-for-bit equality on all five fixtures (integer-exact bar for byte-identical
    arithmetic, not a tolerance)."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    poses = np.stack([demo_pose(), demo_pose(10, 20, 2.8), demo_pose(123, 75, 4.2), demo_pose(300, 45, 3.0),
                      demo_pose(77, 10, 3.6)])
    m_inv = OR.inverse_affine(poses, 64, 128)
    want = OR.resampling_affine(fixtures_vox, m_inv, 128, mode="ordered")
    got = tf_resampling_affine(_dev(fixtures_vox), _dev(m_inv), 128, image_layout=False).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), "max |diff| = %g" % np.abs(got - want).max()


def test_affine_bit_exact_image_layout_and_crop(fixtures_vox):
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    poses = _poses(5, 3)
    m_inv = OR.inverse_affine(poses, 64, 128)
    want = OR.transform_voxel_to_match_image(OR.resampling_affine(fixtures_vox, m_inv, 128, mode="ordered"))
    got = tf_resampling_affine(_dev(fixtures_vox), _dev(m_inv), 128, image_layout=True).cpu().numpy()
    assert np.array_equal(got, want)
    # crop window (tools/model_util.py:95-98) folded into the kernel
    r, c, p = 40, 8, 64
    gotc = tf_resampling_affine(_dev(fixtures_vox), _dev(m_inv), 128, image_layout=True, window=(r, c, p, p))
    assert np.array_equal(gotc.cpu().numpy(), want[:, r:r + p, c:c + p])


def test_pose_path_matches_tf_oracle(fixtures_vox):
    """rn_resample_fwd (pose -> closed-form matrix in-kernel) against the TF-faithful oracle
    (float32 matrix chain + LU inverse + matmul).  The sampler is discontinuous at x=0- and
    x=(S-1)-, so a ~1e-6 coordinate difference may flip isolated samples next to
    occupied border voxels: tolerance 2e-4 on all but <= 1e-5 of the samples."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_rotation_resampling
    poses = np.stack([demo_pose(250 + 15 * i, 60, 3.3) for i in range(5)])
    want = OR.rotation_resampling(fixtures_vox, poses, 64, 128, mode="tf")
    got = tf_rotation_resampling(_dev(fixtures_vox), _dev(poses), 64, 128).cpu().numpy()
    d = np.abs(got - want)
    frac = float((d > 2e-4).mean())
    assert frac <= 1e-5, "fraction of samples off by > 2e-4: %g (max %g)" % (frac, d.max())
    assert abs(float(got.sum()) - float(want.sum())) / float(want.sum()) < 1e-4


def test_pose_to_affine_closed_form():
    from rendernet_amd import ops
    poses = _poses(16, 1)
    want = OR.inverse_affine(poses, 64, 128)
    got = ops.pose_to_affine(_dev(poses), 64, 128).cpu().numpy()
    assert np.abs(got[:, :, :3] - want[:, :, :3]).max() < 1e-6
    assert np.abs(got[:, :, 3] - want[:, :, 3]).max() < 2e-5


def test_multichannel_and_small_grids():
    """C = 4 (texture grids, RenderNet_Texture_Face_Normal.py:169-171), C = 5 (generic path), and
    ragged sizes 16^3 -> 32^3 / 32^3 -> 64^3."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    rng = np.random.default_rng(5)
    for S, N, C in ((16, 32, 4), (32, 64, 1), (16, 32, 5), (8, 16, 2)):
        vox = rng.standard_normal((3, S, S, S, C)).astype(np.float32)
        m_inv = OR.inverse_affine(_poses(3, S), S, N)
        want = OR.transform_voxel_to_match_image(OR.resampling_affine(vox, m_inv, N, mode="ordered"))
        got = tf_resampling_affine(_dev(vox), _dev(m_inv), N, image_layout=True).cpu().numpy()
        assert np.array_equal(got, want), (S, N, C, np.abs(got - want).max())


def test_known_answers_identity_like_pose(fixtures_vox):
    """az = pi/2, el = 0, s = 1 makes R the identity: out[32:95]^3 == in[:63]^3 exactly and the
    last source plane contributes nothing."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    m = np.zeros((5, 3, 4), np.float32)
    m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = 1.0
    m[:, :, 3] = -32.0
    got = tf_resampling_affine(_dev(fixtures_vox), _dev(m), 128, image_layout=False).cpu().numpy()
    assert np.array_equal(got[:, 32:95, 32:95, 32:95], fixtures_vox[:, :63, :63, :63])
    inner = got[:, 32:96, 32:96, 32:96].copy()
    got[:, 32:95, 32:95, 32:95] = 0
    assert np.all(got == 0), "everything outside [0,S-1) must cancel to exactly 0 for this pose"
    assert np.all(inner[:, 63] == 0) and np.all(inner[:, :, 63] == 0) and np.all(inner[:, :, :, 63] == 0)


def test_argument_errors():
    from rendernet_amd import ops
    from rendernet_amd._lib import RenderNetHipError
    vox = torch.zeros((1, 8, 8, 8, 1), device="cuda")
    pose = torch.zeros((1, 3), device="cuda")
    with pytest.raises(RenderNetHipError):
        ops.resample(vox, pose, new_size=12)            # not a supported grid size
    with pytest.raises(RenderNetHipError):
        ops.resample(vox, pose, new_size=16, window=(8, 8, 16, 16))   # window outside the grid
    with pytest.raises(RenderNetHipError):
        ops.resample(vox.cpu(), pose.cpu(), new_size=16)  # no CPU path


def test_five_column_poses_render_like_their_first_three_columns(fixtures_vox):
    """tf_rotation_translation_resampling (tools/resampling_voxel_grid.py:634-650) documents [B,5] poses (azimuth, elevation, scale,
    shiftX, shiftY) and reads columns 0..2 only: the mirror accepts any [B,>=3] and ignores the rest, like the reference; two
    columns fail in the reference at graph construction (column 2 is indexed unconditionally) and raise here."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_rotation_resampling, tf_rotation_translation_resampling, rotation_resampling_to_image
    vox = _dev(fixtures_vox[:2])
    p3 = np.stack([demo_pose(250, 60, 3.3), demo_pose(40, 20, 2.8)]).astype(np.float32)
    p5 = np.concatenate([p3, np.array([[3.0, -7.0], [0.5, 11.0]], np.float32)], 1)
    want = tf_rotation_resampling(vox, _dev(p3), 64, 128)
    assert torch.equal(tf_rotation_translation_resampling(vox, _dev(p5), 64, 128), want)
    assert torch.equal(tf_rotation_resampling(vox, _dev(p5), 64, 128), want)
    assert torch.equal(rotation_resampling_to_image(vox, _dev(p5), 64, 128), rotation_resampling_to_image(vox, _dev(p3), 64, 128))
    with pytest.raises(ValueError):
        tf_rotation_translation_resampling(vox, _dev(p3[:, :2].copy()), 64, 128)


def test_every_path_of_the_tiled_resampler_is_bit_exact(fixtures_vox):
    """The tiled kernel has data-dependent paths: occupancy-grid fast path (taps read from the bitmap) vs float
    gathers, per-sample bit test vs the fallback for boxes too large for the LDS window (zoomed-out poses), occupied
    volume borders (clamped taps with non-cancelling values), a batch that mixes the kinds.  All must equal the
    oracle bit for bit, and the two exact-result switches of RN_RS_DEBUG must not change a bit either."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    rng = np.random.default_rng(11)
    S, N = 64, 128
    sparse_float = (rng.random((1, S, S, S, 1)) < 0.03) * rng.standard_normal((1, S, S, S, 1))        # float values
    dense_binary = (rng.random((1, S, S, S, 1)) < 0.3).astype(np.float32)                                 # occupied borders
    half_values = (fixtures_vox[1:2] * 0.5)                                                               # {0, 0.5}: not an occupancy grid
    vox = np.concatenate([fixtures_vox[0:1], sparse_float, dense_binary, half_values]).astype(np.float32)
    for scales in ((1.0, 1.0, 1.0, 1.0), (0.45, 2.5, 0.6, 1.8)):
        poses = np.stack([[1.0 + i, 0.5 + 0.3 * i, s] for i, s in enumerate(scales)]).astype(np.float32)
        m_inv = OR.inverse_affine(poses, S, N)
        want = OR.transform_voxel_to_match_image(OR.resampling_affine(vox, m_inv, N, mode="ordered"))
        got = tf_resampling_affine(_dev(vox), _dev(m_inv), N, image_layout=True).cpu().numpy()
        assert np.array_equal(got, want), "scales %s: max |diff| = %g" % (scales, np.abs(got - want).max())


def test_empty_and_full_grids(fixtures_vox):
    """Degenerate occupancies: an all-zero grid (no candidate tile anywhere: the output is pure fill), an all-one
    grid (every tile a candidate, clamped border taps everywhere) and two isolated voxels (a corner, an interior one)."""
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    S, N = 64, 128
    corner = np.zeros((1, S, S, S, 1), np.float32)
    corner[0, 0, S - 1, 0, 0] = 1.0                                       # border voxel: only clamped taps reach it
    corner[0, 5, 40, 17, 0] = 1.0                                         # and one isolated interior voxel
    vox = np.concatenate([np.zeros((1, S, S, S, 1), np.float32), np.ones((1, S, S, S, 1), np.float32), corner])
    poses = np.stack([demo_pose(250, 60, 3.3), demo_pose(10, 30, 2.0), demo_pose(135, 80, 3.3)])
    m_inv = OR.inverse_affine(poses, S, N)
    want = OR.transform_voxel_to_match_image(OR.resampling_affine(vox, m_inv, N, mode="ordered"))
    got = tf_resampling_affine(_dev(vox), _dev(m_inv), N, image_layout=True).cpu().numpy()
    assert np.array_equal(got, want)
    # all-one grid: samples outside the volume along an axis take both taps from the clamped border voxel with weights
    # that cancel only up to rounding (a few 1e-5), exactly as the reference's clamp-then-weight arithmetic does
    assert not got[0].any() and got[1].max() <= 1.0 + 1e-5 and got[1].min() >= -1e-4 and got[2].any()


def test_hip_resampler_matches_the_reference_np_interpolate_bit_for_bit():
    """tests/golden/reference_vectors.npz holds outputs of the REFERENCE'S OWN np_interpolate
    (tools/resampling_voxel_grid.py:19-128, the NumPy twin of tf_interpolate) for the chair fixture at the demo pose,
    every 37th sample of the 128^3 grid: the HIP kernel's samples at those points are identical."""
    import os
    from conftest import GOLDEN_DIR
    from rendernet_amd.tools.resampling_voxel_grid import tf_resampling_affine
    ref = np.load(os.path.join(GOLDEN_DIR, "reference_vectors.npz"))
    chair = np.unpackbits(ref["interp_chair_demo_pose_ordered_vox"]).reshape(1, 64, 64, 64, 1).astype(np.float32)
    got = tf_resampling_affine(_dev(chair), _dev(ref["interp_chair_M_inv"]), 128, image_layout=False).cpu().numpy().reshape(-1)
    want = ref["interp_chair_demo_pose_ordered_out"]
    assert np.count_nonzero(want) > 100
    assert np.array_equal(got[5::37], want)
    # float-valued volume through the general (non-occupancy-grid) path: scale the chair by 0.37
    got2 = tf_resampling_affine(_dev(chair * np.float32(0.37)), _dev(ref["interp_chair_M_inv"]), 128, image_layout=False)
    nz = want != 0
    assert np.abs(got2.cpu().numpy().reshape(-1)[5::37][nz] / np.float32(0.37) - want[nz]).max() <= 1e-6


def test_concat_resampler_equals_two_resamplers_and_their_gradients():
    """rn_resample_concat_fwd (the face renderer's two tf_rotation_resampling calls + tf.concat,
    RenderNet_Texture_Face_Normal.py:165-178) is bit-identical, channel by channel, to the two separate calls, with and
    without a crop window; its backward (rn_resample_affine_bwd_strided per source) matches theirs."""
    from rendernet_amd import ops
    rng = np.random.default_rng(11)
    B, S, N = 2, 16, 32
    va = torch.as_tensor((rng.random((B, S, S, S, 1)) < 0.3).astype(np.float32)).cuda()
    vb = torch.as_tensor(rng.standard_normal((B, S, S, S, 4)).astype(np.float32)).cuda()
    pose = torch.as_tensor(np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0], [1.0, 0.7, 0.9]], np.float32)).cuda()
    for window in (None, (8, 16, 16, 8)):
        got = ops.resample_concat(va, vb, pose, N, window)
        want = torch.cat([ops.resample(va, pose, N, window), ops.resample(vb, pose, N, window)], dim=4)
        assert torch.equal(got, want)
    a1, b1, p1 = va.clone().requires_grad_(True), vb.clone().requires_grad_(True), pose.clone().requires_grad_(True)
    a2, b2, p2 = va.clone().requires_grad_(True), vb.clone().requires_grad_(True), pose.clone().requires_grad_(True)
    g = torch.as_tensor(rng.standard_normal((B, N, N, N, 5)).astype(np.float32)).cuda()
    ops.resample_concat(a1, b1, p1, N).backward(g)
    torch.cat([ops.resample(a2, p2, N), ops.resample(b2, p2, N)], dim=4).backward(g)
    for x, y, name in ((a1.grad, a2.grad, "dvox_a"), (b1.grad, b2.grad, "dvox_b"), (p1.grad, p2.grad, "dpose")):
        assert float((x - y).abs().max()) <= 1e-4 * float(y.abs().max()) + 1e-6, name      # atomics: order differs


def test_concat_resampler_brick_form_full_size_and_extreme_boxes(fixtures_vox):
    """The brick form of rn_resample_concat_fwd (8^3-output bricks, the taps' bounding box staged in LDS) at the face
    renderer's size (64^3 -> 128^3, geometry + dense 4-channel volume): bit-identical to the two separate resampler calls
    for the bench poses, for scales whose source box overflows the LDS box (0.2: a brick spans ~20 voxels -> the global
    gather branch) or collapses (3.0), for volumes entirely off one side (taps clamp to the border plane), through the
    affine entry, and for a window that is not a multiple of 8 (the line-per-thread kernel)."""
    from rendernet_amd import ops
    rng = np.random.default_rng(5)
    geo = torch.as_tensor(np.ascontiguousarray(fixtures_vox[[1, 3]])).cuda()
    tex = torch.as_tensor(rng.standard_normal((2, 64, 64, 64, 4)).astype(np.float32)).cuda()
    poses = [[250 * np.pi / 180, 30 * np.pi / 180, 1.0], [0.3, 1.2, 0.85], [2.0, 0.1, 0.2], [4.0, 0.6, 3.0], [1.0, 0.5, 1.14]]
    for k in range(0, len(poses) - 1):
        pose = torch.as_tensor(np.array(poses[k:k + 2], np.float32)).cuda()
        for window in (None, (40, 8, 64, 96), (3, 5, 30, 28)):
            got = ops.resample_concat(geo, tex, pose, 128, window)
            want = torch.cat([ops.resample(geo, pose, 128, window), ops.resample(tex, pose, 128, window)], dim=4)
            assert torch.equal(got, want), (k, window)
    m = ops.pose_to_affine(torch.as_tensor(np.array(poses[:2], np.float32)).cuda(), 64, 128)
    m[1, :, 3] += 90.0                                                # the whole volume off one side: every tap clamps
    got = ops.resample_concat(geo, tex, m, 128, None, True, True)
    want = torch.cat([ops.resample(geo, m, 128, None, True, True), ops.resample(tex, m, 128, None, True, True)], dim=4)
    assert torch.equal(got, want)


def test_concat_entry_from_pose_equals_the_affine_form():
    """ops.resample_concat turns poses into matrices with rn_pose_to_affine first; the C entry fed the poses themselves
    (affine = 0: the closed form evaluated inside the kernel) writes the same bits."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    rng = np.random.default_rng(2)
    B, S, N = 2, 16, 32
    va = torch.as_tensor((rng.random((B, S, S, S, 1)) < 0.3).astype(np.float32)).cuda()
    vb = torch.as_tensor(rng.standard_normal((B, S, S, S, 4)).astype(np.float32)).cuda()
    pose = torch.as_tensor(np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0], [1.0, 0.7, 0.9]], np.float32)).cuda()
    want = ops.resample_concat(va, vb, pose, N)
    for (h0, w0, ph, pw) in ((0, 0, N, N), (1, 2, 30, 16)):              # brick form / line-per-thread form
        got = torch.empty((B, ph, pw, N, 5), device="cuda")
        L.check(L.lib().rn_resample_concat_fwd(L.ptr(va), 1, L.ptr(vb), 4, L.ptr(pose), 0, L.ptr(got), B, S, N, h0, w0, ph, pw, 1,
                                               L.stream_ptr()), "rn_resample_concat_fwd")
        assert torch.equal(got, want[:, h0:h0 + ph, w0:w0 + pw])
