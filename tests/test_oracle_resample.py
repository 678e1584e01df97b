"""Pins the resampler oracle (oracle/resample.py) with analytic known answers -- the reference ships
no test vectors for this path, so these identities are what anchors the oracle to
tools/resampling_voxel_grid.py:381-632."""
import numpy as np

from conftest import demo_pose
from oracle import resample as OR


def _vox(seed=0, S=16, C=1, binary=True):
    rng = np.random.default_rng(seed)
    v = rng.random((2, S, S, S, C))
    return (v < 0.3).astype(np.float32) if binary else v.astype(np.float32)


def test_identity_like_pose_is_a_shifted_copy(fixtures_vox):
    """az=pi/2, el=0, s=1 -> R = I: out[32:95]^3 == in[:63]^3 exactly; plane S-1 and everything
    outside [0,S-1) evaluate to 0 (clamp-then-weight, :417-422 / :465-482)."""
    pose = np.array([[np.pi / 2, 0.0, 1.0]], np.float32)
    out = OR.rotation_resampling(fixtures_vox[1:2], pose, 64, 128)      # bunny touches both faces
    core = out[0, 32:95, 32:95, 32:95, 0]
    # cos(pi/2) != 0 in float32: compare away from the discontinuous planes 0 and S-1
    src = fixtures_vox[1, :63, :63, :63, 0]
    assert np.abs(core[1:62, 1:62, 1:62] - src[1:62, 1:62, 1:62]).max() < 2e-5
    assert np.abs(out[0, :30]).max() < 1e-4 and np.abs(out[0, 98:]).max() < 1e-4


def test_exact_integer_affine():
    """With an exact integer translation matrix every weight is 0 or 1: bit-exact shifted copy,
    last source plane dropped, exact zeros outside."""
    v = _vox(1, 16)
    m = np.zeros((2, 3, 4), np.float32)
    m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = 1
    m[:, :, 3] = -8
    for mode in ("tf", "ordered"):
        out = OR.resampling_affine(v, m, 32, mode)
        assert np.array_equal(out[:, 8:23, 8:23, 8:23], v[:, :15, :15, :15])
        out[:, 8:23, 8:23, 8:23] = 0
        assert np.all(out == 0)


def test_axis_aligned_poses_are_signed_permutations():
    """(az,el) in {(180,0),(270,0),(0,0)} deg map interior source voxels onto a signed
    axis permutation of the grid (<= 1.5e-5), masking source planes 0 and S-1."""
    S, N = 16, 32
    v = _vox(2, S, binary=False)
    for az_deg in (0.0, 180.0, 270.0):
        pose = np.array([[az_deg * np.pi / 180, 0.0, 1.0]] * 2, np.float32)
        out = OR.rotation_resampling(v, pose, S, N)
        M = OR.inverse_affine(pose, S, N)[0]
        # evaluate the expected permutation from the matrix itself (rounded to integers)
        R = np.rint(M[:, :3]).astype(int)
        t = np.rint(M[:, 3]).astype(int)
        assert np.abs(M[:, :3] - R).max() < 1e-6
        zz, yy, xx = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
        src = [R[r, 0] * xx + R[r, 1] * yy + R[r, 2] * zz + t[r] for r in range(3)]    # x,y,z source
        ok = np.ones_like(xx, bool)
        for s_ in src:
            ok &= (s_ >= 1) & (s_ <= S - 3)
        want = v[0][src[2][ok], src[1][ok], src[0][ok], 0]
        got = out[0][..., 0][ok]
        assert ok.sum() > 1000
        assert np.abs(got - want).max() < 5e-5


def test_far_out_of_range_cancels(fixtures_vox):
    """Outside the source support the 8 terms cancel pairwise: |value| <= ~1e-4 (exactly 0 where the
    clamped border voxels are empty)."""
    pose = demo_pose()[None]
    out = OR.rotation_resampling(fixtures_vox[0:1], pose, 64, 128)
    corner = out[0, :8, :8, :8]
    assert np.abs(corner).max() <= 1e-4
    assert abs(float(out.sum()) - 5249.07) < 1.0         # SURVEY App. A.6 (chair, demo pose)
    assert out.min() >= -1e-4 and out.max() <= 1.0 + 1e-5


def test_demo_pose_matrix_matches_survey():
    """M_inv[:3] for az=250, el=60, r=3.3."""
    M = OR.inverse_affine(demo_pose()[None], 64, 128)[0]
    want = np.array([[-0.8138, 0.4698, 0.3420, 32.1236], [0.5, 0.8660, 0, -55.4256],
                     [-0.2962, 0.1710, -0.9397, 100.1524]])
    assert np.abs(M - want).max() < 2e-3


def test_tf_and_ordered_modes_agree():
    """The two coordinate evaluation orders differ by float32 rounding only."""
    v = _vox(3, 16, C=2, binary=False)
    rng = np.random.default_rng(4)
    poses = np.stack([rng.uniform(0, 6.28, 2), rng.uniform(0.2, 2.5, 2), rng.uniform(0.8, 1.2, 2)], 1).astype(np.float32)
    M = OR.inverse_affine(poses, 16, 32)
    a = OR.resampling_affine(v, M, 32, "tf")
    b = OR.resampling_affine(v, M, 32, "ordered")
    d = np.abs(a - b)
    assert (d > 1e-4).mean() < 1e-3


def test_transform_and_crop():
    """tools/model_util.py:47-48: X[b,i,j,k] = R[b,j,N-1-i,k]; :95-98 crop windows."""
    t = np.arange(2 * 4 * 4 * 4 * 1, dtype=np.float32).reshape(2, 4, 4, 4, 1)
    x = OR.transform_voxel_to_match_image(t)
    for i in range(4):
        for j in range(4):
            assert np.array_equal(x[:, i, j], t[:, j, 3 - i])
    img = np.arange(2 * 16 * 16 * 1, dtype=np.float32).reshape(2, 16, 16, 1)
    vp, ip = OR.crop_voxel_image(x, img, (1, 2), 2)
    assert vp.shape == (2, 2, 2, 4, 1) and ip.shape == (2, 8, 8, 1)
    assert np.array_equal(ip, img[:, 4:12, 8:16])
