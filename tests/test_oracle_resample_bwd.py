"""Pins the resampler-backward oracle (oracle/resample.py::resampling_affine_bwd) on CPU: the voxel gradient by the
adjoint identity <J e, dout> = <e, J^T dout> of the linear-in-voxels forward, the matrix gradient by central finite
differences of a float64 restatement of the forward, the pose Jacobian of the closed-form matrix by finite differences."""
import numpy as np

from oracle import resample as OR


def _f64_forward(vox, M, N):
    """float64 forward (same clamp-then-weight rule) for one item; vox [S,S,S,C], M [3,4]."""
    S, C = vox.shape[0], vox.shape[3]
    gx, gy, gz = (g.astype(np.float64) for g in OR.voxel_meshgrid(N, N, N))
    x = M[0, 0] * gx + M[0, 1] * gy + M[0, 2] * gz + M[0, 3]
    y = M[1, 0] * gx + M[1, 1] * gy + M[1, 2] * gz + M[1, 3]
    z = M[2, 0] * gx + M[2, 1] * gy + M[2, 2] * gz + M[2, 3]
    x0 = np.floor(x).astype(int); y0 = np.floor(y).astype(int); z0 = np.floor(z).astype(int)
    x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    cl = lambda v: np.clip(v, 0, S - 1)
    x0, x1, y0, y1, z0, z1 = cl(x0), cl(x1), cl(y0), cl(y1), cl(z0), cl(z1)
    f = vox.reshape(-1, C).astype(np.float64)
    ix = lambda zz, yy, xx: (zz * S + yy) * S + xx
    out = 0
    for (zz, wz) in ((z0, z1 - z), (z1, z - z0)):
        for (yy, wy) in ((y0, y1 - y), (y1, y - y0)):
            for (xx, wx) in ((x0, x1 - x), (x1, x - x0)):
                out = out + (wx * wy * wz)[:, None] * f[ix(zz, yy, xx)]
    return out


def _problem(seed=0, S=8, N=16, C=2):
    rng = np.random.default_rng(seed)
    vox = np.zeros((1, S, S, S, C), np.float32)
    vox[:, 1:-1, 1:-1, 1:-1] = rng.random((1, S - 2, S - 2, S - 2, C))      # empty border: the sampler is continuous
    pose = np.array([[1.0, 0.6, 0.9]], np.float32)
    M = OR.inverse_affine(pose, S, N)
    M[:, :, 3] += np.float32(0.0137)          # keep every sample off the cell boundaries (the interpolant has kinks there)
    dout = rng.standard_normal((1, N, N, N, C)).astype(np.float32)
    return vox, pose, M, dout, S, N


def test_voxel_gradient_is_the_adjoint_of_the_forward():
    vox, pose, M, dout, S, N = _problem()
    dv, _ = OR.resampling_affine_bwd(vox, M, dout, N)
    rng = np.random.default_rng(1)
    for _ in range(3):
        e = rng.standard_normal(vox.shape)
        lhs = np.sum((_f64_forward((vox[0] + e[0]), M[0].astype(np.float64), N) -
                      _f64_forward(vox[0], M[0].astype(np.float64), N)) * dout[0].reshape(-1, vox.shape[4]))
        assert abs(lhs - np.sum(e * dv)) <= 1e-4 * abs(lhs) + 1e-6          # coordinates differ at float32 rounding


def test_matrix_gradient_matches_finite_differences():
    vox, pose, M, dout, S, N = _problem()
    _, dM = OR.resampling_affine_bwd(vox, M, dout, N)
    M64 = M[0].astype(np.float64)
    d = dout[0].reshape(-1, vox.shape[4]).astype(np.float64)
    L = lambda Mx: float(np.sum(_f64_forward(vox[0], Mx, N) * d))
    for (r, c) in [(0, 0), (0, 3), (1, 1), (1, 2), (2, 0), (2, 3)]:
        h = 1e-7
        Mp, Mm = M64.copy(), M64.copy()
        Mp[r, c] += h; Mm[r, c] -= h
        fd = (L(Mp) - L(Mm)) / (2 * h)
        assert abs(fd - dM[0, r, c]) <= 2e-3 * max(abs(fd), abs(dM[0, r, c])) + 1e-3, (r, c, fd, dM[0, r, c])


def test_pose_jacobian_of_the_closed_form_matrix():
    """d M_inv / d (azimuth, elevation, scale): the float64 matrix chain differentiated numerically agrees with the
    closed form the kernels use (rt/s and t = S/2 - sum(a) * N/2)."""
    pose = np.array([[1.0, 0.6, 0.9]])
    S, N = 8, 16
    az, el, s = pose[0, 0] - np.pi / 2, pose[0, 1], pose[0, 2]
    ca, sa, ce, se = np.cos(az), np.sin(az), np.cos(el), np.sin(el)
    rt = np.array([[ce * ca, -se * ca, sa], [se, ce, 0.0], [-ce * sa, se * sa, ca]])
    M = OR.inverse_affine_f64(pose, S, N)[0]
    assert np.allclose(M[:, :3], rt / s, atol=1e-12)
    assert np.allclose(M[:, 3], S / 2 - (rt / s).sum(1) * N / 2, atol=1e-10)
    d_az = np.array([[-ce * sa, se * sa, ca], [0, 0, 0], [-ce * ca, se * ca, -sa]]) / s
    d_el = np.array([[-se * ca, -ce * ca, 0], [ce, -se, 0], [se * sa, ce * sa, 0]]) / s
    d_s = -rt / s ** 2
    for k, dA in enumerate((d_az, d_el, d_s)):
        h = 1e-6
        pp, pm = pose.copy(), pose.copy()
        pp[0, k] += h; pm[0, k] -= h
        fd = (OR.inverse_affine_f64(pp, S, N)[0] - OR.inverse_affine_f64(pm, S, N)[0]) / (2 * h)
        assert np.allclose(fd[:, :3], dA, atol=1e-6)
        assert np.allclose(fd[:, 3], -dA.sum(1) * N / 2, atol=1e-5)
