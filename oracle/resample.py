"""NumPy restatement of the TF half of tools/resampling_voxel_grid.py (lines 381-632) and of
tools/model_util.py:41-49, 77-100.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All arithmetic is float32, op by op, the way the TF graph evaluates it (no FMA contraction:
NumPy elementwise ops round after every multiply and add).
"""
import math
import numpy as np

F32 = np.float32


def rotation_around_grid_centroid(view_params):
    """tools/resampling_voxel_grid.py:515-562.  view_params [B,3] (azimuth, elevation, scale),
    radians.  Returns (Rot_Z @ Rot_Y, Scale), both [B,4,4] float32.  The 3-parameter branch is the
    one TF1 always takes (`tensor == 2` is a Python False, :551)."""
    vp = np.asarray(view_params, dtype=F32)
    B = vp.shape[0]
    az = vp[:, 0] - F32(math.pi * 0.5)                      # :529
    el = vp[:, 1]
    ca, sa = np.cos(az).astype(F32), np.sin(az).astype(F32)
    ce, se = np.cos(el).astype(F32), np.sin(el).astype(F32)
    rot_y = np.zeros((B, 4, 4), F32)                        # :537-541
    rot_y[:, 0, 0] = ca
    rot_y[:, 0, 2] = -sa
    rot_y[:, 1, 1] = 1
    rot_y[:, 2, 0] = sa
    rot_y[:, 2, 2] = ca
    rot_y[:, 3, 3] = 1
    rot_z = np.zeros((B, 4, 4), F32)                        # :544-548
    rot_z[:, 0, 0] = ce
    rot_z[:, 0, 1] = se
    rot_z[:, 1, 0] = -se
    rot_z[:, 1, 1] = ce
    rot_z[:, 2, 2] = 1
    rot_z[:, 3, 3] = 1
    rot = np.matmul(rot_z, rot_y).astype(F32)               # :550
    sc = np.zeros((B, 4, 4), F32)                           # :556-561
    s = vp[:, 2]
    sc[:, 0, 0] = s
    sc[:, 1, 1] = s
    sc[:, 2, 2] = s
    sc[:, 3, 3] = 1
    return rot, sc


def inverse_affine(view_params, size=64, new_size=128):
    """tools/resampling_voxel_grid.py:579-602: total_M = T_new_inv @ S @ R @ T, inverted in
    float32, rows 0:3.  Returns [B,3,4] float32 (output grid coords -> source grid coords)."""
    rot, sc = rotation_around_grid_centroid(view_params)
    B = rot.shape[0]
    T = np.tile(np.array([[1, 0, 0, -size * 0.5],
                          [0, 1, 0, -size * 0.5],
                          [0, 0, 1, -size * 0.5],
                          [0, 0, 0, 1]], F32)[None], (B, 1, 1))
    T_new_inv = np.tile(np.array([[1, 0, 0, new_size * 0.5],
                                  [0, 1, 0, new_size * 0.5],
                                  [0, 0, 1, new_size * 0.5],
                                  [0, 0, 0, 1]], F32)[None], (B, 1, 1))
    total = np.matmul(np.matmul(np.matmul(T_new_inv, sc), rot), T).astype(F32)   # :599
    inv = np.linalg.inv(total).astype(F32)                                        # :601
    return np.ascontiguousarray(inv[:, 0:3, :])                                   # :602


def voxel_meshgrid(height, width, depth):
    """tools/resampling_voxel_grid.py:488-513: 'ij' meshgrid of (depth, height, width), flattened
    with x fastest: flat n = z*H*W + y*W + x.  Returns x, y, z as float32 [N]."""
    z_t, y_t, x_t = np.meshgrid(np.arange(depth, dtype=F32), np.arange(height, dtype=F32),
                                np.arange(width, dtype=F32), indexing='ij')
    return x_t.reshape(-1), y_t.reshape(-1), z_t.reshape(-1)


def interpolate_one(vox, x, y, z):
    """tools/resampling_voxel_grid.py:381-486 for ONE batch item.  vox [H,W,D,C] float32 is
    addressed flat as (dim0=z, dim1=y, dim2=x) (:430-449); x,y,z float32 [N] source coordinates.
    Clamp-then-weight (:417-422, :465-482) and add_n order a..h (:485) are reproduced."""
    H, W, D, C = vox.shape
    max_y, max_x, max_z = H - 1, W - 1, D - 1                     # :405-407
    x0 = np.floor(x).astype(np.int32)
    y0 = np.floor(y).astype(np.int32)
    z0 = np.floor(z).astype(np.int32)
    x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    x0 = np.clip(x0, 0, max_x); x1 = np.clip(x1, 0, max_x)
    y0 = np.clip(y0, 0, max_y); y1 = np.clip(y1, 0, max_y)
    z0 = np.clip(z0, 0, max_z); z1 = np.clip(z1, 0, max_z)
    bz0, bz1 = z0 * (W * H), z1 * (W * H)
    flat = vox.reshape(-1, C)
    Ia = flat[bz0 + y0 * W + x0]
    Ib = flat[bz0 + y1 * W + x0]
    Ic = flat[bz0 + y0 * W + x1]
    Id = flat[bz0 + y1 * W + x1]
    Ie = flat[bz1 + y0 * W + x0]
    If = flat[bz1 + y1 * W + x0]
    Ig = flat[bz1 + y0 * W + x1]
    Ih = flat[bz1 + y1 * W + x1]
    x0f, x1f = x0.astype(F32), x1.astype(F32)
    y0f, y1f = y0.astype(F32), y1.astype(F32)
    z0f, z1f = z0.astype(F32), z1.astype(F32)
    wa = ((x1f - x) * (y1f - y) * (z1f - z))[:, None]
    wb = ((x1f - x) * (y - y0f) * (z1f - z))[:, None]
    wc = ((x - x0f) * (y1f - y) * (z1f - z))[:, None]
    wd = ((x - x0f) * (y - y0f) * (z1f - z))[:, None]
    we = ((x1f - x) * (y1f - y) * (z - z0f))[:, None]
    wf = ((x1f - x) * (y - y0f) * (z - z0f))[:, None]
    wg = ((x - x0f) * (y1f - y) * (z - z0f))[:, None]
    wh = ((x - x0f) * (y - y0f) * (z - z0f))[:, None]
    out = wa * Ia
    out = out + wb * Ib
    out = out + wc * Ic
    out = out + wd * Id
    out = out + we * Ie
    out = out + wf * If
    out = out + wg * Ig
    out = out + wh * Ih
    return out.astype(F32)


def source_coords(M, new_size, mode="tf"):
    """Source coordinates of every output grid point for one [3,4] float32 matrix.
    mode "tf":      grid_transform = total_M @ grid (:605) via a float32 matmul, as TF does.
    mode "ordered": x_s = ((m0*x + m1*y) + m2*z) + m3 with a rounding after every op -- the
                    operation order the HIP kernel uses (rn_resample_affine_fwd), so that the
                    affine entry point can be checked bit-for-bit."""
    gx, gy, gz = voxel_meshgrid(new_size, new_size, new_size)
    M = np.asarray(M, F32)
    if mode == "tf":
        grid = np.stack([gx, gy, gz, np.ones_like(gx)], 0)
        t = np.matmul(M, grid).astype(F32)
        return t[0], t[1], t[2]
    out = []
    for r in range(3):
        v = (M[r, 0] * gx + M[r, 1] * gy)
        v = v + M[r, 2] * gz
        v = v + M[r, 3]
        out.append(v.astype(F32))
    return out[0], out[1], out[2]


def resampling_affine(voxel_array, M_inv, new_size=128, mode="tf"):
    """tools/resampling_voxel_grid.py:603-610 given the inverted matrices.  voxel_array
    [B,S,S,S,C]; M_inv [B,3,4].  Returns [B,N,N,N,C] float32 indexed [b,z,y,x,c]."""
    vox = np.asarray(voxel_array, F32)
    B = vox.shape[0]
    C = vox.shape[4]
    out = np.empty((B, new_size, new_size, new_size, C), F32)
    for b in range(B):
        xs, ys, zs = source_coords(M_inv[b], new_size, mode)
        out[b] = interpolate_one(vox[b], xs, ys, zs).reshape(new_size, new_size, new_size, C)
    return out


def rotation_resampling(voxel_array, view_params, size=64, new_size=128, mode="tf"):
    """tools/resampling_voxel_grid.py:616-632 (tf_rotation_resampling)."""
    return resampling_affine(voxel_array, inverse_affine(view_params, size, new_size), new_size, mode)


def transform_voxel_to_match_image(t):
    """tools/model_util.py:41-49: transpose dims 1<->2 then reverse dim 1."""
    return np.ascontiguousarray(np.transpose(t, [0, 2, 1, 3, 4])[:, ::-1])


def crop_voxel_image(voxels, images, start, patch_size):
    """tools/model_util.py:77-100 with the random start point made an explicit argument
    (the reference draws it with seed=None, :92).  voxels [B,N,N,D,C]; images [B,4N,4N,ch] or None."""
    r, c = int(start[0]), int(start[1])
    vp = voxels[:, r:r + patch_size, c:c + patch_size]
    if images is None:
        return vp, None
    f = images.shape[1] // voxels.shape[1]
    ip = images[:, f * r:f * (r + patch_size), f * c:f * (c + patch_size)]
    return vp, ip


def net_input(voxel_array, view_params, size=64, new_size=128, mode="tf"):
    """RenderNet_Shader.py:150-151: resample then transform to image layout."""
    return transform_voxel_to_match_image(rotation_resampling(voxel_array, view_params, size, new_size, mode))


def resampling_affine_bwd(voxel_array, M_inv, dout, new_size=128):
    """What TensorFlow's autodiff derives from tf_resampling / tf_interpolate (tools/resampling_voxel_grid.py:381-486,
    :603-610) for a loss L with d L / d out = dout [B,N,N,N,C] (raw [b,z,y,x,c] order): floor / clip carry no
    gradient; each of the eight tf.gather's scatters weight * dout back (d L / d vox); the weights are linear in
    the coordinates, which are M_inv @ (gx, gy, gz, 1) (d L / d M_inv).  float64 arithmetic.
    Returns (dvox [B,S,S,S,C], dM [B,3,4])."""
    vox = np.asarray(voxel_array, np.float64)
    B, S, C = vox.shape[0], vox.shape[1], vox.shape[4]
    N = new_size
    dvox = np.zeros_like(vox)
    dM = np.zeros((B, 3, 4), np.float64)
    gx, gy, gz = (g.astype(np.float64) for g in voxel_meshgrid(N, N, N))
    G = np.stack([gx, gy, gz, np.ones_like(gx)], 1)                      # [N^3, 4]
    for b in range(B):
        xs, ys, zs = source_coords(M_inv[b], N, "ordered")               # the float32 coordinates the forward used
        x, y, z = xs.astype(np.float64), ys.astype(np.float64), zs.astype(np.float64)
        x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64); z0 = np.floor(z).astype(np.int64)
        x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
        x0 = np.clip(x0, 0, S - 1); x1 = np.clip(x1, 0, S - 1)
        y0 = np.clip(y0, 0, S - 1); y1 = np.clip(y1, 0, S - 1)
        z0 = np.clip(z0, 0, S - 1); z1 = np.clip(z1, 0, S - 1)
        ax, bx = x1 - x, x - x0
        ay, by = y1 - y, y - y0
        az, bz = z1 - z, z - z0
        d = np.asarray(dout[b], np.float64).reshape(-1, C)
        flat = vox[b].reshape(-1, C)
        dflat = dvox[b].reshape(-1, C)
        idx = lambda zz, yy, xx: (zz * S + yy) * S + xx
        taps = [(idx(z0, y0, x0), ax * ay * az), (idx(z0, y1, x0), ax * by * az), (idx(z0, y0, x1), bx * ay * az),
                (idx(z0, y1, x1), bx * by * az), (idx(z1, y0, x0), ax * ay * bz), (idx(z1, y1, x0), ax * by * bz),
                (idx(z1, y0, x1), bx * ay * bz), (idx(z1, y1, x1), bx * by * bz)]
        for ii, w in taps:
            np.add.at(dflat, ii, w[:, None] * d)
        va, vb, vc, vd, ve, vf, vg, vh = (flat[ii] for ii, _ in taps)
        gxs = np.sum(d * ((ay * az)[:, None] * (vc - va) + (by * az)[:, None] * (vd - vb) +
                          (ay * bz)[:, None] * (vg - ve) + (by * bz)[:, None] * (vh - vf)), 1)
        gys = np.sum(d * ((ax * az)[:, None] * (vb - va) + (bx * az)[:, None] * (vd - vc) +
                          (ax * bz)[:, None] * (vf - ve) + (bx * bz)[:, None] * (vh - vg)), 1)
        gzs = np.sum(d * ((ax * ay)[:, None] * (ve - va) + (ax * by)[:, None] * (vf - vb) +
                          (bx * ay)[:, None] * (vg - vc) + (bx * by)[:, None] * (vh - vd)), 1)
        dM[b, 0] = gxs @ G
        dM[b, 1] = gys @ G
        dM[b, 2] = gzs @ G
    return dvox, dM


def inverse_affine_f64(view_params, size=64, new_size=128):
    """float64 twin of inverse_affine (same matrix chain), for finite-difference Jacobians w.r.t. the pose."""
    vp = np.asarray(view_params, np.float64)
    out = np.zeros((vp.shape[0], 3, 4))
    for b, (az, el, s) in enumerate(vp):
        a = az - math.pi * 0.5
        ca, sa, ce, se = math.cos(a), math.sin(a), math.cos(el), math.sin(el)
        rot_y = np.array([[ca, 0, -sa, 0], [0, 1, 0, 0], [sa, 0, ca, 0], [0, 0, 0, 1.0]])
        rot_z = np.array([[ce, se, 0, 0], [-se, ce, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        sc = np.diag([s, s, s, 1.0])
        T = np.eye(4); T[:3, 3] = -size * 0.5
        Tn = np.eye(4); Tn[:3, 3] = new_size * 0.5
        out[b] = np.linalg.inv(Tn @ sc @ (rot_z @ rot_y) @ T)[:3]
    return out
