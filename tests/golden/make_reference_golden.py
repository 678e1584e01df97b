#!/usr/bin/env python
"""Golden vectors produced by the REFERENCE'S OWN CODE (/root/reference), for the parts of the path whose reference
implementation is pure NumPy and therefore runs in the build container even though TensorFlow does not:

  * tools/resampling_voxel_grid.py::np_interpolate (:19-128) -- the NumPy twin of tf_interpolate (:381-486): the same
    floor / clamp-then-weight / flat-index / eight-gather / weight / sequential-add arithmetic, statement for statement.
    It pins the interpolation kernel of oracle/resample.py (and through it the HIP resampler) bit for bit.
  * tools/Phong_shading.py: np_mask, np_mask_white, np_phong_shading, np_phong_composite, generate_light_pos (:138-253).
  * tools/binvox_rw.py: read_as_3d_array on the five shipped fixtures, write / save_binvox bytes (:45-93, :175-239).
  * RenderNet_demo.py::compute_pose_param (:33-38), tools/data_util.py::extract_param_from_names (:13-29).
  * tools/utils.py::NpyTarWriter (:24-44): a tar written by the reference, to be read by the mirror.

The modules import TensorFlow at the top (`import tensorflow as tf`) although these functions never touch it, so an
empty stand-in module is registered first; `np.bool` / `np.int` (removed in NumPy 2) are aliased for binvox_rw.py:85.
Nothing here is copied into the repository: the script imports the reference where it lies and stores only OUTPUTS.

    python tests/golden/make_reference_golden.py        # writes tests/golden/reference_vectors.npz

The TF graph ops themselves (tf.nn.conv3d / conv2d / *_transpose, tf.matrix_inverse, tf.gather) cannot be executed here;
for those the oracle stays unpinned (oracle/__init__.py).
"""
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("%s is not mounted: this script only runs in the build container" % REF)
    tf = types.ModuleType("tensorflow")
    tf.contrib = types.ModuleType("tensorflow.contrib")
    sys.modules.setdefault("tensorflow", tf)
    sys.modules.setdefault("tensorflow.contrib", tf.contrib)
    if not hasattr(np, "bool"):
        np.bool = bool
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, REF)
    import tools.resampling_voxel_grid as RV
    import tools.Phong_shading as PH
    import tools.binvox_rw as BV
    import tools.data_util as DU
    import tools.utils as UT
    import RenderNet_demo as DEMO
    return RV, PH, BV, DU, UT, DEMO


def interpolation_cases():
    """(name, voxel [B,S,S,S,1] float32, x, y, z float32 [B*n]) -- coordinates include out-of-range ones on every side."""
    sys.path.insert(0, ROOT)
    from oracle import resample as OR
    from oracle.io_phong import read_binvox
    rng = np.random.default_rng(20260926)
    cases = []
    S, n = 16, 6000
    vox = rng.standard_normal((2, S, S, S, 1)).astype(np.float32)
    xyz = rng.uniform(-2.5, S + 1.5, (3, 2 * n)).astype(np.float32)
    # exact grid points, exact borders and .5 points as well
    xyz[:, :64] = np.round(xyz[:, :64])
    xyz[:, 64:128] = np.round(xyz[:, 64:128] * 2) / 2
    cases.append(("random16", vox, xyz[0], xyz[1], xyz[2], S, n))
    # the chair fixture at the demo pose: every 37th sample of the 128^3 output grid, source coordinates of the TF path
    chair = read_binvox(os.path.join(ROOT, "binvox", "chair.binvox")).astype(np.float32)[None, ..., None]
    pose = np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0]], np.float32)
    M = OR.inverse_affine(pose, 64, 128)
    x, y, z = OR.source_coords(M[0], 128, "tf")
    sel = slice(5, None, 37)
    cases.append(("chair_demo_pose", chair, x[sel].copy(), y[sel].copy(), z[sel].copy(), 64, x[sel].size))
    # the same with the coordinate arithmetic in the order the HIP kernel uses (oracle mode "ordered"), so that the
    # kernel's own output at these samples can be compared with the reference's np_interpolate bit for bit
    x, y, z = OR.source_coords(M[0], 128, "ordered")
    cases.append(("chair_demo_pose_ordered", chair, x[sel].copy(), y[sel].copy(), z[sel].copy(), 64, x[sel].size))
    return cases, M


def main():
    RV, PH, BV, DU, UT, DEMO = import_reference()
    out = {}

    # -- np_interpolate ------------------------------------------------------------------------
    cases, M = interpolation_cases()
    out["interp_chair_M_inv"] = M
    for name, vox, x, y, z, S, n in cases:
        B = vox.shape[0]
        # out_size only sizes the per-item run of `base` (:68): [B, n, 1, 1] gives n samples per batch item
        got = RV.np_interpolate(vox, x, y, z, [B, n, 1, 1])
        out["interp_%s_vox" % name] = vox if S <= 16 else np.packbits(vox.astype(bool))
        out["interp_%s_x" % name], out["interp_%s_y" % name], out["interp_%s_z" % name] = x, y, z
        out["interp_%s_out" % name] = np.asarray(got, np.float32).reshape(-1)
        print(name, "samples", got.shape[0], "non-zero", int(np.count_nonzero(got)))

    # -- Phong ----------------------------------------------------------------------------------
    rng = np.random.default_rng(7)
    img = rng.uniform(0.02, 0.98, (2, 9, 7, 3)).astype(np.float64)
    img[0, 0, :, :] = np.linspace(0.33, 0.35, 7)[:, None]                 # black mask transition (|img| ~ 150/255)
    img[1, 1, :, :] = 1.0 - np.linspace(0.17, 0.20, 7)[:, None]           # white mask transition (|1-img| ~ 80/255)
    light = rng.standard_normal((2, 3))
    col = rng.uniform(0.5, 1.0, (2, 3))
    out["phong_img"], out["phong_light"], out["phong_col"] = img, light, col
    out["phong_mask"] = PH.np_mask(img)
    out["phong_mask_white"] = PH.np_mask_white(img)
    out["phong_shading"] = PH.np_phong_shading(img, light.copy(), col, 0.9)
    out["phong_black"] = PH.np_phong_composite(img, light.copy(), col, 0.1, 0.9, background_col="Black")
    out["phong_white"] = PH.np_phong_composite(img, light.copy(), col, 0.1, 0.9, background_col="white")
    out["phong_nomask"] = PH.np_phong_composite(img, light.copy(), col, 0.1, 0.9, with_mask=False)
    angles = np.array([[90, 90], [60, 250], [30, 10], [0, 0], [105, 294]], np.float64)
    out["light_angles"] = angles
    out["light_pos"] = np.concatenate([PH.generate_light_pos(e, a) for e, a in angles])

    # -- binvox ---------------------------------------------------------------------------------
    for name in ("chair", "bunny", "table", "suzanne", "teapot"):
        with open(os.path.join(ROOT, "binvox", name + ".binvox"), "rb") as f:
            m = BV.read_as_3d_array(f)
        out["binvox_%s_bits" % name] = np.packbits(np.asarray(m.data, bool))
        out["binvox_%s_meta" % name] = np.array(list(m.dims) + list(m.translate) + [m.scale], np.float64)
    rng = np.random.default_rng(3)
    small = rng.random((12, 12, 12)) < 0.3
    runs = np.zeros((255 * 3 + 7, 1, 1), bool)
    runs[255:510] = True                                                   # runs that are exact multiples of 255
    for name, arr in (("small", small), ("runs", runs)):
        f = io.BytesIO()
        BV.Voxels(arr, arr.shape, [0.0, 0.0, 0.0], 1.0, 'xyz').write(f)
        out["binvox_write_%s_in" % name] = arr
        out["binvox_write_%s_bytes" % name] = np.frombuffer(f.getvalue(), np.uint8)

    # -- pose helpers -----------------------------------------------------------------------------
    poses = np.array([[250, 60, 3.3], [0, 90, 3.3], [123.5, 10, 2.0], [359, 170, 5.0]], np.float64)
    out["pose_in"] = poses
    out["pose_out"] = np.concatenate([DEMO.compute_pose_param(*p) for p in poses])
    names = ["model_chair_abc_p250_t30_r3.3", "ply80055_p303_t108_r3.3_albedo", "model_chair_x1_p10_t100_r2.5.png"]
    out["names"] = np.array(names)
    out["names_param"] = np.concatenate([DU.extract_param_from_names(n) for n in names])

    # -- tar container ----------------------------------------------------------------------------
    path = "/tmp/_ref_writer.tar"
    w = UT.NpyTarWriter(path)
    arrs = [rng.standard_normal((3, 4)).astype(np.float32), np.arange(10, dtype=np.int64)]
    for i, a in enumerate(arrs):
        w.add(a, "entry%d" % i)
    w.close()
    out["tar_bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8)
    out["tar_entry0"], out["tar_entry1"] = arrs
    os.remove(path)

    dst = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
