"""Pins the oracle (and the host-side mirrors) against outputs of the REFERENCE'S OWN NumPy code, generated in the build
container by tests/golden/make_reference_golden.py (which imports /root/reference with a stand-in for the TensorFlow
import) and committed as tests/golden/reference_vectors.npz:

  np_interpolate (tools/resampling_voxel_grid.py:19-128, the statement-for-statement NumPy twin of tf_interpolate
  :381-486), the NumPy Phong functions, binvox reader / writer, the pose helpers and the tar writer.

The TF graph ops (convolutions, matrix inverse) are not executable here, so the conv half of the oracle stays unpinned.
"""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, BINVOX_DIR, FIXTURES
from oracle import io_phong as OP
from oracle import resample as OR


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(GOLDEN_DIR, "reference_vectors.npz"))


def test_interpolation_kernel_is_bit_exact_against_np_interpolate(ref):
    """oracle/resample.py::interpolate_one == the reference's np_interpolate on the same voxels and coordinates:
    random float volume with coordinates out of range on every side, on grid points and on .5 points; the chair
    fixture at the demo pose (TF coordinate order and the HIP kernel's coordinate order)."""
    vox = ref["interp_random16_vox"]
    n = ref["interp_random16_x"].size // vox.shape[0]
    for b in range(vox.shape[0]):
        sl = slice(b * n, (b + 1) * n)
        got = OR.interpolate_one(vox[b], ref["interp_random16_x"][sl], ref["interp_random16_y"][sl], ref["interp_random16_z"][sl])
        assert np.array_equal(got.reshape(-1), ref["interp_random16_out"][sl])
    out = ref["interp_random16_out"]
    assert np.count_nonzero(out) > 5000 and np.abs(out).max() > 1.0          # a real test, not zeros against zeros
    for case in ("chair_demo_pose", "chair_demo_pose_ordered"):
        chair = np.unpackbits(ref["interp_%s_vox" % case]).reshape(64, 64, 64, 1).astype(np.float32)
        assert np.array_equal(chair[..., 0] > 0, OP.read_binvox(os.path.join(BINVOX_DIR, "chair.binvox")))
        got = OR.interpolate_one(chair, ref["interp_%s_x" % case], ref["interp_%s_y" % case], ref["interp_%s_z" % case])
        assert np.array_equal(got.reshape(-1), ref["interp_%s_out" % case])
        assert np.count_nonzero(ref["interp_%s_out" % case]) > 100


def test_oracle_coordinates_reproduce_the_golden_sample_points(ref):
    """The stored sample points ARE the oracle's source coordinates of the demo pose (so the full oracle render and the
    reference's interpolation meet on the same inputs)."""
    M = ref["interp_chair_M_inv"]
    pose = np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0]], np.float32)
    assert np.array_equal(OR.inverse_affine(pose, 64, 128), M)
    for mode, case in (("tf", "chair_demo_pose"), ("ordered", "chair_demo_pose_ordered")):
        x, y, z = OR.source_coords(M[0], 128, mode)
        sel = slice(5, None, 37)
        assert np.array_equal(x[sel], ref["interp_%s_x" % case]) and np.array_equal(z[sel], ref["interp_%s_z" % case])
    # and the whole-grid oracle output at those points is the reference's value
    chair = OP.read_binvox(os.path.join(BINVOX_DIR, "chair.binvox")).astype(np.float32)[None, ..., None]
    full = OR.resampling_affine(chair, M, 128, "ordered").reshape(-1)
    assert np.array_equal(full[5::37], ref["interp_chair_demo_pose_ordered_out"])


def test_numpy_phong_matches_reference(ref):
    img, light, col = ref["phong_img"], ref["phong_light"], ref["phong_col"]
    assert np.allclose(OP.np_mask(img), ref["phong_mask"], rtol=0, atol=1e-15)
    assert np.allclose(OP.np_mask_white(img), ref["phong_mask_white"], rtol=0, atol=1e-15)
    assert np.allclose(OP.np_phong_shading(img, light, col, 0.9), ref["phong_shading"], rtol=0, atol=1e-15)
    assert np.allclose(OP.np_phong_composite(img, light, col, 0.1, 0.9), ref["phong_black"], rtol=0, atol=1e-15)
    assert np.allclose(OP.np_phong_composite(img, light, col, 0.1, 0.9, background_col="white"), ref["phong_white"], rtol=0, atol=1e-15)
    assert np.allclose(OP.np_phong_composite(img, light, col, 0.1, 0.9, with_mask=False), ref["phong_nomask"], rtol=0, atol=1e-15)
    # the masks are really in their transition bands in this data
    for k in ("phong_mask", "phong_mask_white"):
        assert ((ref[k] > 0.05) & (ref[k] < 0.95)).sum() >= 3
    from rendernet_amd.tools.Phong_shading import generate_light_pos
    for (e, a), want in zip(ref["light_angles"], ref["light_pos"]):
        assert np.allclose(OP.generate_light_pos(e, a), want, rtol=0, atol=1e-15)
        assert np.allclose(generate_light_pos(e, a), want, rtol=0, atol=1e-15)


def test_binvox_reader_and_writer_match_reference(ref):
    from rendernet_amd.tools import binvox_rw as B
    for name in FIXTURES:
        bits = np.unpackbits(ref["binvox_%s_bits" % name]).reshape(64, 64, 64).astype(bool)
        meta = ref["binvox_%s_meta" % name]
        path = os.path.join(BINVOX_DIR, name + ".binvox")
        assert np.array_equal(OP.read_binvox(path), bits)
        with open(path, "rb") as f:
            m = B.read_as_3d_array(f)
        assert np.array_equal(m.data, bits)
        assert list(m.dims) == [64, 64, 64] == [int(v) for v in meta[:3]]
        assert np.allclose(list(m.translate) + [m.scale], meta[3:], rtol=0, atol=0)
    for name in ("small", "runs"):
        arr, want = ref["binvox_write_%s_in" % name], ref["binvox_write_%s_bytes" % name].tobytes()
        f = io.BytesIO()
        B.write(B.Voxels(arr, list(arr.shape), [0.0, 0.0, 0.0], 1.0, 'xyz'), f)
        # the header prints dims as the reference does for a tuple shape: "dim 12 12 12"
        assert f.getvalue() == want
        assert OP.write_binvox_bytes(arr, list(arr.shape)) == want


def test_pose_helpers_match_reference(ref):
    import RenderNet_demo
    from rendernet_amd.tools.data_util import extract_param_from_names
    for p, want in zip(ref["pose_in"], ref["pose_out"]):
        assert np.array_equal(OP.compute_pose_param(*p)[0], want)
        assert np.array_equal(np.asarray(RenderNet_demo.compute_pose_param(*p)).reshape(-1), want)
    for name, want in zip(ref["names"], ref["names_param"]):
        assert np.array_equal(np.asarray(extract_param_from_names(str(name))).reshape(-1), want)


def test_tar_written_by_the_reference_is_read_by_the_mirror(ref, tmp_path):
    from rendernet_amd.tools.utils import NpyTarReader
    path = str(tmp_path / "ref.tar")
    with open(path, "wb") as f:
        f.write(ref["tar_bytes"].tobytes())
    got = list(NpyTarReader(path))
    assert len(got) == 2
    assert np.array_equal(got[0], ref["tar_entry0"]) and np.array_equal(got[1], ref["tar_entry1"])
