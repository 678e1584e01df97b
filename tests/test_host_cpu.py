"""CPU-side checks: binvox reader, C-ABI symbol export, host logic (variable names, shapes,
parameter count, pose convention, CLI surface), weight packing geometry.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import BINVOX_DIR, FIXTURES, ROOT


def test_binvox_fixtures_decode(fixtures_vox):
    """Occupancy counts of the five shipped fixtures, product reader == oracle reader."""
    from rendernet_amd.tools import binvox_rw
    want = {"chair": 5277, "bunny": 57835, "table": 61087, "suzanne": 30270, "teapot": 27933}
    for i, n in enumerate(FIXTURES):
        with open(os.path.join(BINVOX_DIR, n + ".binvox"), "rb") as f:
            v = binvox_rw.read_as_3d_array(f)
        assert v.dims == [64, 64, 64] and v.axis_order == "xyz" and v.data.dtype == bool
        assert int(v.data.sum()) == want[n]
        assert np.array_equal(v.data.astype(np.float32), fixtures_vox[i, ..., 0])
    occ = np.argwhere(fixtures_vox[0, ..., 0] > 0)
    assert (occ.min(0) == [17, 0, 12]).all() and (occ.max(0) == [46, 63, 51]).all()     # chair bbox


def test_binvox_rejects_garbage(tmp_path):
    from rendernet_amd.tools import binvox_rw
    p = tmp_path / "bad.binvox"
    p.write_bytes(b"#notbinvox 1\n")
    with open(p, "rb") as f, pytest.raises(IOError):
        binvox_rw.read_as_3d_array(f)
    p.write_bytes(b"#binvox 1\ndim 4 4 4\ntranslate 0 0 0\nscale 1\ndata\n\x01\x05")
    with open(p, "rb") as f, pytest.raises(IOError):
        binvox_rw.read_as_3d_array(f)                 # decodes to 5 voxels, header says 64
    # fix_coords=False keeps the file's xzy order
    payload = bytes([1, 3, 0, 61])
    p.write_bytes(b"#binvox 1\ndim 4 4 4\ntranslate 0 0 0\nscale 1\ndata\n" + payload)
    with open(p, "rb") as f:
        a = binvox_rw.read_as_3d_array(f, fix_coords=False)
    with open(p, "rb") as f:
        b = binvox_rw.read_as_3d_array(f)
    assert a.axis_order == "xzy" and np.array_equal(np.transpose(a.data, (0, 2, 1)), b.data)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly the functions include/rendernet_hip.h declares,
    and the ctypes binding table lists each of them."""
    from rendernet_amd import _lib
    from rendernet_amd.build import build
    so = build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "rendernet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    assert L.rn_version() == 192  # bump with the header
    # geometry helper is pure host code: [phase][ceil(K/4)][Npad][4]
    n = L.rn_packed_weight_floats(_lib.RN_PACK_CONV, 3, _lib.ivec([5, 5, 5]), 1, 8)
    assert n == 1 * 32 * 32 * 4                               # K=125 -> 32 quads, Npad 32
    n = L.rn_packed_weight_floats(_lib.RN_PACK_CONVT_S2, 2, _lib.ivec([4, 4]), 256, 128)
    assert n == 4 * (4 * 256 // 4) * 128 * 4
    assert L.rn_packed_weight_floats(_lib.RN_PACK_CONVT_S2, 2, _lib.ivec([3, 3]), 8, 8) == 0
    assert b"k=4" in L.rn_last_error()


def test_ops_fail_loudly_without_gpu():
    import torch
    from rendernet_amd import ops
    from rendernet_amd._lib import RenderNetHipError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RenderNetHipError):
        ops.resample(torch.zeros(1, 8, 8, 8, 1), torch.zeros(1, 3), 16)
    from rendernet_amd.shader import Renderer, tiny_spec
    with pytest.raises(RuntimeError):
        Renderer(tiny_spec(), device="cpu")


def test_variable_names_and_parameter_count():
    """TF variable names and the 237,270,425-parameter total (App. B)."""
    from rendernet_amd.shader import ShaderSpec, shader_variable_shapes, init_shader_weights, tiny_spec
    shapes = shader_variable_shapes(ShaderSpec().check())
    names = [n for n, _, _ in shapes]
    assert len(names) == len(set(names))
    assert sum(int(np.prod(s)) for _, s, _ in shapes) == 237270425
    for must in ("encoder/e_conv1/e_conv1/weights", "encoder/e_conv1/alpha", "encoder/res1_7/con1_3X3/weights",
                 "encoder/res1_7/conv2_3x3/biases", "encoder/res1_skip/con1_3X3/weights",
                 "encoder/projection_unit/Conv/weights", "encoder/projection_unit/alpha",
                 "encoder/res2_10/alpha", "encoder/res2_skip/con1_3X3/biases", "encoder/e_conv5/e_conv5/weights",
                 "encoder/res3_5/conv2_3x3/weights", "encoder/e_conv7_1/e_conv7_1/weights", "encoder/e_conv11/weights"):
        assert must in names, must
    d = dict((n, s) for n, s, _ in shapes)
    assert d["encoder/e_conv1/e_conv1/weights"] == [5, 5, 5, 1, 8]
    assert d["encoder/e_conv7/e_conv7/weights"] == [4, 4, 128, 256]          # [kh,kw,Cout,Cin]
    assert d["encoder/projection_unit/Conv/weights"] == [1, 1, 1024, 1024]
    w = init_shader_weights(tiny_spec(3), seed=5)
    assert w["encoder/e_conv11/weights"].shape == (4, 4, 3, 16)
    assert np.all(w["encoder/e_conv1/e_conv1/biases"] == np.float32(0.001))   # tools/layer_util.py:142
    assert np.all(w["encoder/res2_1/con1_3X3/biases"] == 0) and np.all(w["encoder/e_conv5/alpha"] == 0)
    lim = np.sqrt(6.0 / (125 * (1 + 8)))
    assert np.abs(w["encoder/e_conv1/e_conv1/weights"]).max() <= lim
    w2 = init_shader_weights(tiny_spec(3), seed=5)
    assert all(np.array_equal(w[k], w2[k]) for k in w)                         # seeded


def test_variable_store_scopes_on_cpu():
    from rendernet_amd import variables as V
    st = V.VariableStore("cpu", seed=0)
    with st.variable_scope("encoder"):
        with st.variable_scope("e_conv1"):
            v, name = st.get_variable("weights", [3, 3, 2, 4], V.xavier_initializer())
            assert name == "encoder/e_conv1/weights" and tuple(v.shape) == (3, 3, 2, 4)
            v2, _ = st.get_variable("weights", [3, 3, 2, 4], V.xavier_initializer())
            assert v2 is v
            with pytest.raises(ValueError):
                st.get_variable("weights", [1, 1, 2, 4], V.xavier_initializer())
    with pytest.raises(KeyError):
        st.get_variable("missing")
    sd = st.state_dict()
    st2 = V.VariableStore("cpu")
    st2.load_state_dict(sd)
    assert st2.num_parameters() == 72


def test_pose_convention_and_demo_cli_surface():
    """RenderNet_demo.py:33-38 pose parameters and :72-108 flags."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("RenderNet_demo", os.path.join(ROOT, "RenderNet_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    p = demo.compute_pose_param(250, 60, 3.3)
    assert p.shape == (1, 3)
    assert np.allclose(p[0], [250 * np.pi / 180, 30 * np.pi / 180, 1.0])
    args = demo.build_parser().parse_args([])
    assert (args.azimuth, args.elevation, args.light_azimuth, args.light_elevation, args.radius) == (250, 60, 250, 60, 3.3)
    assert args.render_dir == "./render" and args.rotate is False
    assert demo.build_parser().parse_args(["--rotate", "True"]).rotate is True
    from oracle.io_phong import compute_pose_param as ocp
    assert np.allclose(ocp(123, 45, 2.9), demo.compute_pose_param(123, 45, 2.9))


def test_phong_light_and_oracle_composite():
    from rendernet_amd.tools.Phong_shading import generate_light_pos
    from oracle import io_phong as OP
    assert np.allclose(generate_light_pos(60, 250), OP.generate_light_pos(60, 250))
    l = generate_light_pos(90, 90)
    assert np.allclose(l, [[0, 0, -1]], atol=1e-7) or np.allclose(l, [[-6.1e-17, 6.1e-17, -1]], atol=1e-6)
    img = np.random.default_rng(0).uniform(0, 1, (1, 4, 4, 3))
    out = OP.np_phong_composite(img, OP.generate_light_pos(60, 250), np.array([[1., 1., 1.]]), 0.1, 0.9)
    assert out.shape == img.shape and out.min() >= 0 and out.max() <= 1


def test_data_loader_tar_and_pose_parsing(tmp_path):
    """Mirror of tools/data_util.py / tools/utils.py: image tar whose member names carry the pose + binvox folder."""
    import io
    import shutil
    from PIL import Image
    from rendernet_amd.tools import utils, data_util
    p = data_util.extract_param_from_names("model_chair_abc_p250_t30_r3.3")
    assert p.shape == (1, 3)
    assert np.allclose(p[0], [250 * np.pi / 180, (90 - 30) * np.pi / 180, 1.0])
    assert np.allclose(data_util.extract_param_from_names("x_p10_t100_r2.5.png")[0], [10 * np.pi / 180, -10 * np.pi / 180, 3.3 / 2.5])
    with pytest.raises(ValueError):
        data_util.extract_param_from_names("no_pose_here.png")
    shutil.copy(os.path.join(ROOT, "binvox", "chair.binvox"), tmp_path / "model_chair_abc_clean.binvox")
    shutil.copy(os.path.join(ROOT, "binvox", "table.binvox"), tmp_path / "model_normalized_xyz_clean.binvox")
    tarp = str(tmp_path / "imgs.tar")
    w = utils.NpyTarWriter(tarp)
    rng = np.random.default_rng(0)
    imgs = []
    for name in ("model_chair_abc_p250_t30_r3.3", "model_chair_xyz_p10_t100_r2.5", "model_chair_abc_p90_t60_r3.3"):
        img = (rng.random((32, 32, 3)) * 255).astype(np.uint8)
        imgs.append(img)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="PNG")
        w.add_bytes(buf.getvalue(), name + ".png")
    w.add(np.arange(6, dtype=np.float32).reshape(2, 3), "arr")
    w.close()
    # raw reader: images come back as float32 with their member stem, arrays from .npy.z members
    items = list(utils.NpyTarReader(tarp))
    assert items[0][1] == "model_chair_abc_p250_t30_r3.3" and np.array_equal(items[0][0], imgs[0].astype(np.float32))
    assert np.array_equal(items[3], np.arange(6, dtype=np.float32).reshape(2, 3))
    # batches of 2, greyscale: tail of 1 is repeated up to a batch (tools/data_util.py:144-157)
    cfg = {"batch_size": 2, "batches_chunk": 1}
    chunks = list(data_util.data_loader(cfg, tarp, str(tmp_path), flatten=True, img_res=32))
    assert len(chunks) == 2
    ims, mods, params, names = chunks[0]
    assert ims.shape == (2, 32, 32, 1) and mods.shape == (2, 64, 64, 64, 1) and params.shape == (2, 3)
    assert np.allclose(ims[0, :, :, 0], imgs[0].astype(np.float32).mean(axis=2))
    assert mods[0].sum() != mods[1].sum()                       # chair vs the model_normalized_ fallback (table)
    assert np.allclose(params[1], [10 * np.pi / 180, -10 * np.pi / 180, 3.3 / 2.5])
    ims2, mods2, params2, names2 = chunks[1]
    assert ims2.shape[0] == 2 and np.array_equal(ims2[0], ims2[1]) and list(names2) == [names2[0]] * 2
    # colour images keep 3 channels
    ims3 = next(data_util.data_loader(cfg, tarp, str(tmp_path), flatten=False, img_res=32))[0]
    assert ims3.shape == (2, 32, 32, 3) and np.array_equal(ims3[0], imgs[0].astype(np.float32))


def test_binvox_writer_matches_reference_state_machine(tmp_path):
    """tools/binvox_rw.py:175-239: the vectorised writer emits the bytes of the reference's per-voxel state machine
    (runs cut at 255, the (value, 0) pair after a run that is a multiple of 255) and round-trips through the reader."""
    import io
    from rendernet_amd.tools import binvox_rw as B
    from oracle import io_phong as OP
    rng = np.random.default_rng(0)
    cases = [np.zeros((8, 8, 8), bool), np.ones((8, 8, 8), bool), rng.random((7, 9, 9)) < 0.3, rng.random((16, 16, 16)) < 0.02]
    d = np.zeros((255 * 3, 1, 1), bool); d[255 * 2:] = True            # 510 zeros, 255 ones (final run a multiple of 255)
    cases.append(d)
    d = np.zeros((255 * 3 + 7, 1, 1), bool); d[255:510] = True         # 255 zeros, 255 ones, 262 zeros
    cases.append(d)
    for d in cases:
        f = io.BytesIO()
        B.write(B.Voxels(d, list(d.shape), [0.0, 0.0, 0.0], 1.0, 'xyz'), f)
        assert f.getvalue() == OP.write_binvox_bytes(d, list(d.shape))
        assert np.array_equal(B.read_as_3d_array(io.BytesIO(f.getvalue())).data, d)
    path = str(tmp_path / "v.binvox")
    B.save_binvox(cases[3], path)
    with open(path, 'rb') as fh:
        m = B.read_as_3d_array(fh)
    assert np.array_equal(m.data, cases[3]) and m.dims == [16, 16, 16] and m.scale == 1.0


def test_texture_crop_helpers_share_one_window():
    """tools/model_util.py:102-160: voxel grid, texture grid, image and normal map are cropped with ONE window (image
    side scaled by image_dim / voxel_dim); patch == grid is the identity."""
    import torch
    from rendernet_amd.tools import model_util as M
    v = torch.arange(2 * 8 * 8 * 4).float().reshape(2, 8, 8, 4, 1)
    t = v * 2
    im = torch.arange(2 * 32 * 32 * 3).float().reshape(2, 32, 32, 3)
    n = im + 1
    a, b, c, d = M.tf_random_crop_voxel_texture_image_normal(v, t, im, n, 4, start_point=(1, 3))
    assert torch.equal(a, v[:, 1:5, 3:7]) and torch.equal(b, t[:, 1:5, 3:7])
    assert torch.equal(c, im[:, 4:20, 12:28]) and torch.equal(d, n[:, 4:20, 12:28])
    x = M.tf_random_crop_voxel_texture_image(v, t, im, 8)
    assert x[0] is v and x[1] is t and x[2] is im
    g = torch.Generator().manual_seed(0)
    a2, _, c2 = M.tf_random_crop_voxel_texture_image(v, t, im, 4, generator=g)
    assert a2.shape == (2, 4, 4, 4, 1) and c2.shape == (2, 16, 16, 3)


def test_create_tar_feeds_the_tar_reader(tmp_path):
    """tools/create_TAR.py -> tools/utils.py::NpyTarReader: PNG members come back as (float32 image, name without extension)."""
    from PIL import Image
    from rendernet_amd.tools.create_TAR import create_tar
    from rendernet_amd.tools.utils import NpyTarReader
    rng = np.random.default_rng(0)
    imgs = {}
    for name in ("model_chair_a_p250_t30_r3.3", "model_chair_b_p10_t100_r3.3"):
        arr = (rng.random((16, 16)) * 255).astype(np.uint8)
        Image.fromarray(arr).save(str(tmp_path / (name + ".png")))
        imgs[name] = arr
    (tmp_path / "notes.txt").write_text("ignored")
    tarp = str(tmp_path / "set.tar")
    assert create_tar(str(tmp_path), tarp) == 2
    got = dict((n, im) for im, n in NpyTarReader(tarp))
    assert set(got) == set(imgs)
    for n in imgs:
        assert got[n].dtype == np.float32 and np.array_equal(got[n], imgs[n].astype(np.float32))


def test_philox_known_answers_and_dropout_oracle_statistics():
    """oracle/dropout.py: Philox4x32-10 against the known-answer vectors published with Random123 (kat_vectors:
    counter / key all zero, all ones, and the digits of pi), then tf.nn.dropout's formula on its uniforms."""
    from oracle import dropout as OD
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = OD.philox4x32_10([ctr[0]], [ctr[1]], [ctr[2]], [ctr[3]], key[0], key[1])
        assert tuple(int(v[0]) for v in got) == want
    u = OD.uniforms(200003, seed=7, stream=3)
    assert u.dtype == np.float32 and 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3
    x = np.ones(200003, np.float32)
    y = OD.dropout(x, 0.75, 7, 3)
    assert set(np.unique(y)) == {np.float32(0.0), np.float32(1.0) / np.float32(0.75)}
    assert abs((y != 0).mean() - 0.75) < 5e-3 and abs(y.mean() - 1.0) < 5e-3
    assert np.array_equal(OD.dropout(x, 1.0, 7, 3), x)


def test_texture_face_loader_parses_names_and_pads(tmp_path):
    """data_loader_image_texture_normal_face (tools/data_util.py:159-233): pose / model / texture ids from the member
    names, normal maps from PNGs, a short tail repeated up to one batch."""
    import io
    import shutil
    from PIL import Image
    from rendernet_amd.tools import utils
    from rendernet_amd.tools.data_util import data_loader_image_texture_normal_face
    models, tex, nrm = tmp_path / "m", tmp_path / "t", tmp_path / "n"
    for d in (models, tex, nrm):
        d.mkdir()
    shutil.copy(os.path.join(BINVOX_DIR, "chair.binvox"), models / "faceply003.binvox")
    code = np.arange(199, dtype=np.float32)
    np.save(tex / "beta003.npy", code)
    tarp = str(tmp_path / "faces.tar")
    w = utils.NpyTarWriter(tarp)
    names = ["faceply003_p250_t30_r3.3", "faceply003_p10_t100_r2.5", "faceply003_p90_t60_r4.0"]
    rng = np.random.default_rng(0)
    for nme in names:
        buf = io.BytesIO()
        Image.fromarray((rng.random((32, 32, 3)) * 255).astype(np.uint8)).save(buf, format="PNG")
        w.add_bytes(buf.getvalue(), nme + ".png")
        Image.fromarray(np.full((32, 32, 3), 77, np.uint8)).save(nrm / (nme + ".png"))
    w.close()
    cfg = {"batch_size": 2, "batches_chunk": 1}
    chunks = list(data_loader_image_texture_normal_face(cfg, tarp, str(models), str(tex), str(nrm), img_res=32, add_noise=False))
    assert [len(c[5]) for c in chunks] == [2, 2]                               # 3 samples -> a full batch + a padded tail
    ims, nrms, mods, texs, params, nm = chunks[0]
    assert ims.shape == (2, 32, 32, 3) and nrms.shape == (2, 32, 32, 3) and mods.shape == (2, 64, 64, 64, 1)
    assert np.array_equal(texs[0], code) and float(nrms.max()) == 77.0 and mods.sum() > 0
    assert np.allclose(params[0], [250 * np.pi / 180, 60 * np.pi / 180, 1.0]) and np.allclose(params[1][2], 3.3 / 2.5)
    assert list(chunks[1][5]) == [names[2], names[2]]


def test_state_dict_loading_skips_optimizer_state_and_seeds_mix():
    """A training checkpoint also holds '__adam_m__', '__adam_v__', '__global_step__', '__epoch__' (train.py): the inference
    entry points must not upload them as variables.  ops.mix_seed: deterministic, rank-separating."""
    from rendernet_amd import ops
    from rendernet_amd.variables import VariableStore
    st = VariableStore("cpu")
    st.load_state_dict({"encoder/e_conv1/e_conv1/weights": np.ones((5, 5, 5, 1, 8), np.float32), "__adam_m__": np.zeros(1000, np.float32),
                        "__adam_v__": np.zeros(1000, np.float32), "__global_step__": np.int64(7), "__epoch__": np.int64(2),
                        "__l1_all__": np.zeros(3)})
    assert list(st.vars) == ["encoder/e_conv1/e_conv1/weights"] and st.num_parameters() == 1000
    seeds = {ops.mix_seed(1234, r) for r in range(64)}
    assert len(seeds) == 64 and all(0 <= s < 2 ** 64 for s in seeds)
    assert ops.mix_seed(1234, 3) == ops.mix_seed(1234, 3) != ops.mix_seed(1235, 3)
    ops.seed_dropout(5, 9 << 16)
    assert ops.dropout_state() == (5, 9 << 16)
    ops.seed_dropout(0x5EED0FD50)


def _pb_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_field(num, payload, wt=2):
    if wt == 2:
        return _pb_varint(num << 3 | 2) + _pb_varint(len(payload)) + payload
    if wt == 0:
        return _pb_varint(num << 3) + _pb_varint(payload)
    raise ValueError(wt)


def _pb_const_node(name, arr, how="content", op="Const"):
    """A NodeDef as TensorFlow serializes a Const: name, op, attr{dtype}, attr{value: tensor}."""
    import struct
    dt = {np.dtype(np.float32): 1, np.dtype(np.int32): 3}[arr.dtype]
    shape = b"".join(_pb_field(2, _pb_field(1, d, 0)) for d in arr.shape)
    tensor = _pb_field(1, dt, 0) + _pb_field(2, shape)
    if how == "content":
        tensor += _pb_field(4, arr.astype("<" + arr.dtype.str[1:]).tobytes())
    elif how == "packed":
        tensor += _pb_field(5, struct.pack("<%df" % arr.size, *arr.ravel()))
    elif how == "unpacked":
        tensor += b"".join(_pb_varint(5 << 3 | 5) + struct.pack("<f", float(v)) for v in arr.ravel())
    elif how == "splat":
        tensor += _pb_field(5, struct.pack("<f", float(arr.ravel()[0])))
    elif how == "ints":
        tensor += _pb_field(7, b"".join(_pb_varint(int(v)) for v in arr.ravel()))
    attr_value = _pb_field(5, _pb_field(1, b"value") + _pb_field(2, _pb_field(8, tensor)))
    attr_dtype = _pb_field(5, _pb_field(1, b"dtype") + _pb_field(2, _pb_field(6, dt, 0)))
    return _pb_field(1, name.encode()) + _pb_field(2, op.encode()) + attr_dtype + attr_value


def test_frozen_graph_reader_decodes_const_nodes_without_tensorflow(tmp_path):
    """rendernet_amd/tools/graphdef.py: the demo's frozen `.pb` (RenderNet_demo.py:23-30, :111; demo/RenderNet_converter.py) is a
    GraphDef whose variables are Const nodes.  The test serializes a GraphDef itself -- protobuf wire format by hand: tensors as
    tensor_content, packed / unpacked float_val, the one-value splat, an int32 constant, Identity and Placeholder nodes, the
    `versions` field -- and reads it back."""
    from rendernet_amd.tools.graphdef import read_graphdef_constants, load_frozen_weights, GraphDefError
    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    al = rng.standard_normal(8).astype(np.float32)
    nodes = [
        _pb_const_node("encoder/res2_1/con1_3X3/weights", w, "content"),
        _pb_field(1, b"encoder/res2_1/con1_3X3/weights/read") + _pb_field(2, b"Identity") + _pb_field(3, b"encoder/res2_1/con1_3X3/weights"),
        _pb_const_node("encoder/res2_1/con1_3X3/biases", b, "packed"),
        _pb_const_node("encoder/res2_1/alpha", al, "unpacked"),
        _pb_const_node("encoder/zeros", np.full((2, 5), 0.25, np.float32), "splat"),
        _pb_const_node("encoder/Reshape/shape", np.array([-1, 64, 64, 1024], np.int32), "ints"),
        _pb_field(1, b"real_model_in") + _pb_field(2, b"Placeholder"),
    ]
    graph = b"".join(_pb_field(1, n) for n in nodes) + _pb_field(4, _pb_field(1, 26, 0))         # + VersionDef{producer: 26}
    path = tmp_path / "frozen.pb"
    path.write_bytes(graph)
    c = read_graphdef_constants(str(path))
    assert sorted(c) == ["encoder/res2_1/alpha", "encoder/res2_1/con1_3X3/biases", "encoder/res2_1/con1_3X3/weights", "encoder/zeros"]
    assert np.array_equal(c["encoder/res2_1/con1_3X3/weights"], w) and c["encoder/res2_1/con1_3X3/weights"].dtype == np.float32
    assert np.array_equal(c["encoder/res2_1/con1_3X3/biases"], b) and np.array_equal(c["encoder/res2_1/alpha"], al)
    assert np.array_equal(c["encoder/zeros"], np.full((2, 5), 0.25, np.float32))
    ints = read_graphdef_constants(graph, float_only=False)["encoder/Reshape/shape"]
    assert ints.tolist() == [-1, 64, 64, 1024]
    want = {"encoder/res2_1/con1_3X3/weights": w, "encoder/res2_1/alpha": al}
    got = load_frozen_weights(graph, want)
    assert list(got) == list(want) and all(np.array_equal(got[k], want[k]) for k in want)
    with pytest.raises(GraphDefError, match="lacks"):
        load_frozen_weights(graph, {"encoder/e_conv1/e_conv1/weights": np.zeros((5, 5, 5, 1, 8), np.float32)})
    with pytest.raises(GraphDefError, match="shape"):
        load_frozen_weights(graph, {"encoder/res2_1/alpha": np.zeros(16, np.float32)})
    with pytest.raises(GraphDefError):
        read_graphdef_constants(graph[:-3] + b"\xff\xff\xff")                                       # truncated / garbage tail
    # expected shapes without materialised weights; malformed tensors raise GraphDefError, never struct.error / ValueError / MemoryError
    got = load_frozen_weights(graph, {"encoder/res2_1/alpha": (8,)})
    assert np.array_equal(got["encoder/res2_1/alpha"], al)
    with pytest.raises(GraphDefError, match="shape"):
        load_frozen_weights(graph, {"encoder/res2_1/alpha": (16,)})

    def tensor_node(name, tensor_bytes):
        attr = _pb_field(1, b"value") + _pb_field(2, _pb_field(8, tensor_bytes))
        return _pb_field(1, _pb_field(1, name) + _pb_field(2, b"Const") + _pb_field(5, attr))

    shape2 = _pb_field(2, _pb_field(2, _pb_field(1, 2, 0)))                                             # TensorShapeProto{dim{size: 2}}
    bad = {
        "odd tensor_content": _pb_field(1, 1, 0) + shape2 + _pb_field(4, b"\x00" * 7),
        "odd packed float_val": _pb_field(1, 1, 0) + shape2 + _pb_field(5, b"\x00" * 6),
        "negative dim": _pb_field(1, 1, 0) + _pb_field(2, _pb_field(2, _pb_field(1, (1 << 64) - 1, 0))) + _pb_field(5, b"\x00" * 4),
        "huge splat": _pb_field(1, 1, 0) + _pb_field(2, b"".join(_pb_field(2, _pb_field(1, 1 << 20, 0)) for _ in range(3))) + _pb_field(5, b"\x00" * 4),
    }
    for what, t in bad.items():
        with pytest.raises(GraphDefError):
            read_graphdef_constants(tensor_node(b"encoder/bad", t))


def test_bench_golden_parity_indexing_and_per_rank_fields():
    """bench.py's self-check against the committed oracle renders (tests/golden/*.npz): an output assembled FROM the goldens
    passes with error 0, a perturbed one fails, frames a rank does not hold are not counted -- for the render, texture and
    stress lines; per_rank_fields aggregates min / mean / max."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    z = np.load(os.path.join(ROOT, "tests", "golden", "bench_frames.npz"))
    out = torch.zeros((24, 512, 512, 1))
    for k, f in enumerate(z["frames"].tolist()):
        for c, (r0, c0) in enumerate(z["crops"].tolist()):
            out[f, r0:r0 + 128, c0:c0 + 128, 0] = torch.from_numpy(z["output_%d" % k][c])
    rec = bench.golden_parity("render", out, list(range(24)))
    assert rec["ok"] and rec["max_abs_err"] == 0.0 and rec["frames"] == [0, 6, 12, 18, 9]
    rec3 = bench.golden_parity("render", out[:3], [0, 1, 2])                      # an 8-way strong split: rank 0 holds frames 0-2
    assert rec3["ok"] and rec3["frames"] == [0]
    assert bench.golden_parity("render", out[1:3], [1, 2]) is None               # no golden frame held: no record
    bad = out.clone()
    bad[6, 130, 260, 0] += 0.01
    assert not bench.golden_parity("render", bad, list(range(24)))["ok"]
    zt = np.load(os.path.join(ROOT, "tests", "golden", "texture_bench_frames.npz"))
    outt = torch.zeros((4, 512, 512, 6))
    for k, f in enumerate(zt["frames"].tolist()):
        outt[f, 192:320, 192:320, 0:3] = torch.from_numpy(zt["image_%d" % k])
        outt[f, 192:320, 192:320, 3:6] = torch.from_numpy(zt["normal_%d" % k])
    rt = bench.golden_parity("texture", outt, [0, 1, 2, 3])
    assert rt["ok"] and rt["max_abs_err"] == 0.0 and rt["frames"] == [0, 1, 2, 3]
    zs = np.load(os.path.join(ROOT, "tests", "golden", "stress_bench_frames.npz"))
    outs = torch.zeros((2, 1024, 1024, 1))
    for f in range(2):
        outs[f, 448:576, 448:576, 0] = torch.from_numpy(zs["output_%d" % f])
    assert bench.golden_parity("stress", outs, [0, 1])["max_abs_err"] == 0.0
    pr = bench.per_rank_fields([0.2, 0.1, 0.3], [3, 3, 3], 10)
    assert pr["ms_per_step_per_rank"] == {"min": 10.0, "mean": 20.0, "max": 30.0} and pr["per_rank"]["units_per_step"] == [3, 3, 3]


def test_split_filter_gradient_planner_is_a_host_function():
    """rn_winograd_split_wgrad_workspace_bytes plans K splits and tile padding on the host (csrc/conv_wino_bf3_wgrad.hip: wgrad_plan):
    the res2 training shape (T = 1536 tiles, 576 blocks = 4.5 half rounds of 5) takes none; a 256 -> 256 layer is cut in three;
    a single tile is padded to one even pair of K steps."""
    from rendernet_amd import _lib as L
    lib = L.lib()
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, 24, 32, 32, 1024, 1024) == 36 * 1536 * 2048 * 6 + 36 * 1024 * 1024 * 4 + 256
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, 24, 32, 32, 256, 256) == 108 * 512 * 512 * 6 + 108 * 256 * 256 * 4 + 256
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F44, 1, 1, 1, 256, 256) == 49 * 32 * 512 * 6 + 49 * 256 * 256 * 4 + 256
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, 24, 32, 32, 128, 256) == 0           # unsupported widths
    assert lib.rn_winograd_split_wgrad_supported(L.RN_WINO_F63, 1024, 1024) == 0                            # 4x4-output schemes only


def test_default_multiply_stage_mode_and_bench_blocks():
    """Round 5: the product default is the bf16x3 split multiply stage (rendernet_amd.ops.WINO_GEMM = "split"), RN_WINO_GEMM overrides it
    and is validated at import; bench.py times the default as `value` and the two other modes as the named blocks `exact` / `alt` / `alt2`;
    the header's 1x1 scheme constant and the Python mirror agree."""
    import subprocess
    import sys
    import bench
    from rendernet_amd import _lib
    code = "import sys; sys.path.insert(0, %r); from rendernet_amd import ops; print(ops.WINO_GEMM)" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "RN_WINO_GEMM"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "split"
    assert subprocess.run([sys.executable, "-c", code], env=dict(env, RN_WINO_GEMM="f32"), capture_output=True, text=True).stdout.strip() == "f32"
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, RN_WINO_GEMM="bf16"), capture_output=True, text=True)
    assert bad.returncode != 0 and "RN_WINO_GEMM" in bad.stderr
    assert bench.other_modes("split") == (("exact", "f32"), ("alt2", "split16"))
    assert bench.other_modes("f32") == (("alt", "split"), ("alt2", "split16"))
    assert bench.other_modes("split16") == (("exact", "f32"), ("alt", "split"))
    assert set(bench.LINE_DTYPE) == set(bench.ALT_DTYPE) == set(bench.ALT_WHAT) == set(bench.TRAIN_DTYPE) == {"f32", "split", "split16"}
    assert bench.LINE_DTYPE["split"].startswith("f32 (multiply stages: bf16x3-split operands") and bench.LINE_DTYPE["f32"] == "f32"
    hdr = open(os.path.join(ROOT, "include", "rendernet_hip.h")).read()
    assert re.search(r"#define\s+RN_WINO_F11\s+3\b", hdr) and _lib.RN_WINO_F11 == 3
    # the training roofline of a mode is priced against that mode's peak
    class _Ev:
        def __init__(self, ms): self.ms = ms
        def elapsed_time(self, other): return other.ms - self.ms
    ev = [((_Ev(0.0), _Ev(0.5)), (1536, 1024, 1024, "f43"))]
    r32, rs, rh = (bench.train_stage_roofline(ev, "gemm", m, 1024) for m in ("f32", "split", "split16"))
    fl = 2.0 * 36 * 1536 * 1024 * 1024
    assert r32["peak"] == 157.3 and abs(rs["peak"] - 2500 / 6) < 0.01 and abs(rh["peak"] - 2500 / 3) < 0.01
    for r in (r32, rs, rh):
        assert abs(r["achieved"] - fl / 0.5e-3 / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "peak_name" in rs and "peak_name" not in r32


def test_multiply_stage_mode_is_a_per_call_context_not_module_state():
    """Round 6 (review item 10): the mode of a launch is the innermost `with ops.gemm_mode(m)` of the CALLING THREAD, else the process
    default (env RN_WINO_GEMM); Renderer / TextureRenderer / Trainer / Reconstructor carry it as the attribute `gemm` and enter the
    context around what they launch.  No library code writes ops.WINO_GEMM (checked on the sources), so two renderers of one process
    can run different modes (GPU side: tests/test_gpu_wino_split.py::test_two_renderers_of_one_process_run_different_modes)."""
    import threading
    from rendernet_amd import ops
    from rendernet_amd.shader import Renderer
    from rendernet_amd.texture import TextureRenderer
    from rendernet_amd.train import Trainer
    from rendernet_amd.reconstruct import Reconstructor
    default = ops.gemm_mode_now()
    assert default == ops.WINO_GEMM
    seen = {}
    with ops.gemm_mode("f32"):
        assert ops.gemm_mode_now() == "f32"
        with ops.gemm_mode("split16"):
            assert ops.gemm_mode_now() == "split16"
            with ops.gemm_mode(None):                                   # None: whatever is in force
                assert ops.gemm_mode_now() == "split16"
            t = threading.Thread(target=lambda: seen.setdefault("other", ops.gemm_mode_now()))
            t.start(); t.join()
        assert ops.gemm_mode_now() == "f32"
        with pytest.raises(ValueError):
            with ops.gemm_mode("bf16"):
                pass
        assert ops.gemm_mode_now() == "f32"                             # a refused mode leaves the context as it was
    assert ops.gemm_mode_now() == default and seen["other"] == default   # another thread never sees this thread's context
    for cls in (Renderer, TextureRenderer, Trainer, Reconstructor):
        with pytest.raises(ValueError, match="gemm="):                  # validated before anything touches a device
            cls(gemm="bf16")
    for name in ("ops.py", "shader.py", "texture.py", "train.py", "reconstruct.py", "parallel.py"):
        src = open(os.path.join(ROOT, "rendernet_amd", name)).read()
        assert not re.search(r"(?<![A-Za-z_])WINO_GEMM\s*=(?!=)", src.replace('WINO_GEMM = os.environ.get("RN_WINO_GEMM", "split")', "")), name
    for name in ("bench.py", "RenderNet_demo.py", "RenderNet_Shader.py", "RenderNet_Texture_Face_Normal.py", "Reconstruct_RenderNet_Face.py"):
        assert not re.search(r"ops\.WINO_GEMM\s*=(?!=)", open(os.path.join(ROOT, name)).read()), name


def test_train_context_refuses_unregistered_parameters_and_skips_marked_constants():
    """Advisor finding (round 5): TrainContext.grad_if_param / ready skipped ANY tensor that was not registered, so a PReLU slope whose
    registration was forgotten silently never trained.  Constants that ride in a parameter slot are now marked (ops.mark_constant: the
    all-zero slope of tools/layer_util.py:_relu_slope); everything else must be registered or the call raises."""
    import torch
    from rendernet_amd import ops
    from rendernet_amd._lib import RenderNetHipError
    from rendernet_amd.tools.layer_util import _relu_slope
    alpha, stray = torch.zeros(8), torch.zeros(8)
    g = torch.zeros(8)
    done = []
    tc = ops.TrainContext({alpha.data_ptr(): g}, on_ready=done.append, device="cpu")
    assert tc.grad_if_param(alpha) is g and tc.grad_if_param(None) is None
    slope = _relu_slope(8, "cpu")
    assert ops.is_constant(slope) and not ops.is_constant(alpha) and float(slope.abs().max()) == 0.0
    assert tc.grad_if_param(slope) is None
    with pytest.raises(RenderNetHipError, match="mark_constant"):
        tc.grad_if_param(stray)
    tc.ready(alpha, None, slope)
    assert done == [alpha.data_ptr()]
    with pytest.raises(RenderNetHipError):
        tc.ready(stray)
    frozen = ops.TrainContext(frozen=True, device="cpu")                # inverse rendering: no parameter gradients at all
    assert frozen.grad_if_param(stray) is None and frozen.grad(stray) is None


def test_conv3d_split_rule_and_factored_input_transform(monkeypatch):
    """Two host-side facts of round 5's late changes.  (1) ops._conv3d_split: in the split modes the split 3-D kernel takes every map of at least 8 rows of
    tiles PER IMAGE (round 6: the gate no longer looks at the batch size, so a frame is routed alike alone and in a batch; round 5 asked
    for 64 rows in the launch) -- its depth segments keep the workgroups busy for few rows -- never in exact mode, and RN_CONV3D_SPLIT
    forces it either way.  (2) the factored form of F(6x6,3x3)'s B^T that the split input transforms apply
    (csrc/conv_wino_bf3.hip: bt_apply -- rows 1..6 as +/- pairs over the even and the odd inputs, 26 operations instead of 44) is the matrix
    of csrc/wino_mats.h (WinoF63::BT), exactly in float64, and its nesting A^T [(G g G^T) . (B^T d B)] A is the 3x3 correlation."""
    from rendernet_amd import ops
    monkeypatch.setattr(ops, "CONV3D_SPLIT", None)
    for mode, want in (("split", True), ("split16", True), ("f32", False)):
        monkeypatch.setattr(ops._MODE, "mode", mode, raising=False)
        assert ops._conv3d_split(1, 64, 64) is want and ops._conv3d_split(24, 64, 64) is want and ops._conv3d_split() is want
        # round 6: a per-image gate (>= 8 rows of tiles per image) -- the batch size never changes the route of a frame
        for B in (1, 3, 4, 24):
            assert ops._conv3d_split(B, 32, 32) is want and ops._conv3d_split(B, 16, 16) is want            # 16 / 8 rows per image
            assert ops._conv3d_split(B, 8, 8) is False and ops._conv3d_split(B, 14, 32) is False             # 4 / 7 rows
    monkeypatch.setattr(ops._MODE, "mode", "f32", raising=False)
    monkeypatch.setattr(ops, "CONV3D_SPLIT", True)
    assert ops._conv3d_split(1, 8, 8) is True
    monkeypatch.setattr(ops, "CONV3D_SPLIT", False)
    monkeypatch.setattr(ops._MODE, "mode", "split", raising=False)
    assert ops._conv3d_split(24, 64, 64) is False

    BT = np.array([[-1, 0, 21 / 4, 0, -21 / 4, 0, 1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                   [0, .5, .25, -2.5, -1.25, 2, 1, 0], [0, -.5, .25, 2.5, -1.25, -2, 1, 0], [0, 2, 4, -2.5, -5, .5, 1, 0],
                   [0, -2, 4, 2.5, -5, -.5, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]])
    src = open(os.path.join(ROOT, "rendernet_amd", "csrc", "wino_mats.h")).read()
    assert "{0.f, 1.f / 2.f, 1.f / 4.f, -5.f / 2.f, -5.f / 4.f, 2.f, 1.f, 0.f}" in src                   # the table above is WinoF63::BT's

    def bt_apply(d):
        o = [None] * 8
        o[0] = (d[6] - d[0]) + 5.25 * (d[2] - d[4])
        o[7] = (d[7] - d[1]) + 5.25 * (d[3] - d[5])
        e1, f1 = (d[2] + d[6]) - 4.25 * d[4], (d[1] + d[5]) - 4.25 * d[3]
        o[1], o[2] = e1 + f1, e1 - f1
        e2, f2 = (d[6] + 0.25 * d[2]) - 1.25 * d[4], (0.5 * d[1] - 2.5 * d[3]) + 2.0 * d[5]
        o[3], o[4] = e2 + f2, e2 - f2
        e3, f3 = (d[6] + 4.0 * d[2]) - 5.0 * d[4], (2.0 * d[1] - 2.5 * d[3]) + 0.5 * d[5]
        o[5], o[6] = e3 + f3, e3 - f3
        return np.stack(o)

    assert np.array_equal(bt_apply(np.eye(8)), BT)                                                         # column by column: the same matrix
    rng = np.random.default_rng(63)
    d = rng.standard_normal((8, 8))
    V = bt_apply(bt_apply(d).T).T                                                                          # B^T d B: columns, then rows
    assert np.abs(V - BT @ d @ BT.T).max() < 1e-12
    G = np.array([[-1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
                  [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]])
    AT = np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, .5, -.5, 0], [0, 1, 1, 4, 4, .25, .25, 0], [0, 1, -1, 8, -8, .125, -.125, 0],
                   [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]])
    g = rng.standard_normal((3, 3))
    y = AT @ ((G @ g @ G.T) * V) @ AT.T
    want = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(6)] for i in range(6)])
    assert np.abs(y - want).max() < 1e-9 * max(1.0, np.abs(want).max())                              # float64: the transforms' growth (~1e3) x 1e-16
