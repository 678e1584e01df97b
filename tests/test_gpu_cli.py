"""End-to-end run of the reference's training script surface (`python RenderNet_Shader.py <config.json> --train`)
on a tiny synthetic data set: image tar with poses in the member names + binvox folder, two optimiser steps of the
full-size net, checkpoint written with the TF variable names.  -m gpu."""
import io
import json
import os
import shutil

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]      # every test once per multiply-stage mode (conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shader_script_trains_and_checkpoints(tmp_path, capsys):
    from PIL import Image
    import RenderNet_Shader
    from rendernet_amd.tools import utils
    models = tmp_path / "models"
    models.mkdir()
    shutil.copy(os.path.join(ROOT, "binvox", "chair.binvox"), models / "model_chair_abc_clean.binvox")
    shutil.copy(os.path.join(ROOT, "binvox", "table.binvox"), models / "model_chair_xyz_clean.binvox")
    tarp = str(tmp_path / "train.tar")
    w = utils.NpyTarWriter(tarp)
    rng = np.random.default_rng(0)
    for name in ("model_chair_abc_p250_t30_r3.3", "model_chair_xyz_p10_t100_r3.3", "model_chair_abc_p90_t60_r3.3",
                 "model_chair_xyz_p300_t45_r3.3"):
        buf = io.BytesIO()
        Image.fromarray((rng.random((512, 512)) * 255).astype(np.uint8)).save(buf, format="PNG")
        w.add_bytes(buf.getvalue(), name + ".png")
    w.close()
    cfg = {"image_path": tarp, "image_path_valid": "", "model_path": str(models), "is_greyscale": "True", "gpu": 0,
           "batch_size": 2, "max_epochs": 1, "batches_chunk": 1, "threshold": 0.1, "e_eta": 1e-5, "keep_prob": 1.0,
           "decay_steps": 100000, "trained_model_name": "3d2d_renderer", "sample_save": str(tmp_path / "out"),
           "checkpoint_secs": 7200}
    cfgp = str(tmp_path / "config.json")
    json.dump(cfg, open(cfgp, "w"))
    RenderNet_Shader.main([cfgp, "--train", "--max-steps", "2"])
    out = capsys.readouterr().out
    assert "Step 1 Loss" in out and "Step 2 Loss" in out
    losses = [float(l.split("Loss")[1]) for l in out.splitlines() if l.startswith("Step")]
    assert all(np.isfinite(losses)) and losses[0] > 0
    ck = np.load(os.path.join(cfg["sample_save"], "3d2d_renderer.npz"))
    assert "encoder/res2_4/con1_3X3/weights" in ck.files and ck["encoder/e_conv7/e_conv7/weights"].shape == (4, 4, 128, 256)
    names = [k for k in ck.files if not k.startswith("__")]
    assert len(names) == 166                                      # every variable of the Phong-shader graph
    # the checkpoint carries the optimiser state (Adam moments, global_step, epoch) like the reference's Supervisor
    assert int(ck["__global_step__"]) == 2 and int(ck["__epoch__"]) == 1 and ck["__adam_m__"].shape == ck["__adam_v__"].shape
    assert float(np.abs(ck["__adam_v__"]).max()) > 0
    assert not os.path.exists(os.path.join(cfg["sample_save"], "3d2d_renderer.npz.tmp.npz"))
    # a restart resumes: epoch counter past max_epochs -> no further steps, weights untouched
    RenderNet_Shader.main([cfgp, "--train", "--max-steps", "5"])
    assert "Step" not in capsys.readouterr().out
    # and a world size that does not divide the batch is refused up front (an empty shard would dead-lock the all-reduces)
    os.environ["WORLD_SIZE"], os.environ["RANK"] = "3", "0"
    try:
        with pytest.raises(SystemExit, match="not a multiple"):
            RenderNet_Shader.main([cfgp, "--train"])
    finally:
        del os.environ["WORLD_SIZE"], os.environ["RANK"]


def test_texture_script_trains_checkpoints_and_renders(tmp_path, capsys):
    """`python RenderNet_Texture_Face_Normal.py <config.json> [--train]` on a four-image synthetic face set: the loop of
    the reference script (:196-334) with its file naming, the resumable checkpoint, then the render mode."""
    from PIL import Image
    import RenderNet_Texture_Face_Normal as script
    from rendernet_amd.tools import utils
    models, tex, nrm = tmp_path / "models", tmp_path / "beta", tmp_path / "normals"
    for d in (models, tex, nrm):
        d.mkdir()
    rng = np.random.default_rng(1)
    for ident, src in (("007", "suzanne"), ("012", "teapot")):
        shutil.copy(os.path.join(ROOT, "binvox", src + ".binvox"), models / ("faceply%s.binvox" % ident))
        np.save(tex / ("beta%s.npy" % ident), rng.standard_normal(199).astype(np.float32))
    tarp = str(tmp_path / "train.tar")
    w = utils.NpyTarWriter(tarp)
    for name in ("faceply007_p250_t30_r3.3", "faceply012_p10_t100_r3.3", "faceply007_p90_t60_r3.3", "faceply012_p300_t45_r3.3"):
        buf = io.BytesIO()
        Image.fromarray((rng.random((512, 512, 3)) * 255).astype(np.uint8)).save(buf, format="PNG")
        w.add_bytes(buf.getvalue(), name + ".png")
        Image.fromarray((rng.random((512, 512, 3)) * 255).astype(np.uint8)).save(nrm / (name + ".png"))
    w.close()
    cfg = {"image_path": tarp, "image_path_valid": tarp, "normal_path": str(nrm), "texture_path": str(tex),
           "model_path": str(models), "gpu": 0, "batch_size": 2, "max_epochs": 1, "batches_chunk": 1, "threshold": 0.1,
           "e_eta": 1e-5, "keep_prob": 0.75, "decay_steps": 100000, "trained_model_name": "3d2d_renderer",
           "sample_save": str(tmp_path / "out"), "checkpoint_secs": 7200}
    cfgp = str(tmp_path / "config.json")
    json.dump(cfg, open(cfgp, "w"))
    script.main([cfgp, "--train", "--max-steps", "2"])
    out = capsys.readouterr().out
    losses = [float(l.split("Loss")[1]) for l in out.splitlines() if l.startswith("Step")]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[0] > 0
    files = set(os.listdir(cfg["sample_save"]))
    assert {"config.json", "3d2d_renderer.npz", "L1 All.txt.npz"} <= files
    assert sum(f.startswith("VALID_faceply") and "_pred_normal_0" in f for f in files) == 1      # :322-327 naming
    ck = np.load(os.path.join(cfg["sample_save"], "3d2d_renderer.npz"))
    assert int(ck["__global_step__"]) == 2 and any(k.startswith("texture_encoder/") for k in ck.files)
    script.main([cfgp])                                            # render mode: one image + one normal map per model
    files = set(os.listdir(cfg["sample_save"]))
    assert {"VALID_faceply007_pred.png", "VALID_faceply007_pred_normal.png", "VALID_faceply012_pred.png"} <= files
    img = np.asarray(Image.open(os.path.join(cfg["sample_save"], "VALID_faceply007_pred.png")))
    assert img.shape == (512, 512, 3)


def test_demo_cli_renders_and_names_files_like_the_reference(tmp_path):
    """`python RenderNet_demo.py --voxel_path ... --render_dir ...` (RenderNet_demo.py:72-137): one PNG named
    `000_<model>_pose_<az>_<el>_<r>_light_<laz>_<lel>.png`, 512x512 RGB; and the Phong composite kernel against
    the NumPy restatement of tools/Phong_shading.py:202-228."""
    import torch
    from PIL import Image
    import RenderNet_demo
    from oracle import io_phong as OP
    from rendernet_amd.tools import Phong_shading
    out = tmp_path / "render"
    RenderNet_demo.main(["--voxel_path", os.path.join(ROOT, "binvox", "chair.binvox"), "--render_dir", str(out),
                         "--azimuth", "250", "--elevation", "60", "--radius", "3.3"])
    files = sorted(os.listdir(out))
    assert files == ["000_chair_pose_250.000000_60.000000_3.300000_light_250.000000_60.000000.png"]
    img = np.asarray(Image.open(out / files[0]))
    assert img.shape == (512, 512, 3) and img.dtype == np.uint8
    # composite parity on random normal maps
    rng = np.random.default_rng(0)
    normals = rng.random((1, 64, 64, 3)).astype(np.float32)
    light = Phong_shading.generate_light_pos(60, 250)
    got = Phong_shading.np_phong_composite(torch.as_tensor(normals).cuda(), light, RenderNet_demo.LIGHT_COL,
                                           RenderNet_demo.AMBIENT_IN, RenderNet_demo.K_DIFFUSE).cpu().numpy()
    want = OP.np_phong_composite(normals, light, RenderNet_demo.LIGHT_COL, RenderNet_demo.AMBIENT_IN, RenderNet_demo.K_DIFFUSE)
    assert np.abs(got - want).max() <= 1e-5


def test_reconstruct_script_runs_and_writes_the_reference_outputs(tmp_path, capsys):
    """`python Reconstruct_RenderNet_Face.py <config.json>` (config_reconstruction_RenderNet.json keys): shaded target,
    two latent-descent steps of the five hypotheses at full size, per-hypothesis dumps with the reference's file names."""
    from PIL import Image
    import Reconstruct_RenderNet_Face
    from rendernet_amd.tools import binvox_rw
    rng = np.random.default_rng(0)
    for name in ("albedo.png", "normal.png"):
        Image.fromarray((rng.random((512, 512, 3)) * 255).astype(np.uint8)).save(str(tmp_path / name))
    cfg = {"target_albedo": str(tmp_path / "albedo.png"), "target_normal": str(tmp_path / "normal.png"),
           "target_azimuth_light": 294, "target_elevation_light": 105, "weight_dir": str(tmp_path / "nope"),
           "weight_dir_decoder": str(tmp_path / "nope2"), "gpu": 0, "batch_size": 5, "z_dim": 200, "inner_step": 2,
           "max_epochs": 1, "threshold": 0.1, "shape_eta": 0.8, "pose_eta": 0.01, "tex_eta": 0.8, "light_eta": 0.4,
           "keep_prob": 1.0, "decay_steps": 90000, "trained_model_name": "RenderNet_recon", "sample_save": str(tmp_path / "out"),
           "checkpoint_secs": 7200}
    cfgp = str(tmp_path / "config.json")
    json.dump(cfg, open(cfgp, "w"))
    Reconstruct_RenderNet_Face.main([cfgp, "--max-steps", "2"])
    out = capsys.readouterr().out
    assert "BEST LOSS" in out and "BEST PARAM" in out
    files = sorted(os.listdir(cfg["sample_save"]))
    assert "shaded_target.png" in files and "shading.png" in files and "config.json" in files and "2_loss_.txt.npz" in files
    jpgs = [f for f in files if f.endswith(".jpg")]
    assert len(jpgs) == 5 and all(f.split("_")[1] == "2" for f in jpgs)
    vox = [f for f in files if f.endswith(".binvox")]
    assert len(vox) == 5
    with open(os.path.join(cfg["sample_save"], vox[0]), "rb") as fh:
        assert binvox_rw.read_as_3d_array(fh).data.shape == (64, 64, 64)
    assert np.load(os.path.join(cfg["sample_save"], "2_loss_.txt.npz"))["arr_0"].shape == (5,)


def test_demo_rotate_renders_72_numbered_frames_equal_to_single_pose_runs(tmp_path, gemm_mode):
    """`RenderNet_demo.py --rotate True` (RenderNet_demo.py:130-137): 72 files numbered 000..071, azimuth 0..355 in 5
    degree steps, rendered in batches -- pixel-identical to 72 runs of the single-pose path in the exact-fp32 mode AND in the default
    bf16x3 mode (round 6: every route gate is a per-image quantity, so a frame takes the same kernels alone and in a batch of 24) --
    plus the optional GIF.  split16 scales every tensor by the maximum over the whole BATCH: there frames agree to fp32 rounding,
    i.e. at most one grey level on isolated pixels of the 8-bit PNG."""
    from PIL import Image
    import RenderNet_demo
    vox = os.path.join(ROOT, "binvox", "teapot.binvox")
    rot = tmp_path / "rot"
    gif = str(tmp_path / "turn.gif")
    RenderNet_demo.main(["--voxel_path", vox, "--render_dir", str(rot), "--rotate", "True", "--elevation", "40", "--radius", "3.0",
                         "--batch", "24", "--gif", gif])
    files = sorted(os.listdir(rot))
    assert len(files) == 72
    for i, f in enumerate(files):
        assert f == "%03d_teapot_pose_%f_%f_%f_light_%f_%f.png" % (i, 5.0 * i, 40.0, 3.0, 250.0, 60.0)
    with Image.open(gif) as g:
        assert g.n_frames == 72 and g.size == (512, 512)
    for i in (0, 23, 24, 50, 71):                       # both sides of a batch boundary, first and last
        one = tmp_path / ("one%d" % i)
        RenderNet_demo.main(["--voxel_path", vox, "--render_dir", str(one), "--azimuth", str(5.0 * i), "--elevation", "40",
                             "--radius", "3.0"])
        a = np.asarray(Image.open(rot / files[i]))
        b = np.asarray(Image.open(one / os.listdir(one)[0]))
        if gemm_mode in ("f32", "split"):
            assert np.array_equal(a, b), "frame %d of the rotation differs from the single-pose render" % i
        else:
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            assert d.max() <= 1 and (d > 0).mean() <= 1e-3, "frame %d: max diff %d, %.2g of the pixels differ" % (i, d.max(), (d > 0).mean())
