"""Pins oracle/reconstruct.py (inverse-rendering graph, TF Phong composite) on CPU, and the host logic of
rendernet_amd/reconstruct.py that needs no GPU (pretrained-key map, weight-folder loader, pose hypotheses)."""
import math
import os

import numpy as np
import torch

from oracle import io_phong as OP
from oracle import reconstruct as OR
from oracle import resample as R


def _phong_inputs(seed=0, B=2, H=6, W=5):
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.05, 0.95, (B, H, W, 3)).astype(np.float32)
    # pixels on both masks' transition bands: |img| ~ sqrt(3) - 80/255 (white), |img| ~ 80/255 (TF black)
    img[0, 0, 0] = np.float32(0.8185)
    img[0, 0, 1] = np.float32(0.8190)
    img[1, 1, 0] = (0.18, 0.18, 0.19)
    img[1, 1, 1] = (0.10, 0.30, 0.05)
    light = rng.standard_normal((B, 3)).astype(np.float32)
    col = rng.uniform(0.5, 1.0, (B, 3)).astype(np.float32)
    return img, light, col


def test_tf_phong_matches_numpy_twin_where_they_coincide():
    """tools/Phong_shading.py: the TF and NumPy shading are the same function; only the masks differ."""
    img, light, col = _phong_inputs()
    t = lambda a: torch.from_numpy(a)
    got = OR.tf_phong_composite(t(img), t(light), t(col), 0.1, 0.9, with_mask=False).numpy()
    want = OP.np_phong_composite(img, light, col, 0.1, 0.9, with_mask=False)
    assert np.abs(got - want).max() < 2e-6
    got = OR.tf_phong_shading(t(img), t(light), t(col), 0.9).numpy()
    assert np.abs(got - OP.np_phong_shading(img, light, col, 0.9)).max() < 2e-6
    # masks by hand at one pixel
    p = img[0, 2, 3].astype(np.float64)
    sig = lambda x: 1 / (1 + math.exp(-x))
    assert abs(float(OR.tf_mask(t(img))[0, 2, 3, 0]) - sig(255 * np.linalg.norm(p) - 80)) < 1e-5
    assert abs(float(OR.tf_mask_white(t(img))[0, 2, 3, 1]) - sig(255 * (math.sqrt(3) - np.linalg.norm(p)) - 80)) < 1e-5
    assert abs(float(OP.np_mask_white(img)[0, 2, 3, 0]) - sig(255 * np.linalg.norm(1 - p) - 80)) < 1e-5
    assert OR.tf_mask_white(t(img)).shape == (2, 6, 5, 3) and OR.tf_mask(t(img)).shape == (2, 6, 5, 1)
    # the white-background composite of a white pixel is white, of a black-background black pixel is white too (mask 0)
    one = np.ones((1, 1, 1, 3), np.float32) * 0.999
    out = OR.tf_phong_composite(t(one), t(light[:1]), t(col[:1]), 0.0, 1.0).numpy()
    assert np.abs(out - 1.0).max() < 1e-4


def test_tf_phong_gradient_matches_finite_differences():
    img, light, col = _phong_inputs(1)
    for black in (False, True):
        f = lambda i, l: OR.tf_phong_composite(i, l, torch.from_numpy(col).double(), 0.1, 0.9, with_black_background=black)
        i = torch.from_numpy(img).double().requires_grad_(True)
        l = torch.from_numpy(light).double().requires_grad_(True)
        wgt = torch.from_numpy(np.random.default_rng(2).standard_normal(img.shape))
        (f(i, l) * wgt).sum().backward()
        for (b, y, x, c) in [(0, 0, 0, 1), (1, 1, 0, 2), (1, 1, 1, 0), (0, 3, 2, 0)]:
            h = 1e-7
            ip, im = img.astype(np.float64).copy(), img.astype(np.float64).copy()
            ip[b, y, x, c] += h; im[b, y, x, c] -= h
            with torch.no_grad():
                fd = float(((f(torch.from_numpy(ip), l) - f(torch.from_numpy(im), l)) * wgt).sum()) / (2 * h)
            assert abs(fd - float(i.grad[b, y, x, c])) <= 1e-5 * max(1.0, abs(fd)), (black, b, y, x, c, fd, float(i.grad[b, y, x, c]))
        for (b, c) in [(0, 0), (1, 2)]:
            h = 1e-7
            lp, lm = light.astype(np.float64).copy(), light.astype(np.float64).copy()
            lp[b, c] += h; lm[b, c] -= h
            with torch.no_grad():
                fd = float(((f(i, torch.from_numpy(lp)) - f(i, torch.from_numpy(lm))) * wgt).sum()) / (2 * h)
            assert abs(fd - float(l.grad[b, c])) <= 1e-5 * max(1.0, abs(fd))


def test_light_position_and_pose_matrix_chain():
    az = torch.tensor([[0.3], [4.0]], dtype=torch.float64)
    l = OR.tf_generate_light_pos(az, 0.7, 2).numpy()
    for b in range(2):
        a = float(az[b, 0])
        assert np.allclose(l[b], [math.sin(0.7) * math.cos(a), math.sin(0.7) * math.sin(a), math.cos(0.7)], atol=1e-12)
    pose = np.array([[4.36, 0.52, 1.0], [1.2, 1.1, 0.8]])
    want = R.inverse_affine_f64(pose, 16, 32)
    p = torch.from_numpy(pose).requires_grad_(True)
    M = OR.inverse_affine_torch(p, 16, 32)
    assert np.abs(M.detach().numpy() - want).max() < 1e-10
    wgt = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 3, 4)))
    (M * wgt).sum().backward()
    for (b, c) in [(0, 0), (0, 1), (1, 2)]:
        h = 1e-6
        pp, pm = pose.copy(), pose.copy()
        pp[b, c] += h; pm[b, c] -= h
        fd = float(((R.inverse_affine_f64(pp, 16, 32) - R.inverse_affine_f64(pm, 16, 32)) * wgt.numpy()).sum()) / (2 * h)
        assert abs(fd - float(p.grad[b, c])) <= 1e-6 * max(1.0, abs(fd))


def test_resampler_autograd_wrapper_matches_its_numpy_backward():
    rng = np.random.default_rng(4)
    vox = rng.random((1, 8, 8, 8, 2)).astype(np.float32)
    M = R.inverse_affine(np.array([[4.36, 0.52, 1.0]], np.float32), 8, 16)
    M[:, :, 3] += 0.0137
    dout = rng.standard_normal((1, 16, 16, 16, 2))
    v = torch.from_numpy(vox).requires_grad_(True)
    m = torch.from_numpy(M.astype(np.float64)).requires_grad_(True)
    out = OR.resample(v, m, 16)
    assert np.array_equal(out.detach().numpy(), R.resampling_affine(vox, M, 16, "ordered"))
    (out * torch.from_numpy(dout).float()).sum().backward()
    dv, dm = R.resampling_affine_bwd(vox, M, dout.astype(np.float32).astype(np.float64), 16)
    assert np.abs(v.grad.numpy() - dv).max() <= 1e-5 * np.abs(dv).max()
    assert np.abs(m.grad.numpy() - dm).max() <= 1e-9 * np.abs(dm).max()


def _tiny_dicts(seed=77):
    from rendernet_amd import reconstruct as RC
    from rendernet_amd.texture import tiny_texture_spec
    ts, ds = tiny_texture_spec(), RC.tiny_shape_decoder_spec()
    wr, wd = RC.init_pretrained_weight_dicts(ts, ds, seed=seed, perturb=True)
    wd["g_conv4_weights"] = wd["g_conv4_weights"] * 30          # the random-init decoder emits sigmoid(~0) = 0.5 everywhere: give it structure
    return ts, ds, wr, wd


def test_shape_decoder_shapes_and_elu():
    ts, ds, wr, wd = _tiny_dicts(5)
    z = np.random.default_rng(5).standard_normal((2, ds.z_dim)).astype(np.float32)
    taps = {}
    out = OR.decoder_3d_torch(torch.from_numpy(z), wd, taps)
    assert out.shape == (2, 16, 16, 16, 1) and float(out.min()) > 0 and float(out.max()) < 1
    assert taps["gen1"].shape == (2, 4, 4, 4, 16) and taps["gen1"].min() > -1.0 and (taps["gen1"] < 0).any()   # ELU range


def test_pretrained_res_blocks_are_relu_blocks_without_alpha():
    """tools/layer_util.py:75-88, :107-121: with a weight_dict the res blocks are x + conv(relu(conv(x))) and read four keys per
    block; an `alpha` entry in the dict is never looked at.  The oracle's pretrained net therefore (a) runs on dicts WITHOUT any
    res*_alpha key, (b) gives the same output when such keys are added, (c) differs from the training graph's PReLU blocks
    (oracle/texture_net.py) whenever those slopes are non-zero, and equals it when they are zero."""
    from rendernet_amd import reconstruct as RC
    from oracle import texture_net as TN
    ts, ds, wr, wd = _tiny_dicts()
    assert not any(k.startswith("res") and k.endswith("alpha") for k in wr)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.uniform(0, 1, (1, 32, 32, 32, 5)).astype(np.float32))
    with torch.no_grad():
        img, nrm = OR.rendernet_pretrained_torch(x, wr)
        extra = dict(wr)
        for blk, n, c in (("res1", ts.n_res1, ts.c3), ("res2", ts.n_res2, ts.w_res2), ("res3", ts.n_res3, ts.w5)):
            for i in range(1, n + 1):
                extra["%s_%d_alpha" % (blk, i)] = rng.uniform(0.1, 0.25, c).astype(np.float32)
        img2, nrm2 = OR.rendernet_pretrained_torch(x, extra)
        assert torch.equal(img, img2) and torch.equal(nrm, nrm2)
        # the training graph on the same tensors: PReLU slopes 0 -> the same function; non-zero -> another one
        state0 = RC.state_from_pretrained(wr, wd, ts, ds)
        a0, n0 = TN.rendernet_texture_forward_torch(x, state0, ts.n_res1, ts.n_res2, ts.n_res3)
        assert float((a0 - img).abs().max()) <= 1e-6 and float((n0 - nrm).abs().max()) <= 1e-6
        state1 = dict(state0)
        for k in state1:
            if k.endswith("/alpha") and k.split("/")[1].split("_")[0] in ("res1", "res2", "res3"):
                state1[k] = np.full_like(state1[k], 0.2)
        a1, _ = TN.rendernet_texture_forward_torch(x, state1, ts.n_res1, ts.n_res2, ts.n_res3)
        assert float((a1 - img).abs().max()) > 1e-4


def test_inverse_rendering_gradients_match_finite_differences():
    """losses_and_grads on the tiny graph: d sum(recon_loss) / d light and d / d texture code against central
    differences of the oracle's own float32 forward (loose: float32 noise), and the per-hypothesis structure --
    hypothesis b's latents only move loss b."""
    ts, ds, wr, wd = _tiny_dicts()
    rng = np.random.default_rng(6)
    B = 2
    lat = dict(vector=rng.standard_normal((B, ds.z_dim)).astype(np.float32) * 2,
               param=np.array([[4.36, 0.52, 1.0], [4.0, 0.9, 1.0]], np.float32),
               texture=rng.standard_normal((B, ts.z_dim)).astype(np.float32),
               light=np.array([[4.2], [5.1]], np.float32))
    target = rng.uniform(0, 1, (B, 128, 128, 3)).astype(np.float32)
    M = R.inverse_affine(lat["param"], ts.size, ts.new_size)

    def run(l):
        return OR.losses_and_grads(l["vector"], l["param"], l["texture"], l["light"], target, wr, wd, M, ts.size, ts.new_size,
                                   0.26, (1.0, 1.0, 1.0), 0.0, 1.0)

    loss, grads, out = run(lat)
    assert loss.shape == (B,) and out["compos"].shape == (B, 128, 128, 3)
    assert np.allclose(out["compos"], out["img"] * out["shading"], atol=1e-6)
    assert all(np.abs(g).max() > 0 for g in grads.values()), {k: float(np.abs(g).max()) for k, g in grads.items()}
    for name, idx, h in (("light", (0, 0), 2e-2), ("light", (1, 0), 2e-2), ("texture", (0, 3), 5e-2), ("vector", (1, 2), 5e-2)):
        lp = {k: v.copy() for k, v in lat.items()}
        lm = {k: v.copy() for k, v in lat.items()}
        lp[name][idx] += h
        lm[name][idx] -= h
        fp, fm = run(lp)[0], run(lm)[0]
        fd = (fp.sum() - fm.sum()) / (2 * h)
        other = 1 - idx[0]
        assert abs(fp[other] - fm[other]) <= 1e-7 + 1e-6 * abs(fp[other])          # hypothesis b only moves loss b
        g = grads[name][idx]
        assert abs(fd - g) <= 0.08 * max(abs(fd), abs(g)) + 2e-6, (name, idx, fd, g)


def test_pretrained_keys_and_weight_folder_roundtrip(tmp_path):
    """tools/model_util.py:26-39 file naming + the keys Reconstruct_RenderNet_Face.py reads (:40-299).  The folders hold NO
    res*_alpha file (the reference's res blocks with a weight dict have no such variable) -- and load."""
    from rendernet_amd import reconstruct as RC
    from rendernet_amd.texture import tiny_texture_spec, texture_variable_shapes
    ts, ds = tiny_texture_spec(), RC.tiny_shape_decoder_spec()
    keys = [k for k, _, _ in RC.pretrained_rendernet_shapes(ts)]
    dkeys = [k for k, _, _ in RC.pretrained_decoder_shapes(ds)]
    assert len(set(keys)) == len(keys) and not any("res" in k.split("_")[0] and k.endswith("alpha") for k in keys)
    for key in ("e_tex_dc1_g_gc1_weights", "e_tex_dc1_alpha", "e_tex_conv0_conv2d_transpose_weights", "e_tex_conv2_conv3d_biases",
                "e_conv1_e_conv1_weights", "e_conv3_alpha", "res1_1_con1_3X3_weights", "res2_2_conv2_3x3_biases",
                "res3_skip_con1_3X3_weights", "e_conv4_e_conv4_weights", "e_conv4_alpha", "e_conv5_e_conv5_weights",
                "Image_e_conv6_1_e_conv6_1_weights", "Image_e_conv7_1_alpha", "Image_e_conv11_1_e_conv11_1_biases",
                "Normal_e_conv8_2_e_conv8_2_weights", "Normal_e_conv11_2_e_conv11_2_weights"):
        assert key in keys, key
    for key in ("g_zP_g_gc1_weights", "g_conv1_g_conv1_biases", "g_conv4_weights"):
        assert key in dkeys, key
    wr, wd = RC.init_pretrained_weight_dicts(ts, ds, seed=1)
    d1, d2 = tmp_path / "net", tmp_path / "dec"
    os.makedirs(d1); os.makedirs(d2)
    for k, v in wr.items():
        np.savez(os.path.join(d1, k + ".txt.npz"), v)
    for k, v in wd.items():
        np.savez(os.path.join(d2, k + ".txt.npz"), v)
    lr, ld = RC.load_weights(str(d1)), RC.load_weights(str(d2))
    assert set(lr) == set(wr) and set(ld) == set(wd) and all(np.array_equal(lr[k], wr[k]) for k in wr)
    RC.check_pretrained_weight_dicts(lr, ld, ts, ds)
    # into the TRAINING graph's names: every variable of it is produced, the res-block slopes as zeros
    km = RC.pretrained_key_map(ts, ds)
    assert sorted(km) == sorted(set(wr) | set(wd))
    state = RC.state_from_pretrained(lr, ld, ts, ds)
    names = [n for n, _, _ in texture_variable_shapes(ts)] + [n for n, _, _ in RC.shape_decoder_variable_shapes(ds)]
    assert sorted(state) == sorted(names)
    for n in names:
        if n.endswith("/alpha") and n.split("/")[1].split("_")[0] in ("res1", "res2", "res3"):
            assert not state[n].any()
    os.remove(os.path.join(d1, "e_conv4_alpha.txt.npz"))
    for fn in (lambda: RC.state_from_pretrained(RC.load_weights(str(d1)), ld, ts, ds),
               lambda: RC.check_pretrained_weight_dicts(RC.load_weights(str(d1)), ld, ts, ds)):
        try:
            fn()
            assert False, "missing tensor must raise"
        except KeyError as e:
            assert "e_conv4_alpha" in str(e)


def test_create_param_center():
    from rendernet_amd.reconstruct import create_param_center
    p = create_param_center(5, phi_mid=270, phi_range=60, theta_mid=90, theta_range=30)
    rad = math.pi / 180
    assert p.shape == (5, 3) and np.allclose(p[:, 2], 1.0)
    assert np.allclose(p[2, :2], [270 * rad, 0.0], atol=1e-6)
    assert np.allclose(p[0, :2], [240 * rad, 15 * rad], atol=1e-6) and np.allclose(p[4, :2], [300 * rad, -15 * rad], atol=1e-6)
    assert np.allclose(create_param_center(5, 10, 60, 90, 30)[0, 0], 340 * rad, atol=1e-6)     # wraps modulo 360
