"""BASELINE config 3: texture + normal face renderer (RenderNet_Texture_Face_Normal.py) on the HIP path
vs the CPU oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import demo_pose
from oracle import texture_net as OT

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]      # every test once per multiply-stage mode (conftest.py)
TAP_RTOL = 2e-4


def _cmp(got, want, name, rtol=TAP_RTOL):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err, ref = np.abs(got - want).max(), np.abs(want).max()
    assert err <= rtol * ref + 1e-6, "%s: max err %g vs |ref| %g" % (name, err, ref)


def test_tiny_texture_net_taps():
    from rendernet_amd.texture import TextureRenderer, tiny_texture_spec, init_texture_weights
    spec = tiny_texture_spec()
    w = init_texture_weights(spec, seed=77, perturb=True)
    rng = np.random.default_rng(3)
    vox = (rng.random((2, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    z = rng.standard_normal((2, spec.z_dim)).astype(np.float32)
    poses = np.stack([demo_pose(250, 60, 3.3), demo_pose(100, 40, 3.0)])
    otaps = {}
    want_img, want_nrm = OT.render_texture(vox, z, poses, w, 16, 32, spec.tex_res, (spec.n_res1, spec.n_res2, spec.n_res3), otaps)
    r = TextureRenderer(spec, w)
    taps = {}
    img, nrm = r.render(vox, z, poses, taps=taps)
    _cmp(taps["texture_decoded"], otaps["texture_decoded"], "texture_decoded")
    d = np.abs(taps["net_in"].cpu().numpy() - otaps["net_in"])
    assert (d > 2e-4 * max(1.0, np.abs(otaps["net_in"]).max())).mean() <= 1e-3     # resampler border flips (see test_gpu_resample)
    # run the HIP net on the oracle's net input so the taps compare the net itself
    from rendernet_amd import variables as V
    from rendernet_amd.texture import RenderNetTexture
    V.set_default_store(r.store)
    taps2 = {}
    img2, nrm2 = RenderNetTexture(torch.as_tensor(otaps["net_in"]).cuda(), spec=spec, taps=taps2)
    for name in ("enc1", "enc2", "enc3", "enc3_skip", "enc4", "enc4_skip", "enc5", "enc5_skip"):
        _cmp(taps2[name], otaps[name], name)
    assert np.abs(img2.cpu().numpy() - want_img).max() <= 1e-3
    assert np.abs(nrm2.cpu().numpy() - want_nrm).max() <= 1e-3
    for head, t in (("image", img2), ("normal", nrm2)):
        o = t.double()
        lg = torch.log(o / (1 - o)).float().cpu().numpy()
        want = otaps[head + "_logits"]
        assert np.abs(lg - want).max() <= 1e-3 * np.abs(want).max() + 1e-5
    assert img.shape == (2, 128, 128, 3) and nrm.shape == (2, 128, 128, 3)


def test_full_size_texture_frame(fixtures_vox):
    """Reference-size texture net (64^3 + 199-d code -> two 512x512x3 maps), one frame vs the oracle."""
    from rendernet_amd.texture import TextureRenderer, TextureSpec, init_texture_weights
    spec = TextureSpec().check()
    w = init_texture_weights(spec, seed=1234, perturb=True)
    rng = np.random.default_rng(7)
    z = rng.standard_normal((1, 199)).astype(np.float32)          # Reconstruct_RenderNet_Face.py:464 uses randn
    vox, pose = fixtures_vox[3:4], demo_pose()[None]
    want_img, want_nrm = OT.render_texture(vox, z, pose, w)
    img, nrm = TextureRenderer(spec, w).render(vox, z, pose)
    assert np.abs(img.cpu().numpy() - want_img).max() <= 1e-3
    assert np.abs(nrm.cpu().numpy() - want_nrm).max() <= 1e-3


def test_fully_connected_backward():
    """dx, dw, dbias, dalpha of fully_connected + PReLU (tools/layer_util.py:311-343, :27-45) vs torch-CPU autograd."""
    from oracle import layers as OL
    from rendernet_amd import ops
    rng = np.random.default_rng(1)
    B, fin, fout = 5, 37, 1024
    x, w = rng.standard_normal((B, fin)).astype(np.float32), (rng.standard_normal((fin, fout)) * 0.1).astype(np.float32)
    b, al = (rng.standard_normal(fout) * 0.1).astype(np.float32), rng.uniform(0.05, 0.3, fout).astype(np.float32)
    dy = rng.standard_normal((B, fout)).astype(np.float32)
    xt, wt, bt, at = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b, al))
    OL.prelu(OL.fully_connected(xt, wt, bt), at).backward(torch.from_numpy(dy))
    xd = torch.as_tensor(x).cuda().requires_grad_(True)
    wd, bd, ad = torch.as_tensor(w).cuda(), torch.as_tensor(b).cuda(), torch.as_tensor(al).cuda()
    grads = {t.data_ptr(): torch.zeros_like(t) for t in (wd, bd, ad)}
    tc = ops.TrainContext(grads)
    with ops.training(tc):
        y = ops.fully_connected(xd, wd, bd, ad)
    y.backward(torch.as_tensor(dy).cuda())
    _cmp(xd.grad, xt.grad.numpy(), "dx")
    _cmp(grads[wd.data_ptr()], wt.grad.numpy(), "dw")
    _cmp(grads[bd.data_ptr()], bt.grad.numpy(), "dbias")
    _cmp(grads[ad.data_ptr()], at.grad.numpy(), "dalpha")


def test_texture_net_training_step_matches_oracle():
    """One training step of the texture + normal net (RenderNet_Texture_Face_Normal.py:152-186): loss and every
    parameter gradient -- including the texture decoder's, which flow back THROUGH the resampler -- against torch-CPU
    autograd over the oracle graph; then two optimiser steps run and move the weights."""
    from oracle import texture_train as TT
    from rendernet_amd import ops
    from rendernet_amd.texture import tiny_texture_spec, init_texture_weights
    from rendernet_amd.train import TextureTrainer
    spec = tiny_texture_spec()
    w = init_texture_weights(spec, seed=77, perturb=True)
    rng = np.random.default_rng(3)
    B, patch, start = 2, 16, (5, 9)
    vox = (rng.random((B, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    z = rng.standard_normal((B, spec.z_dim)).astype(np.float32)
    poses = np.stack([demo_pose(250, 60, 3.3), demo_pose(100, 40, 3.0)])
    images = rng.uniform(0, 1, (B, 128, 128, 3)).astype(np.float32)
    normals = rng.uniform(0, 1, (B, 128, 128, 3)).astype(np.float32)
    tr = TextureTrainer(spec, w, device="cuda:0", e_eta=1e-3)
    img, nrm, (r, c, p, _) = tr.forward(vox, z, poses, patch, start)
    crop = lambda t: torch.as_tensor(t[:, 4 * r:4 * (r + p), 4 * c:4 * (c + p)]).cuda()
    tr.loss_and_backward(img, nrm, crop(images), crop(normals), B)
    M = ops.pose_to_affine(torch.as_tensor(poses).cuda(), spec.size, spec.new_size).cpu().numpy()
    loss, grads, (oimg, onrm) = TT.loss_and_grads(vox, z, M, images, normals, w, start, patch, spec.size, spec.new_size,
                                                   spec.tex_res, (spec.n_res1, spec.n_res2, spec.n_res3), spec.tex_c0)
    assert np.abs(img.detach().cpu().numpy() - oimg).max() <= 1e-3 and np.abs(nrm.detach().cpu().numpy() - onrm).max() <= 1e-3
    got_loss = float(tr.loss_buf.item())
    assert abs(got_loss - loss) <= 1e-4 * abs(loss), (got_loss, loss)
    worst = 0.0
    for name, g in grads.items():
        got = tr.grad_views[name].cpu().numpy()
        ref, err = np.abs(g).max(), np.abs(got - g).max()
        worst = max(worst, err / (ref + 1e-20))
        assert err <= 2e-3 * ref + 1e-9, "%s: grad err %g vs max|ref| %g" % (name, err, ref)
    assert any(k.startswith("texture_encoder/") and np.abs(v).max() > 0 for k, v in grads.items())
    print("worst relative gradient error %.3g over %d tensors" % (worst, len(grads)))
    tr.apply_gradients()
    l2 = float(tr.step(vox, z, poses, images, normals, patch_size=patch, start_point=start).item())
    assert np.isfinite(l2) and tr.global_step == 2
    sd = tr.state_dict()
    assert max(np.abs(sd[k] - w[k]).max() for k in w) > 1e-4
