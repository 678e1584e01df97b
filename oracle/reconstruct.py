"""CPU restatement of the inverse-rendering graph of Reconstruct_RenderNet_Face.py:31-72, :334-404 and of the
TensorFlow Phong composite tools/Phong_shading.py:23-130 -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU autograd stands in for tf.gradients.  Two pieces need care:
  * the resampler: value from the NumPy oracle (oracle/resample.py, "ordered" coordinates), gradient w.r.t. the voxel
    values AND the matrix from its float64 backward (`resampling_affine_bwd`, itself pinned by finite differences and
    the adjoint identity in tests/test_oracle_resample_bwd.py);
  * pose -> matrix: the float64 chain of `inverse_affine_f64` rebuilt with torch ops, so d matrix / d pose is autograd's.
TF gradient conventions restated: clip_by_value and maximum pass the gradient inside their range (torch.clamp and
torch.clamp_min agree except on the measure-zero ties), tf.norm -> x/|x|.
"""
import math
import numpy as np
import torch

from . import layers as L
from . import resample as R
from . import texture_net as TN
from .texture_train import to_image_layout


# ---------------------------------------------------------------------------------------------
# tools/Phong_shading.py, TensorFlow half
# ---------------------------------------------------------------------------------------------
def tf_mask(images_in):
    """:23-32."""
    return torch.sigmoid(255. * torch.linalg.vector_norm(images_in, dim=3, keepdim=True) - 80)


def tf_mask_white(images_in):
    """:34-44: ones_like(images)*sqrt(3) - norm, so the mask has the image's three (equal) channels."""
    m = torch.ones_like(images_in) * math.sqrt(3) - torch.linalg.vector_norm(images_in, dim=3, keepdim=True)
    return torch.sigmoid(255. * m - 80)


def tf_phong_shading(images_in, light_dir, light_col, k_diffuse):
    """:46-86."""
    shp = images_in.shape
    n = (images_in - 0.5).reshape(-1, 3)
    n = n / torch.linalg.vector_norm(n, dim=1).unsqueeze(1)
    light_dir = light_dir / torch.linalg.vector_norm(light_dir, dim=1).reshape(-1, 1)
    npix = shp[1] * shp[2]
    ld = torch.repeat_interleave(light_dir, npix, dim=0)                 # tf_repeat(light_dir, [H*W, 1])
    lc = torch.repeat_interleave(light_col, npix, dim=0)
    d = torch.clamp_min(torch.sum(n * ld, dim=1, keepdim=True), 0.)
    d = k_diffuse * (d.repeat(1, 3) * lc)
    return torch.clamp(d.reshape(shp), 0., 1.)


def tf_phong_composite(images_in, light_dir, light_col, ambient_in, k_diffuse, with_black_background=False, with_mask=True):
    """:88-111."""
    diffuse = tf_phong_shading(images_in, light_dir, light_col, k_diffuse)
    if with_mask:
        mask = tf_mask(images_in) if with_black_background else tf_mask_white(images_in)
        compos = mask * (ambient_in + diffuse) + (1 - mask)
    else:
        compos = ambient_in + diffuse
    return torch.clamp(compos, 0., 1.)


def tf_generate_light_pos(batch_light_azimuth, light_elevation, batch_size):
    """:113-130."""
    el = torch.full((batch_size, 1), float(light_elevation), dtype=batch_light_azimuth.dtype)
    az = batch_light_azimuth.reshape(batch_size, 1)
    return torch.cat((torch.sin(el) * torch.cos(az), torch.sin(el) * torch.sin(az), torch.cos(el)), dim=1)


# ---------------------------------------------------------------------------------------------
# shape decoder, Reconstruct_RenderNet_Face.py:31-72
# ---------------------------------------------------------------------------------------------
def decoder_3d_torch(z, w, base=4, chans=(256, 128, 64, 32, 16), taps=None):
    x = L.fully_connected(z, w["g_zP/g_gc1/weights"], w["g_zP/g_gc1/biases"])
    x = x.reshape(x.shape[0], base, base, base, chans[0])
    for i in range(1, len(chans)):
        p = "g_conv%d/g_conv%d/" % (i, i)
        x = torch.nn.functional.elu(L.conv3d_transpose(x, w[p + "weights"], w[p + "biases"], (2, 2, 2)))
        if taps is not None:
            taps["gen%d" % i] = x.detach().numpy().copy()
    p = "g_conv%d/" % len(chans)
    return torch.sigmoid(L.conv3d_transpose(x, w[p + "weights"], w[p + "biases"], (1, 1, 1)))


# ---------------------------------------------------------------------------------------------
# differentiable resampler (voxels and matrix) and pose -> matrix
# ---------------------------------------------------------------------------------------------
class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vox, M, new_size):
        v, m = vox.detach().numpy(), M.detach().numpy().astype(np.float32)
        ctx.save_for_backward(vox, M)
        ctx.new_size = new_size
        return torch.from_numpy(R.resampling_affine(v, m, new_size, "ordered"))

    @staticmethod
    def backward(ctx, dout):
        vox, M = ctx.saved_tensors
        dvox, dM = R.resampling_affine_bwd(vox.detach().numpy(), M.detach().numpy().astype(np.float32),
                                           dout.numpy().astype(np.float64), ctx.new_size)
        return torch.from_numpy(dvox.astype(np.float32)), torch.from_numpy(dM).to(M.dtype), None


def resample(vox, M, new_size):
    """[B,S,S,S,C] voxels (float32 tensor), [B,3,4] matrices -> raw [B,N,N,N,C] grid, differentiable in both."""
    return _Resample.apply(vox, M, new_size)


def inverse_affine_torch(pose, size, new_size):
    """oracle/resample.py::inverse_affine_f64 (tools/resampling_voxel_grid.py:526-602) with torch float64 ops."""
    pose = pose.double()
    out = []
    for b in range(pose.shape[0]):
        az, el, s = pose[b, 0] - math.pi * 0.5, pose[b, 1], pose[b, 2]
        ca, sa, ce, se = torch.cos(az), torch.sin(az), torch.cos(el), torch.sin(el)
        zero, one = torch.zeros((), dtype=torch.float64), torch.ones((), dtype=torch.float64)
        rot_y = torch.stack([torch.stack([ca, zero, -sa, zero]), torch.stack([zero, one, zero, zero]),
                             torch.stack([sa, zero, ca, zero]), torch.stack([zero, zero, zero, one])])
        rot_z = torch.stack([torch.stack([ce, se, zero, zero]), torch.stack([-se, ce, zero, zero]),
                             torch.stack([zero, zero, one, zero]), torch.stack([zero, zero, zero, one])])
        sc = torch.diag(torch.stack([s, s, s, one]))
        T = torch.eye(4, dtype=torch.float64); T[:3, 3] = -size * 0.5
        Tn = torch.eye(4, dtype=torch.float64); Tn[:3, 3] = new_size * 0.5
        total = Tn @ sc @ (rot_z @ rot_y) @ T
        out.append(torch.linalg.inv(total)[:3])
    return torch.stack(out)


# ---------------------------------------------------------------------------------------------
# the graph (:356-383) and tf.gradients(recon_loss, latents) (:404)
# ---------------------------------------------------------------------------------------------
def losses_and_grads(vector, param, texture, light, target, weights, M_inv, size, new_size, tex_res, n_res, dec_base, dec_chans,
                     light_elevation, light_col, ambient, k_diffuse, c0=4, taps=None):
    """Latents as ndarrays; `M_inv` [B,3,4] = the matrices the HIP path resampled with (values used as they are; the
    gradient flows through `inverse_affine_torch(param)`).  Returns (recon_loss [B], {latent: gradient}, tensors dict)."""
    w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
    lat = {"vector": vector, "param": param, "texture": texture, "light": light}
    lat = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).requires_grad_(True) for k, v in lat.items()}
    B = lat["param"].shape[0]
    shape = decoder_3d_torch(lat["vector"], w, dec_base, dec_chans, taps)
    tex = TN.decoder_texture_torch(lat["texture"], w, tex_res, c0)
    M_t = inverse_affine_torch(lat["param"], size, new_size)
    M = torch.from_numpy(np.asarray(M_inv, np.float64)) + (M_t - M_t.detach())        # value: the HIP path's matrices
    geo = to_image_layout(resample(shape, M, new_size))
    tex_rot = to_image_layout(resample(tex, M, new_size))
    net_in = torch.cat([geo, tex_rot], dim=4).contiguous()
    img, nrm = TN.rendernet_texture_forward_torch(net_in, w, n_res[0], n_res[1], n_res[2])
    light_dir = tf_generate_light_pos(lat["light"], light_elevation, B)
    lc = torch.from_numpy(np.tile(np.asarray(light_col, np.float32).reshape(1, 3), (B, 1)))
    shading = tf_phong_composite(nrm, light_dir, lc, ambient, k_diffuse, with_mask=True)
    compos = img * shading
    tgt = torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32))
    loss = torch.mean((tgt - compos) ** 2, dim=(1, 2, 3))
    loss.sum().backward()
    grads = {k: v.grad.numpy().copy() for k, v in lat.items()}
    out = {"compos": compos.detach().numpy(), "img": img.detach().numpy(), "normal": nrm.detach().numpy(),
           "shape": shape.detach().numpy(), "shading": shading.detach().numpy(), "net_in": net_in.detach().numpy()}
    return loss.detach().numpy(), grads, out
