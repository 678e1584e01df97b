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
# the three pretrained networks, Reconstruct_RenderNet_Face.py:31-302.  `wd` = a weight dict as tools/model_util.py:26-39
# loads it: {key: array}, keys spelled as the reference spells them at each call site.
# ---------------------------------------------------------------------------------------------
def _w(wd, key):
    v = wd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def decoder_3d_torch(z, wd, taps=None):
    """:31-72: FC -> reshape [B,4,4,4,256] -> 4 x elu(conv3d_transpose k4 s2) (:47-64) -> sigmoid(conv3d_transpose k4 s1)
    (:66-70, scope g_conv5 without the doubled name: keys g_conv5_weights / _biases).  Layer count and widths from the dict."""
    n = 1
    while "g_conv%d_g_conv%d_weights" % (n, n) in wd:
        n += 1
    x = L.fully_connected(z, _w(wd, "g_zP_g_gc1_weights"), _w(wd, "g_zP_g_gc1_biases"))                       # :41-44
    c0 = _w(wd, "g_conv1_g_conv1_weights").shape[4]
    base = int(round((x.shape[1] // c0) ** (1.0 / 3.0)))
    x = x.reshape(x.shape[0], base, base, base, c0)                                                           # :45
    for i in range(1, n):
        k = "g_conv%d_g_conv%d_" % (i, i)
        x = torch.nn.functional.elu(L.conv3d_transpose(x, _w(wd, k + "weights"), _w(wd, k + "biases"), (2, 2, 2)))
        if taps is not None:
            taps["gen%d" % i] = x.detach().numpy().copy()
    return torch.sigmoid(L.conv3d_transpose(x, _w(wd, "g_conv%d_weights" % n), _w(wd, "g_conv%d_biases" % n), (1, 1, 1)))


def texture_decoder_pretrained_torch(z, wd):
    """:74-112: prelu(FC) -> [B,32,32,32,4] -> prelu(conv3d_transpose k4 s1) -> prelu(conv3d_transpose k4 s2) -> prelu(conv3d k4 s1),
    slopes loaded (`alpha=weight_dict[...]`).  The FC's width is the loaded matrix's (tools/layer_util.py:333)."""
    x = L.prelu(L.fully_connected(z, _w(wd, "e_tex_dc1_g_gc1_weights"), _w(wd, "e_tex_dc1_g_gc1_biases")), _w(wd, "e_tex_dc1_alpha"))
    c0 = _w(wd, "e_tex_conv0_conv2d_transpose_weights").shape[4]
    res = int(round((x.shape[1] // c0) ** (1.0 / 3.0)))
    x = x.reshape(x.shape[0], res, res, res, c0)                                                              # :90
    x = L.prelu(L.conv3d_transpose(x, _w(wd, "e_tex_conv0_conv2d_transpose_weights"), _w(wd, "e_tex_conv0_conv2d_transpose_biases"),
                                   (1, 1, 1)), _w(wd, "e_tex_conv0_alpha"))                                   # :91-95
    x = L.prelu(L.conv3d_transpose(x, _w(wd, "e_tex_conv1_conv2d_transpose_weights"), _w(wd, "e_tex_conv1_conv2d_transpose_biases"),
                                   (2, 2, 2)), _w(wd, "e_tex_conv1_alpha"))                                   # :96-101
    return L.prelu(L.conv3d(x, _w(wd, "e_tex_conv2_conv3d_weights"), _w(wd, "e_tex_conv2_conv3d_biases"), (1, 1, 1)),
                   _w(wd, "e_tex_conv2_alpha"))                                                               # :102-107


def res_block_3d_pretrained(x, wd, scope):
    """tools/layer_util.py:74-88, the weight_dict branch: tf.nn.relu (NOT prelu: no alpha exists), keys scope + '_con1_3X3_weights' ..."""
    net = torch.relu(L.conv3d(x, _w(wd, scope + "_con1_3X3_weights"), _w(wd, scope + "_con1_3X3_biases")))
    net = L.conv3d(net, _w(wd, scope + "_conv2_3x3_weights"), _w(wd, scope + "_conv2_3x3_biases"))
    return net + x


def res_block_2d_pretrained(x, wd, scope):
    """tools/layer_util.py:106-121, the weight_dict branch: hand-rolled conv2d + tf.nn.relu, no alpha."""
    net = torch.relu(L.conv2d(x, _w(wd, scope + "_con1_3X3_weights"), _w(wd, scope + "_con1_3X3_biases")))
    net = L.conv2d(net, _w(wd, scope + "_conv2_3x3_weights"), _w(wd, scope + "_conv2_3x3_biases"))
    return net + x


def rendernet_pretrained_torch(x, wd, taps=None):
    """RenderNet_pretrained, :113-302, at prob = 1.0 (every tf.nn.dropout the identity).  x [B,H,W,D,5] -> (albedo, normal)."""
    def tap(n, t):
        if taps is not None:
            taps[n] = t.detach().numpy().copy()
        return t

    for name, s in (("e_conv1", (2, 2, 2)), ("e_conv2", (1, 1, 2)), ("e_conv3", (1, 1, 1))):                 # :128-147
        k = "%s_%s_" % (name, name)
        x = L.prelu(L.conv3d(x, _w(wd, k + "weights"), _w(wd, k + "biases"), s), _w(wd, name + "_alpha"))
        tap("enc" + name[-1], x)
    shortcut = x                                                                                              # :149
    k = 1
    while "res1_%d_con1_3X3_weights" % k in wd:                                                               # :150-159
        x = res_block_3d_pretrained(x, wd, "res1_%d" % k)
        k += 1
    enc3_skip = tap("enc3_skip", L.conv3d(x, _w(wd, "res1_skip_con1_3X3_weights"), _w(wd, "res1_skip_con1_3X3_biases")) + shortcut)
    B, H, W, D, C = enc3_skip.shape
    enc3_2d = enc3_skip.reshape(B, H, W, D * C)                                                               # :172: f = d*C + c
    enc4 = tap("enc4", L.prelu(L.conv2d(enc3_2d, _w(wd, "e_conv4_e_conv4_weights"), _w(wd, "e_conv4_e_conv4_biases")),
                               _w(wd, "e_conv4_alpha")))                                                      # :174-180
    x = enc4
    k = 1
    while "res2_%d_con1_3X3_weights" % k in wd:                                                               # :183-192
        x = res_block_2d_pretrained(x, wd, "res2_%d" % k)
        k += 1
    enc4_skip = tap("enc4_skip", L.conv2d(x, _w(wd, "res2_skip_con1_3X3_weights"), _w(wd, "res2_skip_con1_3X3_biases")) + enc4)
    enc5 = tap("enc5", L.prelu(L.conv2d(enc4_skip, _w(wd, "e_conv5_e_conv5_weights"), _w(wd, "e_conv5_e_conv5_biases")),
                               _w(wd, "e_conv5_alpha")))                                                      # :202-208
    x = enc5
    k = 1
    while "res3_%d_con1_3X3_weights" % k in wd:                                                               # :213-217
        x = res_block_2d_pretrained(x, wd, "res3_%d" % k)
        k += 1
    enc5_skip = tap("enc5_skip", L.conv2d(x, _w(wd, "res3_skip_con1_3X3_weights"), _w(wd, "res3_skip_con1_3X3_biases")) + enc5)
    outs = []
    for head, h in (("Image", "1"), ("Normal", "2")):                                                         # :226-262 | :264-301
        k = "%s_e_conv6_%s_" % (head, h)
        x = L.prelu(L.conv2d(enc5_skip, _w(wd, k + "e_conv6_%s_weights" % h), _w(wd, k + "e_conv6_%s_biases" % h)), _w(wd, k + "alpha"))
        for num in (7, 8, 9):
            k = "%s_e_conv%d_%s_" % (head, num, h)
            x = L.prelu(L.conv2d_transpose(x, _w(wd, k + "e_conv%d_%s_weights" % (num, h)), _w(wd, k + "e_conv%d_%s_biases" % (num, h)), (2, 2)),
                        _w(wd, k + "alpha"))
        k = "%s_e_conv11_%s_e_conv11_%s_" % (head, h, h)
        outs.append(tap(head.lower(), torch.sigmoid(L.conv2d_transpose(x, _w(wd, k + "weights"), _w(wd, k + "biases"), (1, 1)))))
    return outs[0], outs[1]


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
def losses_and_grads(vector, param, texture, light, target, weight_dict_rendernet, weight_dict_decoder, M_inv, size, new_size,
                     light_elevation, light_col, ambient, k_diffuse, taps=None):
    """Latents as ndarrays; weight dicts as `load_weights` returns them (:337-339: the texture decoder reads the RenderNet
    folder); `M_inv` [B,3,4] = the matrices the HIP path resampled with (values used as they are; the gradient flows through
    `inverse_affine_torch(param)`).  Returns (recon_loss [B], {latent: gradient}, tensors dict)."""
    lat = {"vector": vector, "param": param, "texture": texture, "light": light}
    lat = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).requires_grad_(True) for k, v in lat.items()}
    B = lat["param"].shape[0]
    shape = decoder_3d_torch(lat["vector"], weight_dict_decoder, taps)                                        # :356
    tex = texture_decoder_pretrained_torch(lat["texture"], weight_dict_rendernet)                             # :357
    M_t = inverse_affine_torch(lat["param"], size, new_size)
    M = torch.from_numpy(np.asarray(M_inv, np.float64)) + (M_t - M_t.detach())        # value: the HIP path's matrices
    geo = to_image_layout(resample(shape, M, new_size))                                                       # :360-361
    tex_rot = to_image_layout(resample(tex, M, new_size))                                                     # :363-364
    net_in = torch.cat([geo, tex_rot], dim=4).contiguous()                                                    # :366
    img, nrm = rendernet_pretrained_torch(net_in, weight_dict_rendernet, taps)                                # :367
    light_dir = tf_generate_light_pos(lat["light"], light_elevation, B)                                       # :358
    lc = torch.from_numpy(np.tile(np.asarray(light_col, np.float32).reshape(1, 3), (B, 1)))
    shading = tf_phong_composite(nrm, light_dir, lc, ambient, k_diffuse, with_mask=True)                      # :377
    compos = img * shading                                                                                    # :378
    tgt = torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32))
    loss = torch.mean((tgt - compos) ** 2, dim=(1, 2, 3))                                                     # :383
    loss.sum().backward()                                                                                     # :404
    grads = {k: v.grad.numpy().copy() for k, v in lat.items()}
    out = {"compos": compos.detach().numpy(), "img": img.detach().numpy(), "normal": nrm.detach().numpy(),
           "shape": shape.detach().numpy(), "shading": shading.detach().numpy(), "net_in": net_in.detach().numpy()}
    return loss.detach().numpy(), grads, out
