"""torch-CPU restatement of RenderNet_Texture_Face_Normal.py: decoder_texture (:34-46), the two-head
RenderNet (:48-147) and the graph wiring (:165-179).  TEST INFRASTRUCTURE ONLY (oracle/__init__.py)."""
import numpy as np
import torch

from . import layers as L
from . import resample as R
from .rendernet import res_block_2d, res_block_3d, _g


def decoder_texture(z, w, tex_res=32, c0=4):
    """:34-46 (NumPy in/out)."""
    with torch.no_grad():
        return decoder_texture_torch(torch.from_numpy(np.asarray(z, np.float32)), w, tex_res, c0).numpy()


def decoder_texture_torch(z, w, tex_res=32, c0=4):
    """:34-46 on torch tensors (differentiable when `w` holds tensors that require grad)."""
    t = "texture_encoder/"
    if True:
        zP = L.prelu(L.fully_connected(z,
                                       _g(w, t + "e_tex_fc1/fully_connected/weights"),
                                       _g(w, t + "e_tex_fc1/fully_connected/biases")), _g(w, t + "e_tex_fc1/alpha"))
        x = zP.reshape(zP.shape[0], tex_res, tex_res, tex_res, c0)
        x = L.prelu(L.conv3d_transpose(x, _g(w, t + "e_tex_conv0/conv3d_transpose/weights"),
                                       _g(w, t + "e_tex_conv0/conv3d_transpose/biases"), (1, 1, 1)), _g(w, t + "e_tex_conv0/alpha"))
        x = L.prelu(L.conv3d_transpose(x, _g(w, t + "e_tex_conv1/conv3d_transpose/weights"),
                                       _g(w, t + "e_tex_conv1/conv3d_transpose/biases"), (2, 2, 2)), _g(w, t + "e_tex_conv1/alpha"))
        x = L.prelu(L.conv3d(x, _g(w, t + "e_tex_conv2/conv3d/weights"), _g(w, t + "e_tex_conv2/conv3d/biases"), (1, 1, 1)),
                    _g(w, t + "e_tex_conv2/alpha"))
        return x


HEAD_SCOPES = {
    "Image": [("e_conv6_1", "e_conv6_1"), ("e_conv7_1", "e_conv7_2"), ("e_conv8_1", "conv2d_transpose"),
              ("e_conv9_1", "conv2d_transpose"), ("e_conv10_1", "conv2d_transpose")],
    "Normal": [("e_conv6_2", "e_conv6_2"), ("e_conv7_2", "e_conv7_2"), ("e_conv8_2", "e_conv8_2"),
               ("e_conv9_2", "e_conv9_2"), ("e_conv10_2", "e_conv10_2")],
}


def rendernet_texture_forward(models_in, w, n_res1=10, n_res2=10, n_res3=5, taps=None):
    """:48-147.  models_in [B,H,W,D,5] -> (image, normal), each [B,4H,4W,3] (NumPy in/out)."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(models_in, dtype=np.float32))
        a, b = rendernet_texture_forward_torch(x, w, n_res1, n_res2, n_res3, taps)
        return a.numpy(), b.numpy()


def rendernet_texture_forward_torch(x, w, n_res1=10, n_res2=10, n_res3=5, taps=None):
    """The two-head graph on torch tensors (differentiable when `w` holds tensors that require grad)."""
    def tap(n, t):
        if taps is not None:
            taps[n] = t.detach().numpy().copy()
        return t

    e = "encoder/"
    if True:
        for name, s in (("e_conv1", (2, 2, 2)), ("e_conv2", (1, 1, 2)), ("e_conv3", (1, 1, 1))):
            p = e + "%s/%s/" % (name, name)
            x = L.prelu(L.conv3d(x, _g(w, p + "weights"), _g(w, p + "biases"), s), _g(w, e + name + "/alpha"))
            tap("enc" + name[-1], x)
        enc3 = x
        for k in range(1, n_res1 + 1):
            x = res_block_3d(x, w, "res1_%d" % k)
        enc3_skip = tap("enc3_skip", L.conv3d(x, _g(w, e + "res1_skip/con1_3X3/weights"), _g(w, e + "res1_skip/con1_3X3/biases")) + enc3)
        enc4 = tap("enc4", L.projection_unit(enc3_skip, _g(w, e + "projection_unit/Conv/weights"),
                                             _g(w, e + "projection_unit/Conv/biases"), _g(w, e + "projection_unit/alpha")))
        x = enc4
        for k in range(1, n_res2 + 1):
            x = res_block_2d(x, w, "res2_%d" % k)
        enc4_skip = tap("enc4_skip", L.conv2d(x, _g(w, e + "res2_skip/con1_3X3/weights"), _g(w, e + "res2_skip/con1_3X3/biases")) + enc4)
        enc5 = tap("enc5", L.prelu(L.conv2d(enc4_skip, _g(w, e + "e_conv5/e_conv5/weights"), _g(w, e + "e_conv5/e_conv5/biases")),
                                   _g(w, e + "e_conv5/alpha")))
        x = enc5
        for k in range(1, n_res3 + 1):
            x = res_block_2d(x, w, "res3_%d" % k)
        enc5_skip = tap("enc5_skip", L.conv2d(x, _g(w, e + "res3_skip/con1_3X3/weights"), _g(w, e + "res3_skip/con1_3X3/biases")) + enc5)
        outs = []
        for head in ("Image", "Normal"):
            sc = HEAD_SCOPES[head]
            p = e + head + "/"
            x = L.prelu(L.conv2d(enc5_skip, _g(w, p + "%s/%s/weights" % sc[0]), _g(w, p + "%s/%s/biases" % sc[0])),
                        _g(w, p + sc[0][0] + "/alpha"))
            for vs, cs in sc[1:4]:
                x = L.prelu(L.conv2d_transpose(x, _g(w, p + "%s/%s/weights" % (vs, cs)), _g(w, p + "%s/%s/biases" % (vs, cs)), (2, 2)),
                            _g(w, p + vs + "/alpha"))
            vs, cs = sc[4]
            logits = L.conv2d_transpose(x, _g(w, p + "%s/%s/weights" % (vs, cs)), _g(w, p + "%s/%s/biases" % (vs, cs)), (1, 1))
            tap(head.lower() + "_logits", logits)
            outs.append(tap(head.lower(), L.sigmoid(logits)))
        return outs[0], outs[1]


def render_texture(voxels, textures, poses, w, size=64, new_size=128, tex_res=32, n_res=(10, 10, 5), taps=None):
    """Graph :165-179 (eval: crop start forced to 0 at patch == new_size, tools/model_util.py:149)."""
    geo = R.net_input(voxels, poses, size, new_size)
    tex = decoder_texture(textures, w, tex_res)
    if taps is not None:
        taps["texture_decoded"] = tex
    tex_rot = R.net_input(tex, poses, size, new_size)
    x = np.concatenate([geo, tex_rot], axis=4)
    if taps is not None:
        taps["net_in"] = x
    return rendernet_texture_forward(x, w, n_res[0], n_res[1], n_res[2], taps)
