"""torch-CPU restatement of the RenderNet Phong-shader graph, RenderNet_Shader.py:32-131, built
from oracle/layers.py.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Weights are a dict keyed by the TF variable names of the reference graph,
arrays in TF layout.  Inference only: tf.nn.dropout with keep_prob 1 is the identity
(tools/layer_util.py:124-131, config_RenderNet.json:12).
"""
import numpy as np
import torch
from . import layers as L


def _g(w, name):
    v = w[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def res_block_3d(x, w, scope):
    """tools/layer_util.py:60-73."""
    p = "encoder/%s/" % scope
    net = L.prelu(L.conv3d(x, _g(w, p + "con1_3X3/weights"), _g(w, p + "con1_3X3/biases")), _g(w, p + "alpha"))
    net = L.conv3d(net, _g(w, p + "conv2_3x3/weights"), _g(w, p + "conv2_3x3/biases"))
    return net + x


def res_block_2d(x, w, scope):
    """tools/layer_util.py:91-105 (slim branch)."""
    p = "encoder/%s/" % scope
    net = L.prelu(L.conv2d(x, _g(w, p + "con1_3X3/weights"), _g(w, p + "con1_3X3/biases")), _g(w, p + "alpha"))
    net = L.conv2d(net, _g(w, p + "conv2_3x3/weights"), _g(w, p + "conv2_3x3/biases"))
    return net + x


def rendernet_forward(models_in, w, taps=None, n_res1=10, n_res2=10, n_res3=5):
    """RenderNet_Shader.py:32-131.  models_in [B,H,W,D,1] (the resampled, image-aligned voxel
    grid).  Returns the sigmoid output [B,4H,4W,ch]; if `taps` is a dict it is filled with the
    named intermediate tensors (numpy)."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(models_in, dtype=np.float32))
        return rendernet_forward_torch(x, w, taps, n_res1, n_res2, n_res3).numpy()


def rendernet_forward_torch(x, w, taps=None, n_res1=10, n_res2=10, n_res3=5):
    """The graph itself on torch tensors (differentiable when `w` holds tensors that require grad:
    oracle/train.py derives the reference's gradients from it with torch's CPU autograd)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().numpy().copy()
        return t

    if True:
        e = "encoder/"
        enc1 = L.prelu(L.conv3d(x, _g(w, e + "e_conv1/e_conv1/weights"), _g(w, e + "e_conv1/e_conv1/biases"),
                                (2, 2, 2)), _g(w, e + "e_conv1/alpha"))                      # :36-39
        tap("enc1", enc1)
        enc2 = L.prelu(L.conv3d(enc1, _g(w, e + "e_conv2/e_conv2/weights"), _g(w, e + "e_conv2/e_conv2/biases"),
                                (1, 1, 2)), _g(w, e + "e_conv2/alpha"))                      # :40-43
        tap("enc2", enc2)
        enc3 = L.prelu(L.conv3d(enc2, _g(w, e + "e_conv3/e_conv3/weights"), _g(w, e + "e_conv3/e_conv3/biases")),
                       _g(w, e + "e_conv3/alpha"))                                           # :44-47
        tap("enc3", enc3)
        net = enc3
        for k in range(1, n_res1 + 1):                                                       # :51-60
            net = res_block_3d(net, w, "res1_%d" % k)
        tap("res1", net)
        skip = L.conv3d(net, _g(w, e + "res1_skip/con1_3X3/weights"), _g(w, e + "res1_skip/con1_3X3/biases"))
        enc3_skip = tap("enc3_skip", skip + enc3)                                            # :62-64
        enc4 = tap("enc4", L.projection_unit(enc3_skip, _g(w, e + "projection_unit/Conv/weights"),
                                             _g(w, e + "projection_unit/Conv/biases"),
                                             _g(w, e + "projection_unit/alpha")))            # :67
        net = enc4
        for k in range(1, n_res2 + 1):                                                       # :71-80
            net = res_block_2d(net, w, "res2_%d" % k)
        skip = L.conv2d(net, _g(w, e + "res2_skip/con1_3X3/weights"), _g(w, e + "res2_skip/con1_3X3/biases"))
        enc4_skip = tap("enc4_skip", skip + enc4)                                            # :82-84
        enc5 = tap("enc5", L.prelu(L.conv2d(enc4_skip, _g(w, e + "e_conv5/e_conv5/weights"),
                                            _g(w, e + "e_conv5/e_conv5/biases")), _g(w, e + "e_conv5/alpha")))
        net = enc5
        for k in range(1, n_res3 + 1):                                                       # :91-95
            net = res_block_2d(net, w, "res3_%d" % k)
        skip = L.conv2d(net, _g(w, e + "res3_skip/con1_3X3/weights"), _g(w, e + "res3_skip/con1_3X3/biases"))
        enc5_skip = tap("enc5_skip", skip + enc5)                                            # :97-99
        enc6 = tap("enc6", L.prelu(L.conv2d(enc5_skip, _g(w, e + "e_conv6/e_conv6/weights"),
                                            _g(w, e + "e_conv6/e_conv6/biases")), _g(w, e + "e_conv6/alpha")))
        net = enc6
        for name, s in (("e_conv7", 2), ("e_conv7_1", 1), ("e_conv8", 2), ("e_conv9", 2), ("e_conv10", 1)):
            p = e + "%s/%s/" % (name, name)                                                  # :105-123
            net = L.prelu(L.conv2d_transpose(net, _g(w, p + "weights"), _g(w, p + "biases"), (s, s)),
                          _g(w, e + name + "/alpha"))
            tap("enc" + name[6:], net)
        logits = tap("logits", L.conv2d_transpose(net, _g(w, e + "e_conv11/weights"),
                                                  _g(w, e + "e_conv11/biases"), (1, 1)))     # :125-129
        out = tap("output", L.sigmoid(logits))                                               # :127/:130
        return out
