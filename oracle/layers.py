"""torch-CPU restatement of tools/layer_util.py (conv builders) with TensorFlow semantics
(SAME padding, conv_transpose = input-gradient of the SAME forward conv).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Tensors are channels-last NumPy/torch float32 arrays exactly as TF holds them:
3-D features [B,H,W,D,C], 2-D features [B,H,W,C]; filters in TF layout.
"""
import math
import torch
import torch.nn.functional as F


def same_pads(in_size, k, s):
    """TF SAME: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


def conv3d(x, w, b=None, stride=(1, 1, 1)):
    """tools/layer_util.py:228-265: tf.nn.conv3d(x, w[k1,k2,k3,Cin,Cout], [1,s1,s2,s3,1], SAME) + b."""
    x, w = _t(x), _t(w)
    xn = x.permute(0, 4, 1, 2, 3)
    pads = []
    for d in (2, 1, 0):                       # F.pad wants last dim first
        lo, hi = same_pads(x.shape[1 + d], w.shape[d], stride[d])
        pads += [lo, hi]
    y = F.conv3d(F.pad(xn, pads), w.permute(4, 3, 0, 1, 2), None, stride)
    y = y.permute(0, 2, 3, 4, 1)
    if b is not None:
        y = y + _t(b)
    return y.contiguous()


def conv2d(x, w, b=None, stride=(1, 1)):
    """tools/layer_util.py:147-183 / slim.conv2d: tf.nn.conv2d(x, w[kh,kw,Cin,Cout], SAME) + b."""
    x, w = _t(x), _t(w)
    xn = x.permute(0, 3, 1, 2)
    pads = []
    for d in (1, 0):
        lo, hi = same_pads(x.shape[1 + d], w.shape[d], stride[d])
        pads += [lo, hi]
    y = F.conv2d(F.pad(xn, pads), w.permute(3, 2, 0, 1), None, stride)
    y = y.permute(0, 2, 3, 1)
    if b is not None:
        y = y + _t(b)
    return y.contiguous()


def conv2d_transpose(x, w, b=None, stride=(1, 1)):
    """tools/layer_util.py:186-226 / slim.conv2d_transpose: tf.nn.conv2d_transpose(x,
    w[kh,kw,Cout,Cin], output = in*s, SAME).  It is the input-gradient of the SAME forward conv:
    the full transposed output of size (in-1)*s+k is cropped by the forward conv's pad_before at
    the start and kept for in*s samples."""
    x, w = _t(x), _t(w)
    xn = x.permute(0, 3, 1, 2)
    full = F.conv_transpose2d(xn, w.permute(3, 2, 0, 1), None, stride)
    sl = []
    for d in (0, 1):
        out = x.shape[1 + d] * stride[d]
        lo, _ = same_pads(out, w.shape[d], stride[d])
        sl.append(slice(lo, lo + out))
    y = full[:, :, sl[0], sl[1]].permute(0, 2, 3, 1)
    if b is not None:
        y = y + _t(b)
    return y.contiguous()


def conv3d_transpose(x, w, b=None, stride=(1, 1, 1)):
    """tools/layer_util.py:269-309: tf.nn.conv3d_transpose(x, w[k1,k2,k3,Cout,Cin], SAME)."""
    x, w = _t(x), _t(w)
    xn = x.permute(0, 4, 1, 2, 3)
    full = F.conv_transpose3d(xn, w.permute(4, 3, 0, 1, 2), None, stride)
    sl = []
    for d in (0, 1, 2):
        out = x.shape[1 + d] * stride[d]
        lo, _ = same_pads(out, w.shape[d], stride[d])
        sl.append(slice(lo, lo + out))
    y = full[:, :, sl[0], sl[1], sl[2]].permute(0, 2, 3, 4, 1)
    if b is not None:
        y = y + _t(b)
    return y.contiguous()


def fully_connected(x, w, b=None):
    """tools/layer_util.py:311-343: x @ w[in,out] + b."""
    y = _t(x) @ _t(w)
    return y + _t(b) if b is not None else y


def prelu(x, alpha):
    """tools/layer_util.py:27-45: max(0,x) + alpha*min(0,x), alpha per last-dim channel."""
    x = _t(x)
    return torch.clamp(x, min=0) + _t(alpha) * torch.clamp(x, max=0)


def projection_unit(x, w, b, alpha):
    """tools/layer_util.py:8-22: reshape [B,H,W,D,C] -> [B,H,W,D*C] (feature f = d*C + c) then
    prelu(1x1 conv).  w is the slim filter [1,1,F,F]."""
    x = _t(x)
    B, H, W, D, C = x.shape
    return prelu(conv2d(x.reshape(B, H, W, D * C), w, b), alpha)


def sigmoid(x):
    return torch.sigmoid(_t(x))


def bce_loss(pred, target):
    """RenderNet_Shader.py:160-161."""
    p, t = _t(pred), _t(target)
    return torch.mean(-torch.sum(t * torch.log(1e-6 + p) + (1 - t) * torch.log(1e-6 + 1 - p), dim=(1, 2, 3)))
