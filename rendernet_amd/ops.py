"""Tensor-level operators over the C ABI (include/rendernet_hip.h).

Every function takes/returns contiguous float32 torch tensors on the HIP device, channels-last
as TF holds them, launches on torch's current stream and allocates its output from torch's
caching allocator.  These are `torch.autograd.Function`s (forward implemented in HIP; the
backward of the training step -- SURVEY.md K13 -- is not built yet and raises).
"""
import ctypes

import torch

from . import _lib as L


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.RenderNetHipError("rendernet_amd ops need HIP tensors (got %s); there is no CPU path" % t.device)


class PackedWeight:
    """A conv filter converted once from TF layout to the kernels' [phase][K/4][Npad][4] layout."""

    def __init__(self, w_tf, kind, ndim):
        _chk_dev(w_tf)
        w_tf = w_tf.contiguous().float()
        self.kind, self.ndim = kind, ndim
        self.kdims = [int(k) for k in w_tf.shape[:ndim]]
        if kind == L.RN_PACK_CONV:
            self.cin, self.cout = int(w_tf.shape[ndim]), int(w_tf.shape[ndim + 1])
        else:
            self.cout, self.cin = int(w_tf.shape[ndim]), int(w_tf.shape[ndim + 1])
        lib = L.lib()
        n = lib.rn_packed_weight_floats(kind, ndim, L.ivec(self.kdims), self.cin, self.cout)
        if n == 0:
            raise L.RenderNetHipError("rn_packed_weight_floats: %s" % lib.rn_last_error().decode())
        self.data = torch.empty(n, dtype=torch.float32, device=w_tf.device)
        L.check(lib.rn_pack_weights(kind, ndim, L.ivec(self.kdims), self.cin, self.cout,
                                    L.ptr(w_tf), L.ptr(self.data), L.stream_ptr()), "rn_pack_weights")


def pack_conv(w_tf):
    """TF conv filter [k..., Cin, Cout] (2-D or 3-D)."""
    return PackedWeight(w_tf, L.RN_PACK_CONV, w_tf.dim() - 2)


def pack_conv_transpose(w_tf, stride):
    """TF conv_transpose filter [k..., Cout, Cin]."""
    return PackedWeight(w_tf, L.RN_PACK_CONVT_S1 if stride == 1 else L.RN_PACK_CONVT_S2, w_tf.dim() - 2)


def _act_code(alpha, sigmoid):
    return (L.RN_ACT_PRELU if alpha is not None else 0) | (L.RN_ACT_SIGMOID if sigmoid else 0)


class _ForwardOnly(torch.autograd.Function):
    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("rendernet_amd: backward (training step, SURVEY K13) is not implemented yet")


class _Resample(_ForwardOnly):
    @staticmethod
    def forward(ctx, vox, pose, N, window, image_layout, affine):
        _chk_dev(vox, pose)
        B, S, C = vox.shape[0], vox.shape[1], vox.shape[4]
        h0, w0, ph, pw = window
        out = torch.empty((B, ph, pw, N, C), dtype=torch.float32, device=vox.device)
        lib = L.lib()
        nws = int(lib.rn_resample_workspace_bytes(B, S, C))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=vox.device) if nws else None
        fn = lib.rn_resample_affine_fwd if affine else lib.rn_resample_fwd
        L.check(fn(L.ptr(vox), L.ptr(pose), L.ptr(out), B, S, N, C, h0, w0, ph, pw, 1 if image_layout else 0,
                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, nws, L.stream_ptr()),
                "rn_resample_fwd")
        return out


def resample(vox, pose, new_size=128, window=None, image_layout=True, affine=False):
    """vox [B,S,S,S,C], pose [B,3] (or M_inv [B,3,4] with affine=True) -> [B,ph,pw,N,C]."""
    vox = vox.contiguous().float()
    pose = pose.contiguous().float()
    if window is None:
        window = (0, 0, new_size, new_size)
    return _Resample.apply(vox, pose, int(new_size), tuple(int(v) for v in window), image_layout, affine)


def pose_to_affine(pose, size=64, new_size=128):
    pose = pose.contiguous().float()
    _chk_dev(pose)
    m = torch.empty((pose.shape[0], 3, 4), dtype=torch.float32, device=pose.device)
    L.check(L.lib().rn_pose_to_affine(L.ptr(pose), L.ptr(m), pose.shape[0], size, new_size, L.stream_ptr()),
            "rn_pose_to_affine")
    return m


# Optional measurement hook (bench.py): called as LAUNCH_HOOK(mode, x_shape, packed_weight) and must
# return None or a (start_event, end_event) pair that is recorded around the launch on the
# current stream.
LAUNCH_HOOK = None


class _Conv(_ForwardOnly):
    @staticmethod
    def forward(ctx, x, pw, bias, alpha, residual, ksize, stride, sigmoid, mode):
        _chk_dev(x, pw.data, bias, alpha, residual)
        ev = LAUNCH_HOOK(mode, tuple(x.shape), pw) if LAUNCH_HOOK is not None else None
        if ev is not None:
            ev[0].record()
        lib = L.lib()
        act = _act_code(alpha, sigmoid)
        st = L.stream_ptr()
        if mode == "conv3d":
            B, H, W, D, Cin = x.shape
            o = [-(-H // stride[0]), -(-W // stride[1]), -(-D // stride[2])]
            y = torch.empty((B, o[0], o[1], o[2], pw.cout), dtype=torch.float32, device=x.device)
            rc = lib.rn_conv3d_fwd(L.ptr(x), L.ptr(pw.data), L.ptr(bias), L.ptr(alpha), L.ptr(residual), L.ptr(y),
                                   B, H, W, D, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), act, st)
        elif mode == "conv2d":
            B, H, W, Cin = x.shape
            y = torch.empty((B, -(-H // stride[0]), -(-W // stride[1]), pw.cout), dtype=torch.float32, device=x.device)
            rc = lib.rn_conv2d_fwd(L.ptr(x), L.ptr(pw.data), L.ptr(bias), L.ptr(alpha), L.ptr(residual), L.ptr(y),
                                   B, H, W, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), act, st)
        elif mode == "conv2d_transpose":
            B, H, W, Cin = x.shape
            s = stride[0]
            y = torch.empty((B, H * s, W * s, pw.cout), dtype=torch.float32, device=x.device)
            rc = lib.rn_conv2d_transpose_fwd(L.ptr(x), L.ptr(pw.data), L.ptr(bias), L.ptr(alpha), L.ptr(residual),
                                             L.ptr(y), B, H, W, Cin, pw.cout, ksize[0], s, act, st)
        elif mode == "conv3d_transpose":
            B, H, W, D, Cin = x.shape
            s = stride[0]
            y = torch.empty((B, H * s, W * s, D * s, pw.cout), dtype=torch.float32, device=x.device)
            rc = lib.rn_conv3d_transpose_fwd(L.ptr(x), L.ptr(pw.data), L.ptr(bias), L.ptr(alpha), L.ptr(residual),
                                             L.ptr(y), B, H, W, D, Cin, pw.cout, ksize[0], s, act, st)
        else:
            raise ValueError(mode)
        L.check(rc, "rn_%s_fwd" % mode)
        if ev is not None:
            ev[1].record()
        if residual is not None and residual.shape != y.shape:
            raise L.RenderNetHipError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
        return y


def _prep(x, pw, mode_cin):
    x = x.contiguous().float()
    if x.shape[-1] != pw.cin:
        raise L.RenderNetHipError("input has %d channels, filter expects %d" % (x.shape[-1], pw.cin))
    return x


def conv3d(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1, 1), sigmoid=False):
    x = _prep(x, pw, 4)
    return _Conv.apply(x, pw, bias, alpha, residual, tuple(pw.kdims), tuple(stride), sigmoid, "conv3d")


def conv2d(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1), sigmoid=False):
    x = _prep(x, pw, 3)
    return _Conv.apply(x, pw, bias, alpha, residual, tuple(pw.kdims), tuple(stride), sigmoid, "conv2d")


def conv2d_transpose(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1), sigmoid=False):
    x = _prep(x, pw, 3)
    if stride[0] != stride[1] or pw.kdims[0] != pw.kdims[1]:
        raise L.RenderNetHipError("conv2d_transpose: square kernels/strides only")
    return _Conv.apply(x, pw, bias, alpha, residual, tuple(pw.kdims), tuple(stride), sigmoid, "conv2d_transpose")


def conv3d_transpose(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1, 1), sigmoid=False):
    x = _prep(x, pw, 4)
    return _Conv.apply(x, pw, bias, alpha, residual, tuple(pw.kdims), tuple(stride), sigmoid, "conv3d_transpose")


class _Projection(_ForwardOnly):
    @staticmethod
    def forward(ctx, x, pw, bias, alpha):
        _chk_dev(x, pw.data, bias, alpha)
        B, H, W, D, C = x.shape
        y = torch.empty((B, H, W, D * C), dtype=torch.float32, device=x.device)
        L.check(L.lib().rn_projection_fwd(L.ptr(x), L.ptr(pw.data), L.ptr(bias), L.ptr(alpha), L.ptr(y),
                                          B, H, W, D, C, L.stream_ptr()), "rn_projection_fwd")
        return y


def projection(x, pw, bias, alpha):
    """x [B,H,W,D,C] -> prelu(1x1 conv over the depth-flattened features) [B,H,W,D*C]."""
    x = x.contiguous().float()
    if x.shape[3] * x.shape[4] != pw.cin or pw.cin != pw.cout:
        raise L.RenderNetHipError("projection: D*C=%d but filter is %dx%d" % (x.shape[3] * x.shape[4], pw.cin, pw.cout))
    return _Projection.apply(x, pw, bias, alpha)


class _FC(_ForwardOnly):
    @staticmethod
    def forward(ctx, x, w, bias, alpha):
        _chk_dev(x, w, bias, alpha)
        B, fin = x.shape
        fout = w.shape[1]
        y = torch.empty((B, fout), dtype=torch.float32, device=x.device)
        L.check(L.lib().rn_fully_connected_fwd(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(alpha), L.ptr(y),
                                               B, fin, fout, _act_code(alpha, False), L.stream_ptr()),
                "rn_fully_connected_fwd")
        return y


def fully_connected(x, w, bias=None, alpha=None):
    return _FC.apply(x.contiguous().float(), w.contiguous().float(), bias, alpha)


def prelu(x, alpha):
    """Stand-alone PReLU over the last dim (tools/layer_util.py:27-45)."""
    x = x.contiguous().float()
    _chk_dev(x, alpha)
    y = torch.empty_like(x)
    L.check(L.lib().rn_prelu_fwd(L.ptr(x), L.ptr(alpha), L.ptr(y), x.numel(), x.shape[-1], L.stream_ptr()),
            "rn_prelu_fwd")
    return y


def phong_composite(normals, light_dir, light_col, ambient, k_diffuse):
    """normals [B,H,W,3] in [0,1]; light_dir, light_col [B,3] -> shaded [B,H,W,3]."""
    normals = normals.contiguous().float()
    _chk_dev(normals, light_dir, light_col)
    B, H, W, _ = normals.shape
    out = torch.empty_like(normals)
    L.check(L.lib().rn_phong_composite_fwd(L.ptr(normals), L.ptr(light_dir.contiguous().float()),
                                           L.ptr(light_col.contiguous().float()), float(ambient), float(k_diffuse),
                                           L.ptr(out), B, H, W, L.stream_ptr()), "rn_phong_composite_fwd")
    return out
