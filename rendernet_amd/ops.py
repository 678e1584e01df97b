"""Tensor-level operators over the C ABI (include/rendernet_hip.h).

Every function takes/returns contiguous float32 torch tensors on the HIP device, channels-last
as TF holds them, launches on torch's current stream and allocates its output from torch's
caching allocator.  The conv family are `torch.autograd.Function`s whose forward AND backward are
HIP kernels behind the C ABI (dgrad / wgrad / epilogue backward); torch's autograd engine only
orders the calls.  Parameter gradients do not travel through autograd: inside a `training(...)`
context each backward accumulates them straight into the gradient views the context maps the
parameter tensors to (one flat buffer, see rendernet_amd/train.py) and reports completion so that
gradient buckets can be all-reduced while the rest of the backward is still running.
"""
import contextlib
import threading
import os
import ctypes

import torch

from . import _lib as L


# Optional measurement hook (bench.py): called as LAUNCH_HOOK(mode, x_shape, packed_weight) and must
# return None or a (start_event, end_event) pair that is recorded around the launch on the
# current stream.
LAUNCH_HOOK = None


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
        self._n = lib.rn_packed_weight_floats(kind, ndim, L.ivec(self.kdims), self.cin, self.cout)
        if self._n == 0:
            raise L.RenderNetHipError("rn_packed_weight_floats: %s" % lib.rn_last_error().decode())
        self.w_tf = w_tf                  # the TF-layout master copy (what the optimiser updates)
        self._dgrad = None
        # Every packed form is built LAZILY, on first use after the master changed: a layer that runs the Winograd kernel
        # never materialises its direct pack (949 MB for the net), and a training step re-derives only the forms its
        # forward and input-gradient launches actually read.
        self._buf = {"data": None, "wino": None, "wino4": None, "wino43": None, "wino63": None, "wino43s": None, "wino63s": None, "wino43h": None, "wino63h": None, "wino3dh": None,
                     "wino3ds": None, "wino11s": None, "wino11h": None}
        self._dirty = {k: True for k in self._buf}
        self._packed_on = {}              # form -> (stream, event recorded behind its last pack kernel)
        # Winograd F(2x2,3x3) companion (csrc/conv_wino.hip): 3x3 / 3x3x3 filters whose channel counts the kernel takes;
        # used by every stride-1 launch of this filter (forward, and the input gradient through the dgrad pack).
        wkind = {L.RN_PACK_CONV: L.RN_PACK_CONV_WINO, L.RN_PACK_CONVT_S1: L.RN_PACK_CONVT_S1_WINO}.get(kind)
        self._wino_kind = wkind if (wkind is not None and (
            (ndim == 2 and self.kdims == [3, 3] and lib.rn_conv2d_wino_supported(self.cin, self.cout)) or
            (ndim == 3 and self.kdims == [3, 3, 3] and lib.rn_conv3d_wino_supported(self.cin, self.cout)))) else None
        # 4x4 stride-1 filters (e_conv5, e_conv6; e_conv7_1 as a transposed conv): four 2x2 sub-filters, each F(2x2,2x2)
        w4kind = {L.RN_PACK_CONV: L.RN_PACK_CONV_WINO4, L.RN_PACK_CONVT_S1: L.RN_PACK_CONVT_S1_WINO4}.get(kind)
        self._wino4_kind = w4kind if (w4kind is not None and ndim == 2 and self.kdims == [4, 4]
                                      and lib.rn_conv2d_wino4_supported(self.cin, self.cout)) else None
        # 4x4 STRIDE-2 transposed filters (e_conv7, e_conv8, e_conv9 and the texture net's heads): every output phase is a 2x2
        # conv, each F(2x2,2x2); the four phases run as one launch (rides in the "wino4" slot: a filter is one or the other)
        if kind == L.RN_PACK_CONVT_S2 and ndim == 2 and self.kdims == [4, 4] and lib.rn_conv2d_transpose_s2_wino_supported(self.cin, self.cout):
            self._wino4_kind = L.RN_PACK_CONVT_S2_WINO

        # Winograd with 4x4 output tiles in three launches (csrc/conv_wino43.hip): F(4x4,3x3) for the wide 3x3 2-D layers
        # (res2, res3 and their skips), F(4x4,4x4) for the wide 4x4 ones (e_conv5, e_conv6)
        self._wino43_kind = None
        if ndim == 2 and self.kdims == [3, 3] and lib.rn_conv2d_wino43_supported(self.cin, self.cout):
            self._wino43_kind = {L.RN_PACK_CONV: L.RN_PACK_CONV_WINO43, L.RN_PACK_CONVT_S1: L.RN_PACK_CONVT_S1_WINO43}.get(kind)
        elif ndim == 2 and self.kdims == [4, 4] and lib.rn_conv2d_wino44_supported(self.cin, self.cout):
            self._wino43_kind = {L.RN_PACK_CONV: L.RN_PACK_CONV_WINO44, L.RN_PACK_CONVT_S1: L.RN_PACK_CONVT_S1_WINO44}.get(kind)
        # 1x1 filters (the projection unit, tools/layer_util.py:8-22): on the split multiply stage a plain GEMM, scheme RN_WINO_F11 (one plane,
        # identity transforms; csrc/conv_wino_bf3.hip).  The exact-fp32 mode keeps the implicit-GEMM kernel for them.
        self._split11 = (ndim == 2 and self.kdims == [1, 1] and kind in (L.RN_PACK_CONV, L.RN_PACK_CONVT_S1)
                         and bool(lib.rn_winograd_split_supported(L.RN_WINO_F11, self.cin, self.cout)))
        # ... and F(6x6,3x3) for the same 3x3 layers on maps where the 6-pixel tile grid pays (_wino_scheme picks per launch)
        self._wino63_kind = None
        if self._wino43_kind is not None and self.kdims == [3, 3] and lib.rn_conv2d_wino63_supported(self.cin, self.cout):
            self._wino63_kind = {L.RN_PACK_CONV: L.RN_PACK_CONV_WINO63, L.RN_PACK_CONVT_S1: L.RN_PACK_CONVT_S1_WINO63}.get(kind)

    def _packed(self, which, kind):
        if self._dirty[which]:
            lib = L.lib()
            if which in ("wino3ds", "wino3dh"):
                # a split form of the fused 3x3x3 32 -> 32 kernel (csrc/conv3d_wino_bf3.hip), in MFMA fragment order: bf16x3 | fp16x2
                fmt = 1 if which == "wino3dh" else 0
                if self._buf[which] is None:
                    self._buf[which] = torch.empty(lib.rn_conv3d_winograd_split_packed_bytes_ex(fmt, self.cin, self.cout), dtype=torch.uint8,
                                                   device=self.w_tf.device)
                L.check(lib.rn_conv3d_winograd_split_pack_ex(fmt, L.ptr(self.w_tf), ctypes.c_void_p(self._buf[which].data_ptr()), self.cin, self.cout,
                                                             1 if self.kind == L.RN_PACK_CONVT_S1 else 0, L.stream_ptr()),
                        "rn_conv3d_winograd_split_pack_ex")
            elif which.endswith("s") or which.endswith("h"):
                # a split form of the three-launch path (csrc/conv_wino_bf3.hip): `kind` is the scheme (| the operand format flag) here
                if self._buf[which] is None:
                    n = lib.rn_winograd_split_packed_bytes(kind, self.cin, self.cout)
                    if n == 0:
                        raise L.RenderNetHipError("rn_winograd_split_packed_bytes (%s): unsupported filter" % which)
                    self._buf[which] = torch.empty(n, dtype=torch.uint8, device=self.w_tf.device)
                L.check(lib.rn_winograd_split_pack(kind, L.ptr(self.w_tf), ctypes.c_void_p(self._buf[which].data_ptr()), self.cin, self.cout,
                                                   1 if self.kind == L.RN_PACK_CONVT_S1 else 0, L.stream_ptr()), "rn_winograd_split_pack (%s)" % which)
            else:
                if self._buf[which] is None:
                    n = self._n if which == "data" else lib.rn_packed_weight_floats(kind, self.ndim, L.ivec(self.kdims), self.cin, self.cout)
                    if n == 0:
                        raise L.RenderNetHipError("rn_packed_weight_floats (%s): %s" % (which, lib.rn_last_error().decode()))
                    self._buf[which] = torch.empty(n, dtype=torch.float32, device=self.w_tf.device)
                L.check(lib.rn_pack_weights(kind, self.ndim, L.ivec(self.kdims), self.cin, self.cout, L.ptr(self.w_tf),
                                            L.ptr(self._buf[which]), L.stream_ptr()), "rn_pack_weights (%s)" % which)
            self._dirty[which] = False
            # the pack runs on the stream that first needed it; a launch on ANOTHER stream (two-stream serving sharing one
            # Renderer) must not read the buffer before that kernel has finished
            # (inside a hipGraph capture no event is recorded: it would become part of the capture and a later query() from
            # another stream would fail; Renderer.capture joins the warm-up stream before capturing, and a REpack -- an
            # optimiser step or a demotion -- is single-stream by contract: the trainers run pack and readers on one stream)
            if torch.cuda.is_current_stream_capturing():
                self._packed_on[which] = None
            else:
                ev = torch.cuda.Event()
                ev.record()
                self._packed_on[which] = (torch.cuda.current_stream(), ev)
        else:
            on = self._packed_on.get(which)
            if on is not None:
                cur = torch.cuda.current_stream()
                # (not while a hipGraph is being captured: event queries are illegal there, and Renderer.capture has joined
                # the warm-up stream before it starts capturing)
                if cur != on[0] and not torch.cuda.is_current_stream_capturing():
                    if on[1].query():
                        self._packed_on[which] = None      # finished: no stream needs to wait any more
                    else:
                        cur.wait_event(on[1])
        return self._buf[which]

    @property
    def data(self):
        """The direct-kernel pack [phase][K/4][Npad][4]."""
        return self._packed("data", self.kind)

    @property
    def wino(self):
        """The Winograd F(2x2,3x3) pack, or None when this filter does not take that path."""
        return None if self._wino_kind is None else self._packed("wino", self._wino_kind)

    @wino.setter
    def wino(self, value):
        if value is not None:
            raise ValueError("the Winograd pack can only be switched off (set to None)")
        self._wino_kind = None

    @property
    def wino43(self):
        """The Winograd F(4x4,3x3) (3x3 filters) / F(4x4,4x4) (4x4 filters) pack, or None."""
        return None if self._wino43_kind is None else self._packed("wino43", self._wino43_kind)

    @wino43.setter
    def wino43(self, value):
        if value is not None:
            raise ValueError("the F(4x4,3x3) Winograd pack can only be switched off (set to None)")
        self._wino43_kind = None          # the three-launch path as a whole
        self._wino63_kind = None

    @property
    def wino63(self):
        """The Winograd F(6x6,3x3) pack (3x3 filters), or None."""
        return None if self._wino63_kind is None else self._packed("wino63", self._wino63_kind)

    @wino63.setter
    def wino63(self, value):
        if value is not None:
            raise ValueError("the F(6x6,3x3) Winograd pack can only be switched off (set to None)")
        self._wino63_kind = None

    def split(self, which, fmt=0):
        """The split form (uint8 buffer) of scheme `which` ("f43" | "f44" | "f63") for the split GEMM stage, or None.  fmt: 0 = three
        bf16 pieces, L.RN_SPLIT_FMT_H2 = two fp16 pieces of the scaled value."""
        sfx = "h" if fmt else "s"
        if which == "f11":
            return self._packed("wino11" + sfx, L.RN_WINO_F11 | fmt) if self._split11 else None
        if which == "f63":
            return None if self._wino63_kind is None else self._packed("wino63" + sfx, L.RN_WINO_F63 | fmt)
        if self._wino43_kind is None:
            return None
        return self._packed("wino43" + sfx, (L.RN_WINO_F44 if self.kdims == [4, 4] else L.RN_WINO_F43) | fmt)

    def has_split3d(self):
        """Whether this filter has a split form for the fused 3-D kernel -- a pure predicate: nothing is packed or allocated."""
        return (self.ndim == 3 and self.kdims == [3, 3, 3] and self._wino_kind is not None
                and bool(L.lib().rn_conv3d_winograd_split_supported(self.cin, self.cout)))

    def split3d(self, fmt=0):
        """A split form of a 3x3x3 32 -> 32 filter for rn_conv3d_winograd_split_fwd_ex (uint8 buffer), or None.  fmt 0: three bf16
        pieces; 1: two fp16 pieces of the scaled value."""
        if not self.has_split3d():
            return None
        return self._packed("wino3dh" if fmt else "wino3ds", 0)

    @property
    def wino4(self):
        """The 4x4 (four F(2x2,2x2) sub-filters) pack, or None."""
        return None if self._wino4_kind is None else self._packed("wino4", self._wino4_kind)

    @wino4.setter
    def wino4(self, value):
        if value is not None:
            raise ValueError("the 4x4 Winograd pack can only be switched off (set to None)")
        self._wino4_kind = None

    def repack(self):
        """The TF-layout master changed (an optimiser step): every packed form is stale and is rebuilt when next read."""
        for k in self._dirty:
            self._dirty[k] = True
        if self._dgrad is not None:
            self._dgrad.repack()

    def dgrad_pack(self, unit_stride):
        """The packing the layer's input-gradient kernel reads (include/rendernet_hip.h, dgrad section):
        conv, stride 1 -> the same TF tensor packed as a stride-1 transposed-conv filter; strided conv ->
        the forward pack itself; transposed conv -> the same TF tensor packed as a conv filter."""
        if self.kind == L.RN_PACK_CONV and not unit_stride:
            return self
        if self._dgrad is None:
            kind = L.RN_PACK_CONVT_S1 if self.kind == L.RN_PACK_CONV else L.RN_PACK_CONV
            self._dgrad = PackedWeight(self.w_tf, kind, self.ndim)
            if getattr(self, "_wino63_demoted", False):
                self._dgrad.wino63 = None                     # the forward pack was demoted by the F(6x6,3x3) self-check
                self._dgrad._gemm_f32 = True
        return self._dgrad


def pack_conv(w_tf):
    """TF conv filter [k..., Cin, Cout] (2-D or 3-D)."""
    return PackedWeight(w_tf, L.RN_PACK_CONV, w_tf.dim() - 2)


def pack_conv_transpose(w_tf, stride):
    """TF conv_transpose filter [k..., Cout, Cin]."""
    return PackedWeight(w_tf, L.RN_PACK_CONVT_S1 if stride == 1 else L.RN_PACK_CONVT_S2, w_tf.dim() - 2)


def _act_code(alpha, sigmoid, elu=False):
    if elu and (alpha is not None or sigmoid):
        raise L.RenderNetHipError("ELU does not combine with PReLU / sigmoid in one epilogue")
    return (L.RN_ACT_PRELU if alpha is not None else 0) | (L.RN_ACT_SIGMOID if sigmoid else 0) | (L.RN_ACT_ELU if elu else 0)


class _ForwardOnly(torch.autograd.Function):
    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("rendernet_amd: this inference-only entry has no backward; inside ops.training(...) "
                                  "the projection unit runs through the differentiable conv2d path instead")


class _Resample(torch.autograd.Function):
    """Forward: rn_resample_fwd / rn_resample_affine_fwd.  Backward: rn_resample_affine_bwd (+ rn_pose_to_affine_bwd
    for the pose path) -- gradients w.r.t. the voxel grid and the pose / matrix, whichever require grad."""

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
        ev = LAUNCH_HOOK("resample", tuple(vox.shape), None) if LAUNCH_HOOK is not None else None
        if ev is not None:
            ev[0].record()
        L.check(fn(L.ptr(vox), L.ptr(pose), L.ptr(out), B, S, N, C, h0, w0, ph, pw, 1 if image_layout else 0,
                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, nws, L.stream_ptr()),
                "rn_resample_fwd")
        if ev is not None:
            ev[1].record()
        ctx.save_for_backward(vox, pose)
        ctx.cfg = (N, window, image_layout, affine)
        return out

    @staticmethod
    def backward(ctx, dout):
        vox, pose = ctx.saved_tensors
        N, (h0, w0, ph, pw), image_layout, affine = ctx.cfg
        B, S, C = vox.shape[0], vox.shape[1], vox.shape[4]
        lib, st = L.lib(), L.stream_ptr()
        dout = dout.contiguous()
        m = pose if affine else pose_to_affine(pose, S, N).reshape(B, 12)
        want_vox, want_pose = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dvox = torch.zeros_like(vox) if want_vox else None
        dm = torch.zeros((B, 12), dtype=torch.float32, device=vox.device) if want_pose else None
        if want_vox or want_pose:
            L.check(lib.rn_resample_affine_bwd(L.ptr(vox), L.ptr(m.contiguous()), L.ptr(dout), L.ptr(dvox), L.ptr(dm),
                                               B, S, N, C, h0, w0, ph, pw, 1 if image_layout else 0, st),
                    "rn_resample_affine_bwd")
        dpose = None
        if want_pose:
            if affine:
                dpose = dm.reshape(pose.shape)
            else:
                dpose = torch.zeros_like(pose)
                L.check(lib.rn_pose_to_affine_bwd(L.ptr(pose), L.ptr(dm), L.ptr(dpose), B, S, N, st), "rn_pose_to_affine_bwd")
        return dvox, dpose, None, None, None, None


def resample(vox, pose, new_size=128, window=None, image_layout=True, affine=False):
    """vox [B,S,S,S,C], pose [B,3] (or M_inv [B,3,4] with affine=True) -> [B,ph,pw,N,C]."""
    vox = vox.contiguous().float()
    pose = pose.contiguous().float()
    if window is None:
        window = (0, 0, new_size, new_size)
    return _Resample.apply(vox, pose, int(new_size), tuple(int(v) for v in window), image_layout, affine)


class _ResampleConcat(torch.autograd.Function):
    """Two volumes, one pose, one channel-concatenated output (rn_resample_concat_fwd); backward: one
    rn_resample_affine_bwd_strided per source that needs a gradient, the matrix gradient accumulated over both."""

    @staticmethod
    def forward(ctx, vox_a, vox_b, pose, N, window, image_layout, affine):
        _chk_dev(vox_a, vox_b, pose)
        B, S, Ca, Cb = vox_a.shape[0], vox_a.shape[1], vox_a.shape[4], vox_b.shape[4]
        h0, w0, ph, pw = window
        out = torch.empty((B, ph, pw, N, Ca + Cb), dtype=torch.float32, device=vox_a.device)
        ev = LAUNCH_HOOK("resample", (B, S, S, S, Ca + Cb), None) if LAUNCH_HOOK is not None else None
        if ev is not None:
            ev[0].record()
        # poses become matrices in one small launch (the same closed form the kernel would evaluate per workgroup)
        m = pose if affine else pose_to_affine(pose, S, N)
        L.check(L.lib().rn_resample_concat_fwd(L.ptr(vox_a), Ca, L.ptr(vox_b), Cb, L.ptr(m), 1, L.ptr(out),
                                               B, S, N, h0, w0, ph, pw, 1 if image_layout else 0, L.stream_ptr()),
                "rn_resample_concat_fwd")
        if ev is not None:
            ev[1].record()
        ctx.save_for_backward(vox_a, vox_b, pose)
        ctx.cfg = (N, window, image_layout, affine)
        return out

    @staticmethod
    def backward(ctx, dout):
        vox_a, vox_b, pose = ctx.saved_tensors
        N, (h0, w0, ph, pw), image_layout, affine = ctx.cfg
        B, S, Ca, Cb = vox_a.shape[0], vox_a.shape[1], vox_a.shape[4], vox_b.shape[4]
        lib, st = L.lib(), L.stream_ptr()
        dout = dout.contiguous()
        m = (pose if affine else pose_to_affine(pose, S, N).reshape(B, 12)).contiguous()
        want_pose = ctx.needs_input_grad[2]
        dm = torch.zeros((B, 12), dtype=torch.float32, device=vox_a.device) if want_pose else None
        grads = []
        for vox, C, off, want in ((vox_a, Ca, 0, ctx.needs_input_grad[0]), (vox_b, Cb, Ca, ctx.needs_input_grad[1])):
            dv = torch.zeros_like(vox) if want else None
            if want or want_pose:
                L.check(lib.rn_resample_affine_bwd_strided(L.ptr(vox), L.ptr(m), L.ptr(dout), Ca + Cb, off, L.ptr(dv), L.ptr(dm),
                                                           B, S, N, C, h0, w0, ph, pw, 1 if image_layout else 0, st),
                        "rn_resample_affine_bwd_strided")
            grads.append(dv)
        dpose = None
        if want_pose:
            if affine:
                dpose = dm.reshape(pose.shape)
            else:
                dpose = torch.zeros_like(pose)
                L.check(lib.rn_pose_to_affine_bwd(L.ptr(pose), L.ptr(dm), L.ptr(dpose), B, S, N, st), "rn_pose_to_affine_bwd")
        return grads[0], grads[1], dpose, None, None, None, None


def resample_concat(vox_a, vox_b, pose, new_size=128, window=None, image_layout=True, affine=False):
    """vox_a [B,S,S,S,Ca], vox_b [B,S,S,S,Cb], one pose [B,3] (or M_inv, affine=True) -> [B,ph,pw,N,Ca+Cb]: what
    `concat([resample(vox_a), resample(vox_b)], -1)` computes, in one pass and without the intermediates."""
    vox_a, vox_b, pose = vox_a.contiguous().float(), vox_b.contiguous().float(), pose.contiguous().float()
    if vox_a.shape[:4] != vox_b.shape[:4]:
        raise L.RenderNetHipError("resample_concat: grids %s and %s differ" % (tuple(vox_a.shape), tuple(vox_b.shape)))
    if window is None:
        window = (0, 0, new_size, new_size)
    return _ResampleConcat.apply(vox_a, vox_b, pose, int(new_size), tuple(int(v) for v in window), image_layout, affine)


def pose_to_affine(pose, size=64, new_size=128):
    pose = pose.contiguous().float()
    _chk_dev(pose)
    m = torch.empty((pose.shape[0], 3, 4), dtype=torch.float32, device=pose.device)
    L.check(L.lib().rn_pose_to_affine(L.ptr(pose), L.ptr(m), pose.shape[0], size, new_size, L.stream_ptr()),
            "rn_pose_to_affine")
    return m



# ---------------------------------------------------------------------------------------------
# training context
# ---------------------------------------------------------------------------------------------
class TrainContext:
    """Maps parameter tensors (by data_ptr) to the gradient views they accumulate into and is told
    when a parameter's gradient is complete.  `anchor` is a dummy leaf that requires grad: it rides
    along every conv call so that autograd schedules the backward of layers whose data input does
    not require grad (the first conv reads the resampled voxels)."""

    def __init__(self, grad_of=None, on_ready=None, device="cuda", frozen=False):
        self.grad_of = grad_of or {}      # {data_ptr: grad tensor}
        self.on_ready = on_ready          # callable(data_ptr) or None
        self.frozen = frozen              # pretrained weights (inverse rendering): input gradients only, no wgrad
        self.anchor = torch.zeros(1, device=device, requires_grad=True)

    def grad(self, t):
        if self.frozen:
            return None
        g = self.grad_of.get(t.data_ptr())
        if g is None:
            raise L.RenderNetHipError("no gradient buffer registered for a parameter of shape %s" % (tuple(t.shape),))
        return g

    def grad_if_param(self, t):
        """The gradient view of t, or None for a tensor MARKED as a constant that rides in a parameter slot (ops.mark_constant: the
        all-zero PReLU slope that makes the pretrained res blocks' ReLU, tools/layer_util.py:_relu_slope) -- no gradient is produced
        for it.  Any other tensor must be registered: a PReLU slope that was never registered is a bug that would otherwise show up
        as a parameter that silently never trains."""
        if self.frozen or t is None:
            return None
        g = self.grad_of.get(t.data_ptr())
        if g is None and not is_constant(t):
            raise L.RenderNetHipError("no gradient buffer registered for a parameter of shape %s (a constant in a parameter slot "
                                      "must be marked with ops.mark_constant)" % (tuple(t.shape),))
        return g

    def ready(self, *ts):
        if self.on_ready is not None:
            for t in ts:
                if t is None or is_constant(t):
                    continue
                if t.data_ptr() not in self.grad_of:
                    raise L.RenderNetHipError("ready(): a tensor of shape %s is neither a registered parameter nor a marked constant"
                                              % (tuple(t.shape),))
                self.on_ready(t.data_ptr())


def mark_constant(t):
    """Marks a tensor that sits in a parameter slot of an operator (bias / alpha) as a constant: no gradient, no on_ready call."""
    t._rn_constant = True
    return t


def is_constant(t):
    return bool(getattr(t, "_rn_constant", False))


TRAIN = None


@contextlib.contextmanager
def training(ctx):
    """Within this context the conv operators save what their backward needs and are differentiable."""
    global TRAIN
    old, TRAIN = TRAIN, ctx
    try:
        yield ctx
    finally:
        TRAIN = old


AMAX_MISSES = None     # diagnostics: set to a list to record the split-format-H2 launches that had to make their own pass over x
STAGE_HOOK = None      # bench.py: callable(stage, (T, Cin, Cout)) -> (start_event, end_event) | None, brackets the GEMM stage


def _wino_scheme(pw, H, W):
    """Which three-launch scheme a stride-1 launch of this filter on an H x W map takes: "f44" (4x4 filters), "f63" where
    ceil(H/6)*ceil(W/6) tiles of 64 products undercut ceil(H/4)*ceil(W/4) tiles of 36 by at least WINO63_MIN_GAIN (64x64:
    7744 against 9216; 32x32 and 16x16: equal, so the more accurate F(4x4,3x3) keeps them), else "f43"."""
    if pw.kdims == [4, 4]:
        return "f44"
    forced = getattr(pw, "force_scheme", None)        # measurement / tests: pin the scheme of this filter
    if forced is not None:
        return forced
    if pw._wino63_kind is not None:
        c63 = -(-H // 6) * -(-W // 6) * 64
        c43 = -(-H // 4) * -(-W // 4) * 36
        if c63 <= (1.0 - WINO63_MIN_GAIN) * c43:
            return "f63"
    return "f43"


WINO63_MIN_GAIN = float(os.environ.get("RN_WINO63_MIN_GAIN", "0.08"))

# Rounding guard of the F(6x6,3x3) route.  Measured on hostile statistics (profiles/r03a_wino_robustness.md: N(0,1), large
# positive means, log-normal channel gains, sparse spikes, 21 stacked convs) the scheme stays within 4.0e-5 * max|y| of the
# float64 conv at Cin = 1024 (F(4x4,3x3): 9.7e-6, direct: 6e-6), i.e. 5x inside the 2e-4 * max|y| bar of the per-tap tests.
# For weights / activations one does not trust, WINO63_CHECK_TOL (env RN_WINO63_CHECK_TOL, or Renderer.validate_winograd)
# turns on a self-check: before the FIRST F(6x6,3x3) launch of every filter both schemes run on that input WITHOUT epilogue (raw
# conv outputs: a residual would inflate the reference), and if max|y63 - y43| > tol * max|y43| the filter -- and its
# input-gradient pack -- is demoted to F(4x4,3x3) for good (two extra convs and one host sync per layer, once; off by default;
# never inside a hipGraph capture).  WINO63_DEMOTED lists the demotions.  RenderNet_demo.py --weights runs it at load.
WINO63_CHECK_TOL = float(os.environ["RN_WINO63_CHECK_TOL"]) if os.environ.get("RN_WINO63_CHECK_TOL") else None
WINO63_DEMOTED = []


def _wino43_fwd(x, pw, e, B, H, W, Cin, Cout, act, y_t=None, amax_out=None):
    """rn_conv2d_wino43_fwd / _wino63_fwd / _wino44_fwd with the workspace (V and M planes) from torch's caching allocator.
    pw: the PackedWeight whose .wino43 / .wino63 form the launch reads (a stride-1 transposed 4x4 pack pads two pixels before).
    y_t: the output tensor behind e[3] (needed by the F(6x6,3x3) self-check only).  amax_out: a list that receives the device word
    with max|y| when the launch produced one (split format H2)."""
    which = _wino_scheme(pw, H, W)
    if (which == "f63" and WINO63_CHECK_TOL is not None and y_t is not None and getattr(pw, "_wino63_verdict", None) is None
            and getattr(pw, "force_scheme", None) is None and not torch.cuda.is_current_stream_capturing()):
        # Both schemes on the same input with NO epilogue (no bias, activation or residual): the residual of a res block or a
        # *_skip conv would inflate max|y| and hide a conv error well above tol * max|conv|.  Then the real launch.
        # The candidate is the route this launch would take (F(6x6,3x3) in the ACTIVE multiply-stage mode: bf16x3 by default); the
        # yardstick is F(4x4,3x3) on the exact-fp32 MFMA stage, the most accurate three-launch route there is -- so the check covers
        # the transform's rounding and the operand split in one comparison.
        raw = (None, None, None)
        y63, y43 = torch.empty_like(y_t), torch.empty_like(y_t)
        rc = _wino43_run(x, pw, raw + (L.ptr(y63), None), B, H, W, Cin, Cout, 0, "f63")
        if rc != 0:
            return rc
        rc = _wino43_run(x, pw, raw + (L.ptr(y43), None), B, H, W, Cin, Cout, 0, "f43", gemm="f32")
        if rc != 0:
            return rc
        diff, ref = float((y63 - y43).abs().max()), float(y43.abs().max())
        pw._wino63_verdict = diff <= WINO63_CHECK_TOL * ref
        if not pw._wino63_verdict:
            WINO63_DEMOTED.append({"cin": Cin, "cout": Cout, "map": (H, W), "rel_diff": diff / max(ref, 1e-30), "mode": gemm_mode_now()})
            pw.wino63 = None                                  # this filter takes F(4x4,3x3) on the exact-fp32 stage from now on ...
            pw._gemm_f32 = True
            if pw._dgrad is not None:
                pw._dgrad.wino63 = None                       # ... and so does its input-gradient pack
                pw._dgrad._gemm_f32 = True
            pw._wino63_demoted = True
            which = "f43"
    return _wino43_run(x, pw, e, B, H, W, Cin, Cout, act, which, amax_out=amax_out)


# Multiply stage of the three-launch path: "f32" = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), "split" = the same products on
# the 16x faster bf16 pipe with every fp32 operand as three bf16 pieces and six piece products, fp32 accumulation
# (csrc/conv_wino_bf3.hip; fp32-class error, not bit-identical to "f32").  env RN_WINO_GEMM = the process default; per call: ops.gemm_mode(...).
# "split16": the same stage with every operand as TWO fp16 pieces of value / (power-of-two scale of its tensor) and three products --
# half the matrix work of "split"; 22-bit operands, fp32 accumulation (the accumulation error, which all three modes share, dominates).
# DEFAULT since round 5: "split" -- the bf16x3 form carries the full 24-bit significand of every fp32 operand (the three pieces sum
# EXACTLY to the fp32 value) and accumulates in fp32, its error against float64 equals the exact route's on every case of the hostile-
# statistics suite, and it is 1.3x faster end to end.  RN_WINO_GEMM=f32 restores the exact-fp32 MFMA stage everywhere (it is also what a
# filter is demoted to when the F(6x6,3x3) self-check below rejects it); "split16" is the opt-in fast mode.
GEMM_MODES = ("f32", "split", "split16")
WINO_GEMM = os.environ.get("RN_WINO_GEMM", "split")         # the PROCESS DEFAULT only: what a caller that names no mode gets
if WINO_GEMM not in GEMM_MODES:
    raise ValueError("RN_WINO_GEMM=%r: expected one of %s" % (WINO_GEMM, ", ".join(GEMM_MODES)))
# The mode of a call is the innermost `with ops.gemm_mode(m)` of the calling thread, else the process default above.  Renderer /
# TextureRenderer / Trainer / Reconstructor carry it as an attribute (`gemm=`) and enter the context around everything they launch, so
# two renderers of one process can run different modes; a conv's backward runs under the mode its forward captured (autograd calls it
# from its own thread).  Nothing in this module WRITES WINO_GEMM.
_MODE = threading.local()


def gemm_mode_now():
    """The multiply-stage mode of a launch made now by this thread."""
    return getattr(_MODE, "mode", None) or WINO_GEMM


@contextlib.contextmanager
def gemm_mode(mode):
    """`with ops.gemm_mode("f32" | "split" | "split16")`: every launch of this thread inside the block uses that multiply stage;
    None = leave whatever is in force."""
    if mode is None:
        yield
        return
    if mode not in GEMM_MODES:
        raise ValueError("gemm mode %r: expected one of %s" % (mode, ", ".join(GEMM_MODES)))
    prev = getattr(_MODE, "mode", None)
    _MODE.mode = mode
    try:
        yield
    finally:
        _MODE.mode = prev
# The fused 3x3x3 32 -> 32 kernel of the 3-D encoder has a bf16x3 variant too (csrc/conv3d_wino_bf3.hip: 0.50 ms against 0.82 ms on
# the B = 24 64x64x32 layer, error 2.4e-7 .. 3.8e-7 of max|y| against the fp32 kernel's 3.2e-7 .. 4.9e-7).  None: it follows
# WINO_GEMM ("split" turns both on); True / False (env RN_CONV3D_SPLIT=1 / 0) force it independently.
CONV3D_SPLIT = {"": None, "0": False}.get(os.environ.get("RN_CONV3D_SPLIT", ""), True)


# The filter gradients of the wide 2-D convs in split mode too (RN_WGRAD_SPLIT=0: keep them on the exact-fp32 route)
WGRAD_SPLIT = os.environ.get("RN_WGRAD_SPLIT", "1") not in ("", "0")
CARRY_SKIP_GRADIENT = os.environ.get("RN_NO_CARRY", "") in ("", "0")      # res blocks: the skip path's gradient is added inside conv1's input-gradient launch (see _conv_apply)
# the wider of (Cin, Cout) from which the 2-D Winograd filter gradient takes the split GEMM stage.  Measured at crop 64: 1024 -> 1024 1.10 -> 0.79 ms, 1024 -> 512 (4x4)
# 0.92 -> 0.65; 512 -> 512 no gain on the 32x32x16 form of the stage, on the 16x16x32 form the training step goes 87.1 -> 85.9 ms (profiles/r06z_wgrad_min_ch.txt)
WGRAD_SPLIT_MIN_CH = int(os.environ.get("RN_WGRAD_SPLIT_MIN_CH", "512"))


def _conv3d_split(B=None, H=None, W=None):
    """An item of the split kernel is a row of 16 tiles through the depth slices (ceil(H/2) * ceil(W/32) rows per image), cut into depth
    segments when the rows do not fill the 256 workgroups (csrc/conv3d_wino_bf3.hip: c3_depth_segments; res1 layer 0.048 against 0.057 ms
    at B = 1, 0.068 against 0.093 at B = 2).  The gate is a PER-IMAGE quantity (round 6, advisor finding on round 5's B * rows >= 64):
    a frame takes the same kernels, and gets the same bits, alone and inside a batch -- the split kernel's results do not depend on the
    item count (any segment count sums the same taps in the same order).  Maps of fewer than 8 rows of tiles per image (16 x 16 at crop
    32 is the smallest the nets produce: 8 rows) stay on the fp32 kernel.  B is accepted and ignored."""
    if CONV3D_SPLIT is not None:
        return bool(CONV3D_SPLIT)
    return gemm_mode_now() in ("split", "split16") and (H is None or ((H + 1) // 2) * ((W + 31) // 32) >= CONV3D_SPLIT_MIN_ROWS)


CONV3D_SPLIT_MIN_ROWS = int(os.environ.get("RN_CONV3D_SPLIT_MIN_ROWS", "8"))


def _gemm_mode(pw, gemm=None):
    """The multiply-stage mode of a launch of this filter: the per-call override, else exact fp32 for a filter the self-check demoted,
    else the module-wide mode."""
    if gemm is not None:
        return gemm
    return "f32" if getattr(pw, "_gemm_f32", False) else gemm_mode_now()


def _wino43_run(x, pw, e, B, H, W, Cin, Cout, act, which, gemm=None, amax_out=None):
    lib = L.lib()
    gmode = _gemm_mode(pw, gemm)
    f44, f63, f11 = which == "f44", which == "f63", which == "f11"
    transposed = 1 if pw.kind == L.RN_PACK_CONVT_S1 else 0
    m = 6 if f63 else 1 if f11 else 4
    T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
    scheme, nxi = (L.RN_WINO_F44, 49) if f44 else (L.RN_WINO_F63, 64) if f63 else (L.RN_WINO_F11, 1) if f11 else (L.RN_WINO_F43, 36)
    if f11 and gmode == "f32":
        raise L.RenderNetHipError("the 1x1 GEMM route exists on the split multiply stage only")
    st = L.stream_ptr()
    if gmode in ("split", "split16") and lib.rn_winograd_split_supported(scheme, Cin, Cout):
        fmt = L.RN_SPLIT_FMT_H2 if gmode == "split16" else 0
        us = ctypes.c_void_p(pw.split(which, fmt).data_ptr())
        scheme |= fmt
        ws = torch.empty(lib.rn_winograd_split_workspace_bytes(scheme, B, H, W, Cin, Cout), dtype=torch.uint8, device=x.device)
        wsp = ctypes.c_void_p(ws.data_ptr())
        # format H2 scales every operand tensor by (a bound from) its max|x|: the launch that produced x left it in x._rn_amax (a device
        # word); without one the launcher makes a pass over x.  This launch leaves max|y| for the next layer the same way.
        ax = _amax_of(x) if fmt else None
        if fmt and ax is None and AMAX_MISSES is not None:
            AMAX_MISSES.append(("conv2d", which, Cin, Cout, H, W))
        ay = torch.empty(1, dtype=torch.int32, device=x.device) if fmt else None
        axp = ctypes.c_void_p(ax.data_ptr()) if ax is not None else None
        ayp = ctypes.c_void_p(ay.data_ptr()) if ay is not None else None
        if ay is not None and amax_out is not None:
            amax_out.append(ay)
        ev = STAGE_HOOK("gemm", (T, Cin, Cout, which)) if STAGE_HOOK is not None and T * max(Cin, Cout) * 4 < 0x7fffff00 else None
        if ev is None:
            return lib.rn_conv2d_winograd_split_fwd_ex(scheme, L.ptr(x), us, *e, wsp, B, H, W, Cin, Cout, transposed, act, axp, ayp, st)
        M = ctypes.c_void_p(ws.data_ptr() + lib.rn_winograd_split_v_bytes(scheme, T, Cin))
        rc = lib.rn_winograd_split_input_transform_ex(scheme, L.ptr(x), wsp, B, H, W, Cin, 0 if f11 else 2 if (f44 and transposed) else 1, axp, st)
        if rc != 0:
            return rc
        ev[0].record()
        rc = lib.rn_winograd_split_gemm(scheme, wsp, us, M, T, Cin, Cout, st)
        ev[1].record()
        if rc != 0:
            return rc
        return lib.rn_winograd_output_transform_ex(scheme & 0xff, M, *e, B, H, W, Cout, act, ayp, st)
    u = pw.wino63 if f63 else pw.wino43
    n = (lib.rn_conv2d_wino44_workspace_floats if f44 else lib.rn_conv2d_wino63_workspace_floats if f63
         else lib.rn_conv2d_wino43_workspace_floats)(B, H, W, Cin, Cout)
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    ev = STAGE_HOOK("gemm", (T, Cin, Cout, which)) if STAGE_HOOK is not None and T * max(Cin, Cout) * 4 < 0x7fffff00 else None
    if ev is None:
        if f44:
            return lib.rn_conv2d_wino44_fwd(L.ptr(x), L.ptr(u), *e, L.ptr(ws), B, H, W, Cin, Cout, transposed, act, L.stream_ptr())
        if f63:
            return lib.rn_conv2d_wino63_fwd(L.ptr(x), L.ptr(u), *e, L.ptr(ws), B, H, W, Cin, Cout, act, L.stream_ptr())
        return lib.rn_conv2d_wino43_fwd(L.ptr(x), L.ptr(u), *e, L.ptr(ws), B, H, W, Cin, Cout, act, L.stream_ptr())
    # the same three launches through the stage entry points, the GEMM bracketed by the caller's events
    V, M = L.ptr(ws), ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * Cin)
    rc = lib.rn_winograd_input_transform(scheme, L.ptr(x), V, B, H, W, Cin, 2 if (f44 and transposed) else 1, st)
    if rc != 0:
        return rc
    ev[0].record()
    rc = lib.rn_winograd_gemm(scheme, V, L.ptr(u), M, T, Cin, Cout, st)
    ev[1].record()
    if rc != 0:
        return rc
    return lib.rn_winograd_output_transform(scheme, M, *e, B, H, W, Cout, act, st)


def _use_wino43(pw, H, W):
    return pw._wino43_kind is not None and H * W >= WINO43_MIN_PIXELS


def _use_split11(pw, pixels):
    """A 1x1 filter takes the split GEMM stage (three launches: split, GEMM, epilogue) in the split modes when ONE IMAGE has at least
    `SPLIT11_MIN_PIXELS` pixels (pixels = H * W: a per-image gate, so that a frame is routed alike alone and in a batch); below that the
    one-launch implicit-GEMM kernel wins.  32 x 32 (the projection unit at crop 64) is the smallest map measured faster on the stage."""
    return pw._split11 and _gemm_mode(pw) in ("split", "split16") and pixels >= SPLIT11_MIN_PIXELS


SPLIT11_MIN_PIXELS = int(os.environ.get("RN_SPLIT11_MIN_PIXELS", "1024"))


WINO43_MIN_PIXELS = int(os.environ.get("RN_WINO43_MIN_PIXELS", "64"))


def _amax_of(x):
    """The device word with max|x| that the launch which produced x left on it (split format H2) -- only while x is what that launch
    wrote: an in-place change bumps the tensor's version counter and the word is ignored (the launcher then makes its own pass)."""
    tag = getattr(x, "_rn_amax", None)
    if tag is None:
        return None
    word, version = tag
    return word if _version_of(x) == version else None


def _version_of(t):
    """t._version, or None where the counter does not exist (tensors created under torch.inference_mode() raise on access): such a
    tensor gets no tag and the launcher makes its own pass over it."""
    try:
        return t._version
    except RuntimeError:
        return None


def _tag_amax(y, word):
    v = _version_of(y)
    if word is not None and v is not None:
        y._rn_amax = (word, v)


def _conv3d_split_fmt():
    return 1 if gemm_mode_now() == "split16" else 0


def _conv3d_split_launch(x, pw, e, B, H, W, D, Cin, Cout, act, st, amax_out=None):
    """The fused 3x3x3 32 -> 32 kernel in the split format of the mode: bf16x3 ("split"), fp16x2 ("split16": max|x| from the producing
    launch via x._rn_amax, else a pass; max|y| left for the consumer in amax_out)."""
    lib = L.lib()
    fmt = _conv3d_split_fmt()
    us = ctypes.c_void_p(pw.split3d(fmt).data_ptr())
    if not fmt:
        return lib.rn_conv3d_winograd_split_fwd_ex(0, L.ptr(x), us, *e, B, H, W, D, Cin, Cout, act, None, None, None, st)
    ax = _amax_of(x)
    if ax is None and AMAX_MISSES is not None:
        AMAX_MISSES.append(("conv3d", Cin, Cout, H, W, D))
    words = torch.empty(2, dtype=torch.int32, device=x.device)       # [0]: max|y| for the consumer, [1]: scratch for a pass over x
    if amax_out is not None:
        amax_out.append(words[:1])
    return lib.rn_conv3d_winograd_split_fwd_ex(1, L.ptr(x), us, *e, B, H, W, D, Cin, Cout, act,
                                               ctypes.c_void_p(ax.data_ptr()) if ax is not None else None,
                                               ctypes.c_void_p(words.data_ptr() + 4), ctypes.c_void_p(words.data_ptr()), st)


def _launch_conv(mode, x, pw, bias, alpha, residual, y, z, ksize, stride, act, amax_out=None):
    lib, st = L.lib(), L.stream_ptr()
    e = (L.ptr(bias), L.ptr(alpha), L.ptr(residual), L.ptr(y), L.ptr(z))          # the epilogue arguments of every entry
    unit = all(int(v) == 1 for v in stride)
    if mode == "conv3d":
        B, H, W, D, Cin = x.shape
        if unit and _conv3d_split(B, H, W) and pw.has_split3d():
            return _conv3d_split_launch(x, pw, e, B, H, W, D, Cin, pw.cout, act, st, amax_out)
        if unit and pw.wino is not None:
            return lib.rn_conv3d_wino_fwd(L.ptr(x), L.ptr(pw.wino), *e, B, H, W, D, Cin, pw.cout, act, st)
        return lib.rn_conv3d_fwd_train(L.ptr(x), L.ptr(pw.data), *e, B, H, W, D, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), act, st)
    if mode == "conv2d":
        B, H, W, Cin = x.shape
        if unit and _use_wino43(pw, H, W):
            return _wino43_fwd(x, pw, e, B, H, W, Cin, pw.cout, act, y, amax_out)
        if unit and _use_split11(pw, H * W):
            return _wino43_run(x, pw, e, B, H, W, Cin, pw.cout, act, "f11", amax_out=amax_out)
        if unit and pw.wino is not None:
            return lib.rn_conv2d_wino_fwd(L.ptr(x), L.ptr(pw.wino), *e, B, H, W, Cin, pw.cout, act, st)
        if unit and pw.wino4 is not None:
            return lib.rn_conv2d_wino4_fwd(L.ptr(x), L.ptr(pw.wino4), *e, B, H, W, Cin, pw.cout, 0, act, st)
        return lib.rn_conv2d_fwd_train(L.ptr(x), L.ptr(pw.data), *e, B, H, W, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), act, st)
    if mode == "conv2d_transpose":
        B, H, W, Cin = x.shape
        if unit and _use_wino43(pw, H, W):
            return _wino43_fwd(x, pw, e, B, H, W, Cin, pw.cout, act, None, amax_out)
        if unit and pw.wino4 is not None:
            return lib.rn_conv2d_wino4_fwd(L.ptr(x), L.ptr(pw.wino4), *e, B, H, W, Cin, pw.cout, 1, act, st)
        if (not unit) and int(stride[0]) == 2 and pw.kind == L.RN_PACK_CONVT_S2 and pw.wino4 is not None:
            return lib.rn_conv2d_transpose_s2_wino_fwd(L.ptr(x), L.ptr(pw.wino4), *e, B, H, W, Cin, pw.cout, act, st)
        return lib.rn_conv2d_transpose_fwd_train(L.ptr(x), L.ptr(pw.data), *e, B, H, W, Cin, pw.cout, ksize[0], stride[0], act, st)
    if mode == "conv3d_transpose":
        B, H, W, D, Cin = x.shape
        return lib.rn_conv3d_transpose_fwd_train(L.ptr(x), L.ptr(pw.data), *e, B, H, W, D, Cin, pw.cout, ksize[0], stride[0], act, st)
    raise ValueError(mode)


def _out_shape(mode, x, pw, stride):
    sp = x.shape[1:-1]
    if mode in ("conv3d", "conv2d"):
        return (x.shape[0],) + tuple(-(-int(n) // int(s)) for n, s in zip(sp, stride)) + (pw.cout,)
    return (x.shape[0],) + tuple(int(n) * int(stride[0]) for n in sp) + (pw.cout,)


_EPI_WS = {}


def _epilogue_ws(device, C):
    """The partial-sum workspace of rn_epilogue_bwd_ws: one per (device, stream) -- its contents live only inside a call, calls on one
    stream are ordered -- grown to the widest layer seen (512 row blocks x [2][C] floats: 4 MiB at C = 1024)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    n = L.lib().rn_epilogue_bwd_workspace_floats(0, int(C))
    ws = _EPI_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _EPI_WS[key] = torch.empty(n, dtype=torch.float32, device=device)
    return ws


class _Conv(torch.autograd.Function):
    """y = sigmoid?( prelu?( conv(x, w) + bias ) + residual ) for the four conv flavours; one HIP launch
    forward (plus the saved pre-activation when training), three backward (epilogue, dgrad, wgrad)."""

    @staticmethod
    def forward(ctx, x, pw, bias, alpha, residual, ksize, stride, sigmoid, mode, anchor, elu=False, amax_box=None, carry=False):
        _chk_dev(x, pw.w_tf, bias, alpha, residual)
        ev = LAUNCH_HOOK(mode, tuple(x.shape), pw) if LAUNCH_HOOK is not None else None
        if ev is not None:
            ev[0].record()
        act = _act_code(alpha, sigmoid, elu)
        train = anchor is not None
        y = torch.empty(_out_shape(mode, x, pw, stride), dtype=torch.float32, device=x.device)
        if residual is not None and residual.shape != y.shape:
            raise L.RenderNetHipError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
        z = torch.empty_like(y) if (train and alpha is not None) else None
        # split format H2: the launch leaves the device word with max|y| in `amax`; it travels to the consumer as a tag on y (the
        # caller's box carries it across _Conv.apply, which may hand back another tensor object for the same storage)
        amax = [] if amax_box is None else amax_box
        L.check(_launch_conv(mode, x, pw, bias, alpha, residual, y, z, ksize, stride, act, amax), "rn_%s_fwd" % mode)
        if amax:
            _tag_amax(y, amax[-1])
        if ev is not None:
            ev[1].record()
        if train:
            ctx.save_for_backward(x, z, y if (sigmoid or elu) else None)
            ctx.cfg = (pw, bias, alpha, residual is not None, tuple(ksize), tuple(stride), act, mode, TRAIN)
            ctx.gemm = gemm_mode_now()                # the backward runs on autograd's thread: it takes the forward's mode along
            ctx.carry = bool(carry)
        if carry:
            # training only (conv*(..., carry=True)): x rides along as a second output.  A res block hands THAT to its second conv as the
            # residual, so the skip path's gradient comes back into THIS node (as the gradient of the second output) and is added to dx
            # in the epilogue of the input-gradient launch -- instead of a separate read-modify-write add_ by the autograd engine
            # (27 per training step of the shader net, 100 MB tensors)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dcarry=None):
        with gemm_mode(ctx.gemm):
            return _Conv._backward(ctx, dy, dcarry)

    @staticmethod
    def _backward(ctx, dy, dcarry=None):
        x, z, y = ctx.saved_tensors
        pw, bias, alpha, has_res, ksize, stride, act, mode, tc = ctx.cfg
        lib, st = L.lib(), L.stream_ptr()
        dy = dy.contiguous()
        C = dy.shape[-1]
        M = dy.numel() // C
        # 1. epilogue backward: dz (new buffer only when the values change), dbias, dalpha
        dz = torch.empty_like(dy) if act else dy
        if act or (bias is not None and not tc.frozen):
            ws = _epilogue_ws(dy.device, C)
            L.check(lib.rn_epilogue_bwd_ws(L.ptr(dy), L.ptr(z), L.ptr(y), L.ptr(alpha), L.ptr(dz) if act else None,
                                           L.ptr(tc.grad(bias)) if bias is not None else None,
                                           L.ptr(tc.grad_if_param(alpha)),
                                           M, C, act, L.ptr(ws), ws.numel(), st), "rn_epilogue_bwd_ws")
        d_res = None
        if has_res and ctx.needs_input_grad[4]:
            if act & L.RN_ACT_SIGMOID:
                raise L.RenderNetHipError("backward of sigmoid + residual in one epilogue is not supported")
            d_res = dy                      # PReLU sits before the residual add: its gradient is dy itself
        # 2. wgrad, accumulated into the registered gradient view (TF layout); none with frozen weights
        dw = tc.grad(pw.w_tf)
        unit = all(int(v) == 1 for v in stride)
        if mode in ("conv3d", "conv3d_transpose"):
            B, H, W, D, Cin = x.shape
        else:
            B, H, W, Cin = x.shape
        if dw is None:
            rc = 0
        elif (mode == "conv3d" and unit and tuple(ksize) == (3, 3, 3) and _gemm_mode(pw) in ("split", "split16") and WGRAD_SPLIT
              and lib.rn_conv3d_wgrad_split_supported(Cin, pw.cout)
              and H * W * D * Cin * 4 < 0x80000000):      # one image inside the 2 GiB buffer window (else the exact kernel's launcher reports the limit)
            # the 3-D encoder's 32 -> 32 filter gradients: the reduction over the positions on the bf16 pipe (bf16x3 in both split modes)
            rc = lib.rn_conv3d_wgrad_split(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, Cin, pw.cout, st)
        elif mode == "conv3d":
            rc = lib.rn_conv3d_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), st)
        elif (mode == "conv2d" and unit and tuple(ksize) in ((3, 3), (4, 4)) and _use_wino43(pw, H, W) and _gemm_mode(pw) in ("split", "split16") and WGRAD_SPLIT
              and max(Cin, pw.cout) >= WGRAD_SPLIT_MIN_CH
              and lib.rn_winograd_split_wgrad_supported(L.RN_WINO_F43 if ksize[0] == 3 else L.RN_WINO_F44, Cin, pw.cout)):
            # the reduction over the tiles on the bf16 pipe (csrc/conv_wino_bf3_wgrad.hip)
            sch = L.RN_WINO_F43 if ksize[0] == 3 else L.RN_WINO_F44
            ws = torch.empty(lib.rn_winograd_split_wgrad_workspace_bytes(sch, B, H, W, Cin, pw.cout), dtype=torch.uint8, device=x.device)
            wev = STAGE_HOOK("wgrad", (B * (-(-H // 4)) * (-(-W // 4)), Cin, pw.cout, "f43s" if ksize[0] == 3 else "f44s")) if STAGE_HOOK is not None else None
            if wev is not None:
                wev[0].record()
            rc = lib.rn_conv2d_winograd_split_wgrad(sch, L.ptr(x), L.ptr(dz), L.ptr(dw), ctypes.c_void_p(ws.data_ptr()), B, H, W, Cin, pw.cout, st)
            if wev is not None:
                wev[1].record()
        elif (mode == "conv2d" and unit and tuple(ksize) == (3, 3) and _use_wino43(pw, H, W)
              and lib.rn_conv2d_wino43_wgrad_supported(Cin, pw.cout)):
            ws = torch.empty(lib.rn_conv2d_wino43_wgrad_workspace_floats(B, H, W, Cin, pw.cout), dtype=torch.float32, device=x.device)
            wev = STAGE_HOOK("wgrad", (B * (-(-H // 4)) * (-(-W // 4)), Cin, pw.cout, "f43")) if STAGE_HOOK is not None else None
            if wev is not None:
                wev[0].record()
            rc = lib.rn_conv2d_wino43_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(ws), B, H, W, Cin, pw.cout, st)
            if wev is not None:
                wev[1].record()
        elif (mode == "conv2d" and unit and tuple(ksize) == (4, 4) and _use_wino43(pw, H, W)
              and lib.rn_conv2d_wino44_wgrad_supported(Cin, pw.cout)):
            ws = torch.empty(lib.rn_conv2d_wino44_wgrad_workspace_floats(B, H, W, Cin, pw.cout), dtype=torch.float32, device=x.device)
            rc = lib.rn_conv2d_wino44_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(ws), B, H, W, Cin, pw.cout, st)
        elif (mode == "conv2d" and unit and tuple(ksize) == (3, 3) and pw._wino_kind is not None
              and lib.rn_conv2d_wino_wgrad_supported(Cin, pw.cout)):
            rc = lib.rn_conv2d_wino_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, Cin, pw.cout, st)
        elif mode == "conv2d":
            rc = lib.rn_conv2d_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), st)
        elif mode == "conv2d_transpose":
            rc = lib.rn_conv2d_transpose_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, Cin, pw.cout, ksize[0], stride[0], st)
        else:
            rc = lib.rn_conv3d_transpose_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, Cin, pw.cout, ksize[0], stride[0], st)
        L.check(rc, "rn_%s_wgrad" % mode)
        # 3. dgrad
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            dp = pw.dgrad_pack(unit)
            # the gradient that came in through the carried copy of x (a res block's skip path): added by the launch's residual epilogue where
            # the route has one, by one add_ otherwise
            rp, fused = None, False
            if dcarry is not None:
                dcarry = dcarry.contiguous()
                rp = L.ptr(dcarry)
            if mode == "conv3d" and unit and _conv3d_split(B, H, W) and dp.has_split3d():
                rc = _conv3d_split_launch(dz, dp, (None, None, rp, L.ptr(dx), None), B, H, W, D, pw.cout, Cin, 0, st); fused = True
            elif mode == "conv3d" and unit and dp.wino is not None:
                rc = lib.rn_conv3d_wino_fwd(L.ptr(dz), L.ptr(dp.wino), None, None, rp, L.ptr(dx), None,
                                            B, H, W, D, pw.cout, Cin, 0, st); fused = True
            elif mode == "conv3d":
                rc = lib.rn_conv3d_dgrad(L.ptr(dz), L.ptr(dp.data), L.ptr(dx), B, H, W, D, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), st)
            elif mode in ("conv2d", "conv2d_transpose") and unit and _use_wino43(dp, H, W):
                rc = _wino43_fwd(dz, dp, (None, None, rp, L.ptr(dx), None), B, H, W, pw.cout, Cin, 0); fused = True
            elif mode == "conv2d" and unit and _use_split11(dp, H * W):
                # 1x1: the input gradient is the GEMM with the transposed filter (the same TF tensor packed the other way round)
                rc = _wino43_run(dz, dp, (None, None, None, L.ptr(dx), None), B, H, W, pw.cout, Cin, 0, "f11")
            elif mode == "conv2d" and unit and dp.wino is not None:
                # stride-1 3x3: the input gradient is the same conv with the flipped, transposed filter
                rc = lib.rn_conv2d_wino_fwd(L.ptr(dz), L.ptr(dp.wino), None, None, rp, L.ptr(dx), None,
                                            B, H, W, pw.cout, Cin, 0, st); fused = True
            elif mode == "conv2d" and unit and dp.wino4 is not None:
                # stride-1 4x4: the input gradient is the stride-1 transposed conv of the same filter
                rc = lib.rn_conv2d_wino4_fwd(L.ptr(dz), L.ptr(dp.wino4), None, None, None, L.ptr(dx), None,
                                             B, H, W, pw.cout, Cin, 1, 0, st)
            elif mode == "conv2d_transpose" and unit and dp.wino4 is not None:
                # ... and a stride-1 transposed conv's is the plain conv of the same filter
                rc = lib.rn_conv2d_wino4_fwd(L.ptr(dz), L.ptr(dp.wino4), None, None, None, L.ptr(dx), None,
                                             B, H, W, pw.cout, Cin, 0, 0, st)
            elif mode == "conv2d":
                rc = lib.rn_conv2d_dgrad(L.ptr(dz), L.ptr(dp.data), L.ptr(dx), B, H, W, Cin, pw.cout, L.ivec(ksize), L.ivec(stride), st)
            elif mode == "conv2d_transpose":
                rc = lib.rn_conv2d_transpose_dgrad(L.ptr(dz), L.ptr(dp.data), L.ptr(dx), B, H, W, Cin, pw.cout, ksize[0], stride[0], st)
            else:
                rc = lib.rn_conv3d_transpose_dgrad(L.ptr(dz), L.ptr(dp.data), L.ptr(dx), B, H, W, D, Cin, pw.cout, ksize[0], stride[0], st)
            L.check(rc, "rn_%s_dgrad" % mode)
            if dcarry is not None and not fused:
                dx.add_(dcarry)
        if not tc.frozen:
            tc.ready(pw.w_tf, bias, alpha)
        return dx, None, None, None, d_res, None, None, None, None, None, None, None, None


def _prep(x, pw, mode_cin):
    x = x.contiguous().float()
    if x.shape[-1] != pw.cin:
        raise L.RenderNetHipError("input has %d channels, filter expects %d" % (x.shape[-1], pw.cin))
    return x


def _conv_apply(x, pw, bias, alpha, residual, ksize, stride, sigmoid, mode, elu=False, carry=False):
    """carry=True: returns (y, x') with x' = x; under an ops.training context x' is a second output of the conv's graph node, and a gradient
    that reaches it (the skip path of a res block that uses x' as its residual) is added to dx inside the input-gradient launch."""
    anchor = TRAIN.anchor if (TRAIN is not None and torch.is_grad_enabled()) else None
    if anchor is None and not torch.is_grad_enabled():
        # inference: no graph to record -- skip the autograd.Function machinery (at batch 1 the 80 launches of a render
        # are host-bound; this is a fifth of the per-launch cost)
        y = _Conv.forward(None, x, pw, bias, alpha, residual, tuple(ksize), tuple(stride), sigmoid, mode, None, elu)
        return (y, x) if carry else y
    box = []
    fuse = bool(carry) and anchor is not None and CARRY_SKIP_GRADIENT
    out = _Conv.apply(x, pw, bias, alpha, residual, tuple(ksize), tuple(stride), sigmoid, mode, anchor, elu, box, fuse)
    xc = x
    if fuse:
        out, xc = out
    if box and getattr(out, "_rn_amax", None) is None:
        _tag_amax(out, box[-1])                                     # autograd returned another tensor object for the same storage
    return (out, xc) if carry else out


def conv3d(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1, 1), sigmoid=False, elu=False, carry=False):
    x = _prep(x, pw, 4)
    return _conv_apply(x, pw, bias, alpha, residual, pw.kdims, stride, sigmoid, "conv3d", elu, carry)


def conv2d(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1), sigmoid=False, elu=False, carry=False):
    x = _prep(x, pw, 3)
    return _conv_apply(x, pw, bias, alpha, residual, pw.kdims, stride, sigmoid, "conv2d", elu, carry)


def conv2d_transpose(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1), sigmoid=False, elu=False):
    x = _prep(x, pw, 3)
    if stride[0] != stride[1] or pw.kdims[0] != pw.kdims[1]:
        raise L.RenderNetHipError("conv2d_transpose: square kernels/strides only")
    return _conv_apply(x, pw, bias, alpha, residual, pw.kdims, stride, sigmoid, "conv2d_transpose", elu)


def conv3d_transpose(x, pw, bias=None, alpha=None, residual=None, stride=(1, 1, 1), sigmoid=False, elu=False):
    x = _prep(x, pw, 4)
    return _conv_apply(x, pw, bias, alpha, residual, pw.kdims, stride, sigmoid, "conv3d_transpose", elu)


def _res_stack_unfused(x, blocks, skip):
    net = x
    for pw1, b1, a1, pw2, b2 in blocks:
        h, net = conv2d(net, pw1, b1, a1, carry=True)      # (training: the skip path's gradient joins dx inside this conv's input-gradient launch)
        net = conv2d(h, pw2, b2, None, net)
    if skip is not None:
        net = conv2d(net, skip[0], skip[1], None, skip[2])
    return net


# Measured on MI355X (scripts/outin_bench.py, res2 shape, B=24, profiles/r03b_outin_bench.txt): conv1 -> conv2 fused 0.457 ms against
# 0.202 + 0.210 ms for the two launches, conv2 -> conv1 (+ residual, + y) 1.106 against 0.315 + 0.210; whole step 114.5 against
# 105.4 ms.  The ring of 18 pixel rows x 16 channels (78 KB) leaves two 128-thread workgroups per CU, i.e. one wave per SIMD:
# the fused kernel is latency-bound at 2.1-3.3 TB/s where the separate launches stream at 5-5.8 TB/s with ~6 waves per SIMD.
# Off by default (RN_RES_STACK_FUSION=1 turns it on); kept because it is bit-exact and tested, and for the measurement.
RES_STACK_FUSED = bool(os.environ.get("RN_RES_STACK_FUSION"))
RES_STACK_STATS = {"fused": 0, "unfused": 0}          # how many stacks took which path (tests; diagnostics)


def res_stack_2d(x, blocks, skip=None):
    """A stack of res_block_2d (tools/layer_util.py:91-105: x + conv(prelu(conv(x)))) and, optionally, the *_skip conv behind
    it (conv(x_n) + skip residual; RenderNet_Shader.py:71-84, :91-99), 3x3 stride 1, C -> C throughout.
        blocks: [(pw1, bias1, alpha1, pw2, bias2), ...]    skip: (pw, bias, residual tensor) | None
    By default this is the loop over the per-layer launches.  With RES_STACK_FUSED (opt-in, see above: measured slower) inference
    on the three-launch Winograd path runs the 2n+1 convs as ONE chain: input transform, then per conv the GEMM stage and --
    instead of an output transform followed by the next conv's input transform -- the fused transform
    (rn_winograd_output_input_transform): the activation between two convs never goes to HBM unless it is a block's output
    (the next block's residual).  Values are bit-identical to the per-layer path."""
    x = x.contiguous().float()
    pws = [p for blk in blocks for p in (blk[0], blk[3])] + ([skip[0]] if skip is not None else [])
    B, H, W, C = x.shape
    lib = L.lib()
    fused = (RES_STACK_FUSED and len(blocks) > 0 and not (TRAIN is not None and torch.is_grad_enabled()) and not torch.is_grad_enabled()
             and WINO63_CHECK_TOL is None and gemm_mode_now() == "f32"            # the chain runs the exact-fp32 GEMM stage
             and all(p.kind == L.RN_PACK_CONV and p.kdims == [3, 3] and p.cin == C and p.cout == C and _use_wino43(p, H, W) for p in pws))
    which = None
    if fused:
        schemes = {_wino_scheme(p, H, W) for p in pws}
        which = schemes.pop() if len(schemes) == 1 else None
        scheme, nxi, m = {"f43": (L.RN_WINO_F43, 36, 4), "f63": (L.RN_WINO_F63, 64, 6)}.get(which, (None, 0, 1))
        T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
        fused = (scheme is not None and T * C * 4 < 0x7fffff00 and
                 bool(lib.rn_winograd_output_input_supported(scheme, H, W, C, L.RN_ACT_PRELU)))
    RES_STACK_STATS["fused" if fused else "unfused"] += 1
    if not fused:
        return _res_stack_unfused(x, blocks, skip)
    _chk_dev(x)
    st = L.stream_ptr()
    ws = torch.empty(2 * nxi * T * C, dtype=torch.float32, device=x.device)
    V, M = L.ptr(ws), ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * C)
    u = (lambda p: p.wino63) if which == "f63" else (lambda p: p.wino43)

    def layer(pw, first, fn):
        """GEMM stage of `pw` + the transform behind it (fn), bracketed for bench.py like a per-layer launch."""
        ev = LAUNCH_HOOK("conv2d", (B, H, W, C), pw) if LAUNCH_HOOK is not None else None
        if ev is not None:
            ev[0].record()
        if first:
            L.check(lib.rn_winograd_input_transform(scheme, L.ptr(x), V, B, H, W, C, 1, st), "rn_winograd_input_transform")
        sev = STAGE_HOOK("gemm", (T, C, C, which)) if STAGE_HOOK is not None else None
        if sev is not None:
            sev[0].record()
        L.check(lib.rn_winograd_gemm(scheme, V, L.ptr(u(pw)), M, T, C, C, st), "rn_winograd_gemm")
        if sev is not None:
            sev[1].record()
        out = fn()
        if ev is not None:
            ev[1].record()
        return out

    def outin(bias, alpha, res, y, act):
        L.check(lib.rn_winograd_output_input_transform(scheme, M, L.ptr(bias), L.ptr(alpha), L.ptr(res), L.ptr(y), V,
                                                       B, H, W, C, act, st), "rn_winograd_output_input_transform")
        return y

    def out(bias, res):
        y = torch.empty_like(x)
        L.check(lib.rn_winograd_output_transform(scheme, M, L.ptr(bias), None, L.ptr(res), L.ptr(y), None, B, H, W, C, 0, st),
                "rn_winograd_output_transform")
        return y

    cur = x                                   # the current block's input = its residual
    for k, (pw1, b1, a1, pw2, b2) in enumerate(blocks):
        _chk_dev(pw1.w_tf, b1, a1, pw2.w_tf, b2)
        layer(pw1, k == 0, lambda: outin(b1, a1, None, None, L.RN_ACT_PRELU if a1 is not None else 0))
        last = k == len(blocks) - 1
        if last and skip is None:
            return layer(pw2, False, lambda: out(b2, cur))
        # a block's output is the next block's residual: it goes to HBM as well; the last block's (input of the skip conv only) does not
        ynew = None if last else torch.empty_like(x)
        layer(pw2, False, lambda: outin(b2, None, cur, ynew, 0))
        cur = ynew
    if skip[2] is not None and skip[2].shape != x.shape:
        raise L.RenderNetHipError("res_stack_2d: skip residual shape %s != %s" % (tuple(skip[2].shape), tuple(x.shape)))
    return layer(skip[0], False, lambda: out(skip[1], skip[2]))


class _Projection(_ForwardOnly):
    @staticmethod
    def forward(ctx, x, pw, bias, alpha):
        _chk_dev(x, pw.w_tf, bias, alpha)
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
    B, H, W, D, C = x.shape
    if (TRAIN is not None and torch.is_grad_enabled()) or _use_split11(pw, H * W):
        # training: the same GEMM through the differentiable 1x1 conv2d path (the reshape is a view); split modes: that path takes the 1x1
        # filter through the split GEMM stage (exact fp32: the one-launch kernel below)
        return _conv_apply(x.view(B, H, W, D * C), pw, bias, alpha, None, (1, 1), (1, 1), False, "conv2d")
    return _Projection.apply(x, pw, bias, alpha)


class _FC(torch.autograd.Function):
    """y = prelu?(x @ w + b) (tools/layer_util.py:311-343); backward through rn_epilogue_bwd + rn_fully_connected_bwd."""

    @staticmethod
    def forward(ctx, x, w, bias, alpha, anchor):
        _chk_dev(x, w, bias, alpha)
        B, fin = x.shape
        fout = w.shape[1]
        train = anchor is not None
        y = torch.empty((B, fout), dtype=torch.float32, device=x.device)
        z = torch.empty_like(y) if (train and alpha is not None) else None
        act = _act_code(alpha, False)
        L.check(L.lib().rn_fully_connected_fwd_train(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(alpha), L.ptr(y), L.ptr(z),
                                                     B, fin, fout, act, L.stream_ptr()), "rn_fully_connected_fwd")
        if train:
            ctx.save_for_backward(x, z)
            ctx.cfg = (w, bias, alpha, act, TRAIN)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z = ctx.saved_tensors
        w, bias, alpha, act, tc = ctx.cfg
        lib, st = L.lib(), L.stream_ptr()
        dy = dy.contiguous()
        B, fin = x.shape
        fout = w.shape[1]
        dz = torch.empty_like(dy) if act else dy
        if act or (bias is not None and not tc.frozen):
            L.check(lib.rn_epilogue_bwd(L.ptr(dy), L.ptr(z), None, L.ptr(alpha), L.ptr(dz) if act else None,
                                        L.ptr(tc.grad(bias)) if bias is not None else None,
                                        L.ptr(tc.grad(alpha)) if alpha is not None else None, B, fout, act, st),
                    "rn_epilogue_bwd")
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        L.check(lib.rn_fully_connected_bwd(L.ptr(x), L.ptr(w), L.ptr(dz), L.ptr(dx), L.ptr(tc.grad(w)), B, fin, fout, st),
                "rn_fully_connected_bwd")
        if not tc.frozen:
            tc.ready(w, bias, alpha)
        return dx, None, None, None, None


def fully_connected(x, w, bias=None, alpha=None):
    anchor = TRAIN.anchor if (TRAIN is not None and torch.is_grad_enabled()) else None
    return _FC.apply(x.contiguous().float(), w.contiguous().float(), bias, alpha, anchor)


class _Dropout(torch.autograd.Function):
    """tf.nn.dropout through rn_dropout: the mask is a pure function of (seed, stream id, element index), so the
    backward regenerates it by running the same call on the gradient -- nothing is saved."""

    @staticmethod
    def forward(ctx, x, keep_prob, seed, stream_id):
        _chk_dev(x)
        y = torch.empty_like(x)
        L.check(L.lib().rn_dropout(L.ptr(x), L.ptr(y), x.numel(), keep_prob, seed, stream_id, L.stream_ptr()), "rn_dropout")
        ctx.cfg = (keep_prob, seed, stream_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        keep_prob, seed, stream_id = ctx.cfg
        dy = dy.contiguous()              # may be an offset view of a larger gradient buffer: rn_dropout takes any alignment
        dx = torch.empty_like(dy)
        L.check(L.lib().rn_dropout(L.ptr(dy), L.ptr(dx), dy.numel(), keep_prob, seed, stream_id, L.stream_ptr()), "rn_dropout (bwd)")
        return dx, None, None, None


_DROPOUT_SEED = [0x5EED0FD50, 0]          # process-wide (seed, next stream id): every call draws a fresh stream


def mix_seed(*parts):
    """splitmix64 over a sequence of integers -> one 64-bit seed (e.g. (trainer seed, rank): ranks of a data-parallel job
    must not draw the same masks for their shards)."""
    z = 0x9E3779B97F4A7C15
    for p in parts:
        z = (z + (int(p) & (2 ** 64 - 1)) * 0xBF58476D1CE4E5B9 + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & (2 ** 64 - 1)
        z ^= z >> 31
    return z


def seed_dropout(seed, first_stream=0):
    """Re-seed the dropout generator and set the id of the next stream: two runs with the same (seed, first_stream) and the
    same sequence of dropout calls draw the same masks.  The trainers call this at the start of every forward with
    seed = mix_seed(trainer seed, rank) and first_stream = global_step << 16, so a mask is a pure function of
    (seed, rank, step, position of the dropout site in the graph): independent across ranks and steps like tf.nn.dropout's,
    and a resumed run continues the sequence instead of replaying it."""
    _DROPOUT_SEED[0], _DROPOUT_SEED[1] = int(seed) & (2 ** 64 - 1), int(first_stream)


def dropout_state():
    return tuple(_DROPOUT_SEED)


def dropout(x, keep_prob, seed=None, stream_id=None):
    """tf.nn.dropout(x, keep_prob) (RenderNet_Shader.py:39 ...): x / keep_prob * floor(keep_prob + U[0,1)); the identity
    at keep_prob >= 1.  Differentiable.  (seed, stream_id) pin the mask; by default every call takes the next stream of
    the process-wide generator (see seed_dropout)."""
    if keep_prob >= 1.0 or x.numel() == 0:
        return x
    if seed is None:
        seed = _DROPOUT_SEED[0]
    if stream_id is None:
        stream_id = _DROPOUT_SEED[1]
        _DROPOUT_SEED[1] += 1
    return _Dropout.apply(x.contiguous().float(), float(keep_prob), int(seed), int(stream_id))


def prelu(x, alpha):
    """Stand-alone PReLU over the last dim (tools/layer_util.py:27-45)."""
    x = x.contiguous().float()
    _chk_dev(x, alpha)
    y = torch.empty_like(x)
    L.check(L.lib().rn_prelu_fwd(L.ptr(x), L.ptr(alpha), L.ptr(y), x.numel(), x.shape[-1], L.stream_ptr()),
            "rn_prelu_fwd")
    return y


PHONG_MODES = {"np_black": L.RN_PHONG_NP_BLACK, "np_white": L.RN_PHONG_NP_WHITE, "tf_black": L.RN_PHONG_TF_BLACK,
               "tf_white": L.RN_PHONG_TF_WHITE, "none": L.RN_PHONG_NO_MASK}


class _Phong(torch.autograd.Function):
    """Phong composite (all flavours of tools/Phong_shading.py) with TF's gradients w.r.t. the normal map, the light
    direction and the albedo it is multiplied with (Reconstruct_RenderNet_Face.py:377-378)."""

    @staticmethod
    def forward(ctx, normals, light_dir, light_col, albedo, ambient, k_diffuse, mode):
        _chk_dev(normals, light_dir, light_col, albedo)
        B, H, W, _ = normals.shape
        out = torch.empty_like(normals)
        L.check(L.lib().rn_phong_composite_ex_fwd(L.ptr(normals), L.ptr(light_dir), L.ptr(light_col), L.ptr(albedo),
                                                  ambient, k_diffuse, L.ptr(out), B, H, W, mode, L.stream_ptr()),
                "rn_phong_composite_ex_fwd")
        ctx.save_for_backward(normals, light_dir, light_col, albedo)
        ctx.cfg = (ambient, k_diffuse, mode)
        return out

    @staticmethod
    def backward(ctx, dout):
        normals, light_dir, light_col, albedo = ctx.saved_tensors
        ambient, k_diffuse, mode = ctx.cfg
        B, H, W, _ = normals.shape
        dout = dout.contiguous()
        dn = torch.empty_like(normals) if ctx.needs_input_grad[0] else None
        dl = torch.zeros_like(light_dir) if ctx.needs_input_grad[1] else None
        da = torch.empty_like(albedo) if (albedo is not None and ctx.needs_input_grad[3]) else None
        L.check(L.lib().rn_phong_composite_bwd(L.ptr(normals), L.ptr(light_dir), L.ptr(light_col), L.ptr(albedo),
                                               ambient, k_diffuse, L.ptr(dout), L.ptr(dn), L.ptr(dl), L.ptr(da),
                                               B, H, W, mode, L.stream_ptr()), "rn_phong_composite_bwd")
        return dn, dl, None, da, None, None, None


def phong_composite(normals, light_dir, light_col, ambient, k_diffuse, mode="np_black", albedo=None):
    """normals [B,H,W,3] in [0,1]; light_dir, light_col [B,3] -> shading [B,H,W,3] (times `albedo` when given).
    mode: mask flavour, see include/rendernet_hip.h (np_black = the demo)."""
    normals = normals.contiguous().float()
    if normals.dim() != 4 or normals.shape[-1] != 3:
        raise L.RenderNetHipError("phong_composite: normals must be [B,H,W,3], got %s" % (tuple(normals.shape),))
    B = normals.shape[0]
    light_dir = light_dir.float().expand(B, 3).contiguous()
    light_col = light_col.float().expand(B, 3).contiguous()
    if albedo is not None:
        albedo = albedo.contiguous().float()
        if albedo.shape != normals.shape:
            raise L.RenderNetHipError("phong_composite: albedo shape %s != normals shape %s" % (tuple(albedo.shape), tuple(normals.shape)))
    return _Phong.apply(normals, light_dir, light_col, albedo, float(ambient), float(k_diffuse), PHONG_MODES[mode])
