"""ctypes binding of librendernet_hip.so (include/rendernet_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol is absent, importing
this module's `lib()` raises.  Tensors are torch CUDA(=HIP) tensors; only their device pointers
and the current stream cross the ABI.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RN_HIP_LIBRARY") or os.path.join(_HERE, "lib", "librendernet_hip.so")   # override: A/B builds

RN_ACT_NONE, RN_ACT_PRELU, RN_ACT_SIGMOID, RN_ACT_ELU = 0, 1, 2, 4
RN_PHONG_NP_BLACK, RN_PHONG_NP_WHITE, RN_PHONG_TF_BLACK, RN_PHONG_TF_WHITE, RN_PHONG_NO_MASK = 0, 1, 2, 3, 4
RN_PACK_CONV, RN_PACK_CONVT_S1, RN_PACK_CONVT_S2, RN_PACK_CONV_WINO, RN_PACK_CONVT_S1_WINO = 0, 1, 2, 3, 4
RN_PACK_CONV_WINO4, RN_PACK_CONVT_S1_WINO4 = 5, 6
RN_PACK_CONV_WINO43, RN_PACK_CONVT_S1_WINO43 = 7, 8
RN_PACK_CONV_WINO44, RN_PACK_CONVT_S1_WINO44 = 9, 10
RN_PACK_CONV_WINO63, RN_PACK_CONVT_S1_WINO63 = 11, 12
RN_PACK_CONVT_S2_WINO = 13
RN_WINO_F43, RN_WINO_F44, RN_WINO_F63, RN_WINO_F11 = 0, 1, 2, 3
RN_SPLIT_FMT_H2 = 0x100                  # operand format flag of the rn_winograd_split_* entries (OR-ed into the scheme)

_c_int, _c_vp, _c_f = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); must list every symbol include/rendernet_hip.h declares
SIGNATURES = {
    "rn_version": (_c_int, []),
    "rn_last_error": (ctypes.c_char_p, []),
    "rn_resample_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int]),
    "rn_resample_fwd": (_c_int, [_c_vp, _c_vp, _c_vp] + [_c_int] * 9 + [_c_vp, ctypes.c_size_t, _c_vp]),
    "rn_resample_affine_fwd": (_c_int, [_c_vp, _c_vp, _c_vp] + [_c_int] * 9 + [_c_vp, ctypes.c_size_t, _c_vp]),
    "rn_resample_concat_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_int, _c_vp] + [_c_int] * 8 + [_c_vp]),
    "rn_resample_affine_bwd_strided": (_c_int, [_c_vp] * 3 + [_c_int, _c_int, _c_vp, _c_vp] + [_c_int] * 9 + [_c_vp]),
    "rn_pose_to_affine": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "rn_packed_weight_floats": (ctypes.c_size_t, [_c_int, _c_int, _ip, _c_int, _c_int]),
    "rn_pack_weights": (_c_int, [_c_int, _c_int, _ip, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "rn_conv3d_fwd": (_c_int, [_c_vp] * 6 + [_c_int] * 6 + [_ip, _ip, _c_int, _c_vp]),
    "rn_conv2d_fwd": (_c_int, [_c_vp] * 6 + [_c_int] * 5 + [_ip, _ip, _c_int, _c_vp]),
    "rn_conv2d_transpose_fwd": (_c_int, [_c_vp] * 6 + [_c_int] * 8 + [_c_vp]),
    "rn_conv3d_transpose_fwd": (_c_int, [_c_vp] * 6 + [_c_int] * 9 + [_c_vp]),
    "rn_projection_fwd": (_c_int, [_c_vp] * 5 + [_c_int] * 5 + [_c_vp]),
    "rn_conv2d_wino_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino_fwd": (_c_int, [_c_vp] * 7 + [_c_int] * 6 + [_c_vp]),
    "rn_conv2d_wino4_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino4_fwd": (_c_int, [_c_vp] * 7 + [_c_int] * 7 + [_c_vp]),
    "rn_conv2d_transpose_s2_wino_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_transpose_s2_wino_fwd": (_c_int, [_c_vp] * 7 + [_c_int] * 6 + [_c_vp]),
    "rn_winograd_output_input_supported": (_c_int, [_c_int] * 5),
    "rn_winograd_output_input_transform": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 5 + [_c_vp]),
    "rn_conv3d_wino_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv3d_wino_fwd": (_c_int, [_c_vp] * 7 + [_c_int] * 7 + [_c_vp]),
    "rn_fully_connected_fwd": (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    "rn_prelu_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_size_t, _c_int, _c_vp]),
    "rn_phong_composite_fwd": (_c_int, [_c_vp] * 3 + [_c_f, _c_f, _c_vp] + [_c_int] * 3 + [_c_vp]),
    # training step
    "rn_conv3d_fwd_train": (_c_int, [_c_vp] * 7 + [_c_int] * 6 + [_ip, _ip, _c_int, _c_vp]),
    "rn_conv2d_fwd_train": (_c_int, [_c_vp] * 7 + [_c_int] * 5 + [_ip, _ip, _c_int, _c_vp]),
    "rn_conv2d_transpose_fwd_train": (_c_int, [_c_vp] * 7 + [_c_int] * 8 + [_c_vp]),
    "rn_conv3d_transpose_fwd_train": (_c_int, [_c_vp] * 7 + [_c_int] * 9 + [_c_vp]),
    "rn_fully_connected_fwd_train": (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [_c_vp]),
    "rn_fully_connected_bwd": (_c_int, [_c_vp] * 5 + [_c_int] * 3 + [_c_vp]),
    "rn_epilogue_bwd": (_c_int, [_c_vp] * 7 + [ctypes.c_size_t, _c_int, _c_int, _c_vp]),
    "rn_epilogue_bwd_workspace_floats": (ctypes.c_size_t, [ctypes.c_size_t, _c_int]),
    "rn_epilogue_bwd_ws": (_c_int, [_c_vp] * 7 + [ctypes.c_size_t, _c_int, _c_int, _c_vp, ctypes.c_size_t, _c_vp]),
    "rn_conv3d_dgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 6 + [_ip, _ip, _c_vp]),
    "rn_conv2d_dgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_ip, _ip, _c_vp]),
    "rn_conv2d_transpose_dgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [_c_vp]),
    "rn_conv3d_transpose_dgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 8 + [_c_vp]),
    "rn_conv3d_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 6 + [_ip, _ip, _c_vp]),
    "rn_conv2d_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_ip, _ip, _c_vp]),
    "rn_conv3d_wgrad_split_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv3d_wgrad_split": (_c_int, [_c_vp] * 3 + [_c_int] * 6 + [_c_vp]),
    "rn_conv2d_wino43_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino43_workspace_floats": (ctypes.c_size_t, [_c_int] * 5),
    "rn_conv2d_wino43_fwd": (_c_int, [_c_vp] * 8 + [_c_int] * 6 + [_c_vp]),
    "rn_conv2d_wino63_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino63_workspace_floats": (ctypes.c_size_t, [_c_int] * 5),
    "rn_conv2d_wino63_fwd": (_c_int, [_c_vp] * 8 + [_c_int] * 6 + [_c_vp]),
    "rn_conv2d_wino44_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino44_workspace_floats": (ctypes.c_size_t, [_c_int] * 5),
    "rn_conv2d_wino44_fwd": (_c_int, [_c_vp] * 8 + [_c_int] * 7 + [_c_vp]),
    "rn_winograd_input_transform": (_c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    "rn_winograd_gemm": (_c_int, [_c_int] + [_c_vp] * 3 + [ctypes.c_longlong, _c_int, _c_int, _c_vp]),
    "rn_winograd_output_transform": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 5 + [_c_vp]),
    "rn_winograd_split_supported": (_c_int, [_c_int] * 3),
    "rn_winograd_split_packed_bytes": (ctypes.c_size_t, [_c_int] * 3),
    "rn_winograd_split_v_bytes": (ctypes.c_size_t, [_c_int, ctypes.c_longlong, _c_int]),
    "rn_winograd_split_workspace_bytes": (ctypes.c_size_t, [_c_int] * 6),
    "rn_winograd_split_pack": (_c_int, [_c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "rn_winograd_split_input_transform": (_c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    "rn_winograd_split_gemm": (_c_int, [_c_int] + [_c_vp] * 3 + [ctypes.c_longlong, _c_int, _c_int, _c_vp]),
    "rn_conv2d_winograd_split_fwd": (_c_int, [_c_int] + [_c_vp] * 8 + [_c_int] * 7 + [_c_vp]),
    "rn_conv2d_winograd_split_fwd_ex": (_c_int, [_c_int] + [_c_vp] * 8 + [_c_int] * 7 + [_c_vp] * 3),
    "rn_winograd_split_input_transform_ex": (_c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "rn_winograd_output_transform_ex": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 5 + [_c_vp, _c_vp]),
    "rn_absmax": (_c_int, [_c_vp, ctypes.c_longlong, _c_vp, _c_vp]),
    "rn_winograd_split_wgrad_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "rn_winograd_split_wgrad_workspace_bytes": (ctypes.c_size_t, [_c_int] * 6),
    "rn_conv2d_winograd_split_wgrad": (_c_int, [_c_int] + [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    "rn_conv3d_winograd_split_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv3d_winograd_split_packed_bytes_ex": (ctypes.c_size_t, [_c_int, _c_int, _c_int]),
    "rn_conv3d_winograd_split_pack_ex": (_c_int, [_c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "rn_conv3d_winograd_split_fwd_ex": (_c_int, [_c_int] + [_c_vp] * 7 + [_c_int] * 7 + [_c_vp] * 4),
    "rn_conv3d_winograd_split_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "rn_conv3d_winograd_split_pack": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "rn_conv3d_winograd_split_fwd": (_c_int, [_c_vp] * 7 + [_c_int] * 7 + [_c_vp]),
    "rn_conv2d_wino43_wgrad_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino43_wgrad_workspace_floats": (ctypes.c_size_t, [_c_int] * 5),
    "rn_conv2d_wino43_wgrad": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    "rn_conv2d_wino44_wgrad_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino44_wgrad_workspace_floats": (ctypes.c_size_t, [_c_int] * 5),
    "rn_conv2d_wino44_wgrad": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    "rn_conv2d_wino_wgrad_supported": (_c_int, [_c_int, _c_int]),
    "rn_conv2d_wino_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    "rn_conv2d_transpose_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [_c_vp]),
    "rn_conv3d_transpose_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 8 + [_c_vp]),
    "rn_resample_affine_bwd": (_c_int, [_c_vp] * 5 + [_c_int] * 9 + [_c_vp]),
    "rn_pose_to_affine_bwd": (_c_int, [_c_vp] * 3 + [_c_int] * 3 + [_c_vp]),
    "rn_dropout": (_c_int, [_c_vp, _c_vp, ctypes.c_size_t, _c_f, ctypes.c_uint64, ctypes.c_uint64, _c_vp]),
    "rn_loss_fwd_bwd": (_c_int, [_c_vp] * 4 + [ctypes.c_size_t, ctypes.c_double, _c_int, _c_vp]),
    "rn_adam_step": (_c_int, [_c_vp] * 4 + [ctypes.c_size_t] + [_c_f] * 5 + [_c_vp]),
    "rn_sgd_step": (_c_int, [_c_vp, _c_vp, ctypes.c_size_t, _c_f, _c_vp]),
    # inverse rendering
    "rn_phong_composite_ex_fwd": (_c_int, [_c_vp] * 4 + [_c_f, _c_f, _c_vp] + [_c_int] * 4 + [_c_vp]),
    "rn_phong_composite_bwd": (_c_int, [_c_vp] * 4 + [_c_f, _c_f] + [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
}

_lib = None


class RenderNetHipError(RuntimeError):
    pass


_F32 = [None]          # torch.float32, filled in by lib() (torch is imported lazily)


def lib():
    """Load (once) and return the ctypes library.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RenderNetHipError(
                "librendernet_hip.so not found at %s -- build it with `python -m rendernet_amd.build` "
                "(there is no CPU fallback for the render path)" % LIB_PATH)
        import torch  # first: the library must bind to the HIP runtime torch ships, not a second copy
        _F32[0] = torch.float32
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().rn_last_error()
        raise RenderNetHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def ivec(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def ptr(t):
    """Device pointer of a contiguous float32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RenderNetHipError("expected a CUDA/HIP tensor, got device %s" % t.device)
    if _F32[0] is None:
        lib()
    if t.dtype is not _F32[0] or not t.is_contiguous():
        raise RenderNetHipError("expected a contiguous float32 tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
