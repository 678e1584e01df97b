#!/usr/bin/env python
"""rn_winograd_output_transform on the res2 shape (F(6x6,3x3), C = 1024, B = 24) with and without a residual.  Development tool."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    B, H, W, C = 24, 64, 64, 1024
    T = B * 11 * 11
    g = torch.Generator(device="cuda").manual_seed(0)
    M = torch.randn((64, T, C), device="cuda", generator=g)
    bias = torch.randn(C, device="cuda", generator=g)
    alpha = torch.rand(C, device="cuda", generator=g)
    res = torch.randn((B, H, W, C), device="cuda", generator=g)
    y = torch.empty((B, H, W, C), device="cuda")
    st = L.stream_ptr()
    outs = {}
    for name, r, act in (("bias + PReLU", None, 1), ("bias + residual", res, 0)):
        f = lambda: lib.rn_winograd_output_transform(L.RN_WINO_F63, L.ptr(M), L.ptr(bias), L.ptr(alpha) if act else None, L.ptr(r) if r is not None else None,
                                                     L.ptr(y), None, B, H, W, C, act, st)
        for _ in range(3):
            L.check(f(), name)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(f(), name)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        nbytes = M.numel() * 4 + y.numel() * 4 * (2 if r is not None else 1)
        outs[name] = float(y.double().sum())
        print("%-16s %.4f ms  %.2f TB/s   checksum %.6e" % (name, best, nbytes / best / 1e9, outs[name]), flush=True)


if __name__ == "__main__":
    main()
