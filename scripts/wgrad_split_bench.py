#!/usr/bin/env python
"""The F(4x4,3x3) / F(4x4,4x4) filter gradient, exact-fp32 route vs the split (bf16x3) one, on the training shapes (crop 64, B = 24).
Development tool.   python scripts/wgrad_split_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402


def bench(k, B, hw, cin, cout, iters=10):
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, hw, hw, cin), device="cuda", generator=g)
    dz = torch.randn((B, hw, hw, cout), device="cuda", generator=g)
    sch = L.RN_WINO_F43 if k == 3 else L.RN_WINO_F44
    st = L.stream_ptr()
    dw0 = torch.zeros((k, k, cin, cout), device="cuda")
    dw1 = torch.zeros_like(dw0)
    wsf = torch.empty((lib.rn_conv2d_wino43_wgrad_workspace_floats if k == 3 else lib.rn_conv2d_wino44_wgrad_workspace_floats)(B, hw, hw, cin, cout), device="cuda")
    wss = torch.empty(lib.rn_winograd_split_wgrad_workspace_bytes(sch, B, hw, hw, cin, cout), dtype=torch.uint8, device="cuda")
    f32 = (lib.rn_conv2d_wino43_wgrad if k == 3 else lib.rn_conv2d_wino44_wgrad)
    runs = {"f32": lambda dw: L.check(f32(L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(wsf), B, hw, hw, cin, cout, st), "f32"),
            "split": lambda dw: L.check(lib.rn_conv2d_winograd_split_wgrad(sch, L.ptr(x), L.ptr(dz), L.ptr(dw), ctypes.c_void_p(wss.data_ptr()),
                                                                          B, hw, hw, cin, cout, st), "split")}
    out = {}
    for name, dw in (("f32", dw0), ("split", dw1)):
        runs[name](dw)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            runs[name](dw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[name] = best
    T = B * (-(-hw // 4)) ** 2
    fl = 2.0 * (36 if k == 3 else 49) * T * cin * cout
    d = float((dw0 / (iters + 1) - dw1 / (iters + 1)).abs().max()) / float((dw0 / (iters + 1)).abs().max())
    print("k%d B=%d %dx%d %d->%d (T=%d): exact fp32 %.3f ms (%.1f TFLOP/s executed)   split %.3f ms (%.1f fp32-equivalent)   |diff| %.1e of max"
          % (k, B, hw, hw, cin, cout, T, out["f32"], fl / out["f32"] / 1e9, out["split"], fl / out["split"] / 1e9, d), flush=True)


if __name__ == "__main__":
    bench(3, 24, 32, 1024, 1024)     # res2 at crop 64
    bench(3, 24, 16, 512, 512)       # res3
    bench(4, 24, 32, 1024, 512)      # e_conv5
    bench(4, 24, 16, 512, 256)       # e_conv6
    bench(3, 24, 64, 1024, 1024)     # res2 at the full 128^3 grid
