#!/usr/bin/env python
"""Filter gradient of the 3-D encoder's 3x3x3 32 -> 32 convs at the training bench's shape (crop 64: B = 24, 32 x 32 x 16): the exact-fp32
kernel (rn_conv3d_wgrad) against the bf16x3 one (rn_conv3d_wgrad_split).  Development tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    for (B, H, W, D) in ((24, 32, 32, 16), (24, 64, 64, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((B, H, W, D, 32), device="cuda", generator=g)
        dz = torch.randn((B, H, W, D, 32), device="cuda", generator=g)
        dw = torch.zeros((3, 3, 3, 32, 32), device="cuda")
        st = L.stream_ptr()
        calls = {"exact fp32": lambda: lib.rn_conv3d_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, 32, 32, L.ivec((3, 3, 3)), L.ivec((1, 1, 1)), st),
                 "bf16x3": lambda: lib.rn_conv3d_wgrad_split(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, 32, 32, st)}
        fl = 2.0 * 27 * 32 * 32 * B * H * W * D
        for name, f in calls.items():
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
            print("B=%d %dx%dx%d  %-10s %.3f ms  (%.1f TFLOP/s fp32-equivalent)" % (B, H, W, D, name, best, fl / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
