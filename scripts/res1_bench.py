#!/usr/bin/env python
"""The 3x3x3 32->32 layer of the 3-D encoder (RenderNet_Shader.py:51-64) in its three epilogue flavours -- bias+PReLU (conv1 of a
res-block), bias+residual (conv2, *_skip), bias only -- timed with HIP events.  Development tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import ops  # noqa: E402
from scripts.layer_bench import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B, 64, 64, 32, 32), device="cuda", generator=g)
r = torch.randn((B, 64, 64, 32, 32), device="cuda", generator=g)
w = torch.randn((3, 3, 3, 32, 32), device="cuda", generator=g) * 0.05
b = torch.randn(32, device="cuda", generator=g) * 0.1
al = torch.rand(32, device="cuda", generator=g) * 0.25
pw = ops.pack_conv(w)
fl = 2.0 * B * 64 * 64 * 32 * 27 * 32 * 32 * 12.0 / 27.0
with torch.no_grad():
    for name, fn in (("bias+prelu", lambda: ops.conv3d(x, pw, b, al)), ("bias+residual", lambda: ops.conv3d(x, pw, b, None, r)),
                     ("bias", lambda: ops.conv3d(x, pw, b))):
        fn()
        ms = min(timeit(fn, 20) for _ in range(3))
        print("res1 %-14s %.4f ms  %.1f TFLOP/s executed = %.3f of 157.3" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3), flush=True)
