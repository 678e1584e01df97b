#!/usr/bin/env python
"""Fused output+input Winograd transform (rn_winograd_output_input_transform) against the two separate launches on the res2 /
res3 shapes: HIP-event times and effective bandwidth on the algorithmic bytes.  Development tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402
from scripts.layer_bench import timeit  # noqa: E402

lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for (H, C) in ((64, 1024), (64, 512)):
    nxi, m, scheme = 64, 6, L.RN_WINO_F63
    th = -(-H // m)
    T = B * th * th
    g = torch.Generator(device="cuda").manual_seed(0)
    M = torch.randn((nxi, T, C), device="cuda", generator=g)
    V = torch.empty((nxi, T, C), device="cuda")
    y = torch.empty((B, H, H, C), device="cuda")
    res = torch.randn((B, H, H, C), device="cuda", generator=g)
    bias = torch.randn(C, device="cuda", generator=g)
    al = torch.rand(C, device="cuda", generator=g) * 0.25
    st = L.stream_ptr()
    out1 = lambda: L.check(lib.rn_winograd_output_transform(scheme, L.ptr(M), L.ptr(bias), L.ptr(al), None, L.ptr(y), None, B, H, H, C, 1, st), "o")
    out2 = lambda: L.check(lib.rn_winograd_output_transform(scheme, L.ptr(M), L.ptr(bias), None, L.ptr(res), L.ptr(y), None, B, H, H, C, 0, st), "o")
    inp = lambda: L.check(lib.rn_winograd_input_transform(scheme, L.ptr(y), L.ptr(V), B, H, H, C, 1, st), "i")
    f1 = lambda: L.check(lib.rn_winograd_output_input_transform(scheme, L.ptr(M), L.ptr(bias), L.ptr(al), None, None, L.ptr(V), B, H, H, C, 1, st), "f")
    f2 = lambda: L.check(lib.rn_winograd_output_input_transform(scheme, L.ptr(M), L.ptr(bias), None, L.ptr(res), L.ptr(y), L.ptr(V), B, H, H, C, 0, st), "f")
    mb, yb = 4.0 * nxi * T * C, 4.0 * B * H * H * C
    for name, fn, by in (("out (prelu)", out1, mb + yb), ("out (+res)", out2, mb + 2 * yb), ("in", inp, yb + mb),
                         ("fused conv1->conv2 (no y)", f1, 2 * mb), ("fused conv2->conv1 (+res, y)", f2, 2 * mb + 2 * yb)):
        fn()
        ms = min(timeit(fn, 10) for _ in range(3))
        print("%dx%dx%d B=%d  %-30s %.4f ms  %.2f TB/s algorithmic" % (H, H, C, B, name, ms, by / ms / 1e9), flush=True)
