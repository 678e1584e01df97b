#!/usr/bin/env python
"""Timing of the wgrad / dgrad / epilogue-backward kernels on the layer shapes of the training step
(batch B, crop 64 -> trunk 32x32).  Development tool."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402
from scripts.layer_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    B = args.batch
    g = torch.Generator(device="cuda").manual_seed(0)
    lib = L.lib()

    def rnd(*s):
        return torch.randn(s, device="cuda", generator=g)

    def wg2d(name, H, W, Cin, Cout, k):
        x, dz = rnd(B, H, W, Cin), rnd(B, H, W, Cout)
        dw = torch.zeros(k, k, Cin, Cout, device="cuda")
        fn = lambda: L.check(lib.rn_conv2d_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, Cin, Cout, L.ivec([k, k]),
                                                 L.ivec([1, 1]), L.stream_ptr()), name)
        ms = timeit(fn, args.iters)
        print("%-14s %8.3f ms %7.1f TFLOP/s" % (name, ms, 2.0 * B * H * W * k * k * Cin * Cout / ms / 1e9), flush=True)

    def wg3d(name, H, W, D, Cin, Cout, k):
        x, dz = rnd(B, H, W, D, Cin), rnd(B, H, W, D, Cout)
        dw = torch.zeros(k, k, k, Cin, Cout, device="cuda")
        fn = lambda: L.check(lib.rn_conv3d_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), B, H, W, D, Cin, Cout, L.ivec([k, k, k]),
                                                 L.ivec([1, 1, 1]), L.stream_ptr()), name)
        ms = timeit(fn, args.iters)
        print("%-14s %8.3f ms %7.1f TFLOP/s" % (name, ms, 2.0 * B * H * W * D * k ** 3 * Cin * Cout / ms / 1e9), flush=True)

    def epi(name, M, C):
        dy, z, al = rnd(M, C), rnd(M, C), torch.rand(C, device="cuda")
        dz, db, da = torch.empty_like(dy), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        fn = lambda: L.check(lib.rn_epilogue_bwd(L.ptr(dy), L.ptr(z), None, L.ptr(al), L.ptr(dz), L.ptr(db), L.ptr(da),
                                                 M, C, 1, L.stream_ptr()), name)
        ms = timeit(fn, 20)
        print("%-14s %8.3f ms %7.1f GB/s" % (name, ms, 3.0 * M * C * 4 / ms / 1e6), flush=True)

    on = lambda n: (not args.only) or args.only in n
    if on("res2"):
        wg2d("wgrad res2", 32, 32, 1024, 1024, 3)
    if on("res3"):
        wg2d("wgrad res3", 32, 32, 512, 512, 3)
    if on("e_conv5"):
        wg2d("wgrad e_conv5", 32, 32, 1024, 512, 4)
    if on("proj"):
        wg2d("wgrad proj", 32, 32, 1024, 1024, 1)
    if on("res1"):
        wg3d("wgrad res1", 32, 32, 32, 32, 32, 3)
    if on("epi"):
        epi("epi 1024", B * 32 * 32, 1024)
        epi("epi 32", B * 32 * 32 * 32, 32)


if __name__ == "__main__":
    main()
