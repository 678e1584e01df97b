#!/usr/bin/env python
"""A/B timing of the stride-1 3x3 2-D convs: Winograd F(2x2,3x3) kernel vs the direct implicit-GEMM kernel (HIP events).
TFLOP/s are quoted on the DIRECT-equivalent FLOPs (2*M*9*Cin*Cout) for both, and on executed MFMA FLOPs
(2*(M/4)*16*Cin*Cout) for the Winograd kernel.  Development tool."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import ops  # noqa: E402
from scripts.layer_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--shapes", type=str, default="64x1024,64x512")
    ap.add_argument("--wino-only", action="store_true")
    ap.add_argument("--only", choices=["f63", "f43", "f23", "direct"], default=None, help="time just one implementation")
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    for sh in args.shapes.split(","):
        hw, c = (int(v) for v in sh.split("x"))
        B = args.batch
        x = torch.randn((B, hw, hw, c), device="cuda", generator=g)
        w = torch.randn((3, 3, c, c), device="cuda", generator=g) * 0.02
        b = torch.randn(c, device="cuda", generator=g) * 0.1
        al = torch.rand(c, device="cuda", generator=g) * 0.25
        pw = ops.pack_conv(w)
        flop = 2.0 * B * hw * hw * 9 * c * c
        res = {}
        if args.only in ("f23", "direct"):
            pw.wino43 = None
        if args.only == "direct":
            pw.wino = None
        if args.only == "f43":
            pw.wino63 = None
        if pw._wino63_kind is not None and ops._wino_scheme(pw, hw, hw) == "f63":
            ms = timeit(lambda: ops.conv2d(x, pw, b, al), args.iters)
            y63 = ops.conv2d(x, pw, b, al)
            T = B * (-(-hw // 6)) ** 2
            print("%s B=%d  F(6x6,3x3) %8.3f ms  %7.2f TFLOP/s direct-equivalent, %7.2f TFLOP/s executed (GEMM stage FLOPs / whole time)"
                  % (sh, B, ms, flop / ms / 1e9, 2.0 * 64 * T * c * c / ms / 1e9), flush=True)
            if args.only == "f63":
                continue
            pw.wino63 = None
            y43 = ops.conv2d(x, pw, b, al)
            print("   max|F63-F43| = %.3g (max|y| %.3g)" % (float((y63 - y43).abs().max()), float(y43.abs().max())), flush=True)
        elif args.only == "f63":
            continue
        if pw.wino43 is not None:
            ms = timeit(lambda: ops.conv2d(x, pw, b, al), args.iters)
            y43 = ops.conv2d(x, pw, b, al)
            print("%s B=%d  F(4x4,3x3) %8.3f ms  %7.2f TFLOP/s direct-equivalent, %7.2f TFLOP/s executed (GEMM stage FLOPs / whole time)"
                  % (sh, B, ms, flop / ms / 1e9, flop / 4.0 / ms / 1e9), flush=True)
            if args.only == "f43":
                continue
            pw.wino43 = None
            y23 = ops.conv2d(x, pw, b, al)
            print("   max|F43-F23| = %.3g (max|y| %.3g)" % (float((y43 - y23).abs().max()), float(y23.abs().max())), flush=True)
        if pw.wino is not None:
            ms = timeit(lambda: ops.conv2d(x, pw, b, al), args.iters)
            yw = ops.conv2d(x, pw, b, al)
            res["wino"] = ms
            print("%s B=%d  winograd %8.3f ms  %7.2f TFLOP/s direct-equivalent, %7.2f TFLOP/s executed"
                  % (sh, B, ms, flop / ms / 1e9, flop / 2.25 / ms / 1e9), flush=True)
        if args.wino_only or args.only == "f23":
            continue
        wn = pw.wino
        pw.wino = None
        pw.wino43 = None
        ms = timeit(lambda: ops.conv2d(x, pw, b, al), args.iters)
        yd = ops.conv2d(x, pw, b, al)
        print("%s B=%d  direct   %8.3f ms  %7.2f TFLOP/s" % (sh, B, ms, flop / ms / 1e9), flush=True)
        if wn is not None:
            print("   max|wino-direct| = %.3g (max|y| %.3g)" % (float((yw - yd).abs().max()), float(yd.abs().max())), flush=True)


if __name__ == "__main__":
    main()
