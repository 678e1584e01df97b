#!/usr/bin/env python
"""Time the GEMM stage of the Winograd F(4x4,3x3) path on its own (rn_wino43_gemm, HIP events).  Development tool."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="6144x1024x1024,1536x1024x1024,1536x512x512,8192x2048x2048")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    for sh in args.shapes.split(","):
        T, Cin, Cout = (int(v) for v in sh.split("x"))
        V = torch.randn(36 * T * Cin, device="cuda")
        U = torch.randn(36 * Cin * Cout, device="cuda") * 0.02
        M = torch.empty(36 * T * Cout, device="cuda")
        def run():
            L.check(L.lib().rn_winograd_gemm(L.RN_WINO_F43, L.ptr(V), L.ptr(U), L.ptr(M), T, Cin, Cout, L.stream_ptr()), "gemm")
        run(); run()
        torch.cuda.synchronize()
        evs = []
        for _ in range(args.iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ms = min(a.elapsed_time(b) for a, b in evs)
        fl = 2.0 * 36 * T * Cin * Cout
        print("T=%d %d->%d  %.3f ms  %.1f TFLOP/s (%.3f of 157.3)" % (T, Cin, Cout, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3), flush=True)


if __name__ == "__main__":
    main()
