#!/usr/bin/env python
"""Per-stage timing (HIP events) of the three-launch Winograd path -- input transform, GEMM stage, output transform -- for
F(4x4,3x3) and F(6x6,3x3) on the same layer, through the stage entry points of the C ABI, plus the max difference of the two
results.  Development tool.   usage: python scripts/wino_stage_bench.py --shapes 64x1024,128x512 --batch 24"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--shapes", type=str, default="64x1024,128x512")
    args = ap.parse_args()
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    for sh in args.shapes.split(","):
        hw, c = (int(v) for v in sh.split("x"))
        B = args.batch
        x = torch.randn((B, hw, hw, c), device="cuda", generator=g)
        w = torch.randn((3, 3, c, c), device="cuda", generator=g) * 0.02
        b = torch.randn(c, device="cuda", generator=g) * 0.1
        al = torch.rand(c, device="cuda", generator=g) * 0.25
        res = torch.randn((B, hw, hw, c), device="cuda", generator=g)
        pw = ops.pack_conv(w)
        outs = {}
        for name, scheme, nxi, m, u in (("F(4x4,3x3)", L.RN_WINO_F43, 36, 4, pw.wino43), ("F(6x6,3x3)", L.RN_WINO_F63, 64, 6, pw.wino63)):
            T = B * (-(-hw // m)) ** 2
            ws = torch.empty(nxi * T * 2 * c, device="cuda")
            y = torch.empty_like(x)
            V, M = L.ptr(ws), ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * c)
            st = L.stream_ptr()
            stages = [
                lambda: L.check(lib.rn_winograd_input_transform(scheme, L.ptr(x), V, B, hw, hw, c, 1, st), "input"),
                lambda: L.check(lib.rn_winograd_gemm(scheme, V, L.ptr(u), M, T, c, c, st), "gemm"),
                lambda: L.check(lib.rn_winograd_output_transform(scheme, M, L.ptr(b), L.ptr(al), L.ptr(res), L.ptr(y), None, B, hw, hw, c, 1, st), "output"),
            ]
            for f in stages * 2:
                f()
            torch.cuda.synchronize()
            best = [1e9] * 3
            for _ in range(args.iters):
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                evs[0].record()
                for i, f in enumerate(stages):
                    f()
                    evs[i + 1].record()
                torch.cuda.synchronize()
                best = [min(best[i], evs[i].elapsed_time(evs[i + 1])) for i in range(3)]
            fl = 2.0 * nxi * T * c * c
            inb = 4.0 * (B * hw * hw * c + nxi * T * c)
            outb = 4.0 * (nxi * T * c + 2 * B * hw * hw * c)
            print("%s B=%d %s T=%d: input %.3f ms (%.2f TB/s)  gemm %.3f ms (%.1f TFLOP/s, %.3f of peak)  output %.3f ms (%.2f TB/s)  total %.3f ms"
                  % (sh, B, name, T, best[0], inb / best[0] / 1e9, best[1], fl / best[1] / 1e9, fl / best[1] / 1e9 / 157.3,
                     best[2], outb / best[2] / 1e9, sum(best)), flush=True)
            outs[name] = y
        a, bb = outs["F(4x4,3x3)"], outs["F(6x6,3x3)"]
        print("   max|F63 - F43| = %.3g (max|y| %.3g)" % (float((a - bb).abs().max()), float(a.abs().max())), flush=True)


if __name__ == "__main__":
    main()
