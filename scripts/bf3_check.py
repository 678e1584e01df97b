#!/usr/bin/env python
"""Split (bf16x3) GEMM stage of the three-launch Winograd path against the exact-fp32 route and a float64 conv, and
per-stage timings of both (HIP events).  Development tool.
    python scripts/bf3_check.py [--quick] [--batch 24] [--iters 5]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L, ops  # noqa: E402


def conv_f64(x, w, pads):
    xn = torch.as_tensor(x).double().permute(0, 3, 1, 2)
    xn = F.pad(xn, (pads[0], pads[1], pads[0], pads[1]))
    return F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1)).permute(0, 2, 3, 1).contiguous()


def accuracy(B, hw, cin, cout, k, seed=0, forced=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, hw, hw, cin)).astype(np.float32)
    lim = np.sqrt(6.0 / ((cin + cout) * k * k))
    w = rng.uniform(-lim, lim, (k, k, cin, cout)).astype(np.float32)
    want = conv_f64(x, w, (1, 1) if k == 3 else (1, 2))
    ymax = float(want.abs().max())
    xd, wd = torch.as_tensor(x).cuda(), torch.as_tensor(w).cuda()
    out = {}
    for mode in ("f32", "split", "split16"):
        ops.WINO_GEMM = mode
        pw = ops.pack_conv(wd)
        if forced:
            pw.force_scheme = forced
        with torch.no_grad():
            y = ops.conv2d(xd, pw)
        out[mode] = float((y.cpu().double() - want).abs().max()) / ymax
        out[mode + "_y"] = y
    d = float((out["f32_y"] - out["split_y"]).abs().max()) / ymax
    print("B=%d %dx%d %d->%d k%d %s: f32 %.2e  split (bf16x3) %.2e  split16 (fp16x2) %.2e  |f32-split| %.2e  (x max|y| = %.3g)"
          % (B, hw, hw, cin, cout, k, forced or ops._wino_scheme(pw, hw, hw), out["f32"], out["split"], out["split16"], d, ymax), flush=True)
    ops.WINO_GEMM = "f32"
    return out["f32"], out["split"]


def timing(B, hw, cin, cout, k, iters, forced=None):
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, hw, hw, cin), device="cuda", generator=g)
    w = torch.randn((k, k, cin, cout), device="cuda", generator=g) * 0.02
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    pw = ops.pack_conv(w)
    which = forced or ops._wino_scheme(pw, hw, hw)
    scheme, nxi, m = {"f43": (L.RN_WINO_F43, 36, 4), "f44": (L.RN_WINO_F44, 49, 4), "f63": (L.RN_WINO_F63, 64, 6)}[which]
    T = B * (-(-hw // m)) ** 2
    st = L.stream_ptr()
    y = torch.empty((B, hw, hw, cout), device="cuda")
    res = {}
    for mode in ("f32", "split", "split16"):
        if mode == "f32":
            u = pw.wino63 if which == "f63" else pw.wino43
            ws = torch.empty(nxi * T * (cin + cout), device="cuda")
            V, M = L.ptr(ws), ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * cin)
            stages = [
                lambda: L.check(lib.rn_winograd_input_transform(scheme, L.ptr(x), V, B, hw, hw, cin, 1, st), "input"),
                lambda: L.check(lib.rn_winograd_gemm(scheme, V, L.ptr(u), M, T, cin, cout, st), "gemm"),
                lambda: L.check(lib.rn_winograd_output_transform(scheme, M, L.ptr(b), None, None, L.ptr(y), None, B, hw, hw, cout, 0, st), "output"),
            ]
        else:
            fmt = L.RN_SPLIT_FMT_H2 if mode == "split16" else 0
            sf = scheme | fmt
            us = ctypes.c_void_p(pw.split(which, fmt).data_ptr())
            ws = torch.empty(lib.rn_winograd_split_workspace_bytes(sf, B, hw, hw, cin, cout), dtype=torch.uint8, device="cuda")
            V = ctypes.c_void_p(ws.data_ptr())
            M = ctypes.c_void_p(ws.data_ptr() + lib.rn_winograd_split_v_bytes(sf, T, cin))
            stages = [
                lambda sf=sf, V=V: L.check(lib.rn_winograd_split_input_transform(sf, L.ptr(x), V, B, hw, hw, cin, 1, st), "input"),
                lambda sf=sf, V=V, us=us, M=M: L.check(lib.rn_winograd_split_gemm(sf, V, us, M, T, cin, cout, st), "gemm"),
                lambda M=M: L.check(lib.rn_winograd_output_transform(scheme, M, L.ptr(b), None, None, L.ptr(y), None, B, hw, hw, cout, 0, st), "output"),
            ]
        for f in stages * 2:
            f()
        torch.cuda.synchronize()
        best = [1e9] * 3
        for _ in range(iters):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            evs[0].record()
            for i, f in enumerate(stages):
                f()
                evs[i + 1].record()
            torch.cuda.synchronize()
            best = [min(best[i], evs[i].elapsed_time(evs[i + 1])) for i in range(3)]
        fl = 2.0 * nxi * T * cin * cout
        if mode == "f32":
            print("  %-5s input %.3f ms  gemm %.3f ms (%.1f TFLOP/s fp32-equivalent, %.3f of the fp32 MFMA peak)  output %.3f ms  total %.3f ms"
                  % (mode, best[0], best[1], fl / best[1] / 1e9, fl / best[1] / 1e9 / 157.3, best[2], sum(best)), flush=True)
        else:
            npr = 3 if mode == "split16" else 6
            print("  %-7s input %.3f ms  gemm %.3f ms (%.1f TFLOP/s fp32-equivalent; %d %s products: %.0f TFLOP/s = %.3f of the 16-bit peak)  output %.3f ms  total %.3f ms"
                  % (mode, best[0], best[1], fl / best[1] / 1e9, npr, "fp16" if npr == 3 else "bf16", npr * fl / best[1] / 1e9,
                     npr * fl / best[1] / 1e9 / 2500.0, best[2], sum(best)), flush=True)
        res[mode] = best
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--shapes", type=str, default="", help="indices into the timing shape list, e.g. 0,2")
    args = ap.parse_args()
    if not args.no_accuracy:
        accuracy(1, 24, 256, 256, 3, forced="f43")
        accuracy(1, 24, 256, 256, 3, forced="f63")
        accuracy(2, 30, 64, 512, 3, forced="f63")           # ragged planes, several K steps only
        accuracy(3, 16, 512, 256, 4)
        accuracy(5, 64, 256, 256, 3)                         # T = 605: two whole row blocks + a ragged one
        accuracy(3, 64, 256, 512, 3)                         # T = 363: walked in 128-row items
        accuracy(9, 64, 64, 256, 3)                          # T = 1089: 4 whole blocks x 64 xi = 256 items + ragged 65 rows
        accuracy(2, 64, 64, 256, 4)                          # F44: T = 512, 98 items (less than one round)
        if not args.quick:
            accuracy(1, 64, 1024, 1024, 3)
            accuracy(1, 64, 1024, 1024, 3, forced="f43")
            accuracy(1, 64, 1024, 512, 4)
    if not args.no_timing:
        B = args.batch
        shapes = ((64, 1024, 1024, 3, None), (64, 512, 512, 3, None), (64, 1024, 512, 4, None), (64, 512, 256, 4, None), (64, 1024, 1024, 3, "f43"), (64, 512, 512, 3, "f43"))
        if args.shapes:
            shapes = tuple(shapes[int(i)] for i in args.shapes.split(","))
        for (hw, cin, cout, k, forced) in shapes:
            print("B=%d %dx%d %d->%d k%d %s" % (B, hw, hw, cin, cout, k, forced or ""), flush=True)
            timing(B, hw, cin, cout, k, args.iters, forced)


if __name__ == "__main__":
    main()
