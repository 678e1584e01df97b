#!/usr/bin/env python
"""The split (bf16x3) fused 3x3x3 32 -> 32 kernel (csrc/conv3d_wino_bf3.hip) against the fp32 Winograd kernel, the oracle conv and a
float64 conv; and its time next to the fp32 kernel's.  Development tool.   python scripts/c3_check.py [--batch 24]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import ops  # noqa: E402


def conv_f64(x, w):
    xn = F.pad(torch.as_tensor(x).double().permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1))
    return F.conv3d(xn, torch.as_tensor(w).double().permute(4, 3, 0, 1, 2)).permute(0, 2, 3, 4, 1).contiguous()


def accuracy(B, H, W, D, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    lim = np.sqrt(6.0 / (64 * 27))
    w = rng.uniform(-lim, lim, (3, 3, 3, 32, 32)).astype(np.float32)
    b = (0.1 * rng.standard_normal(32)).astype(np.float32)
    al = rng.uniform(0, 0.25, 32).astype(np.float32)
    res = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    want = conv_f64(x, w) + torch.as_tensor(b).double()
    wantp = torch.clamp(want, min=0) + torch.as_tensor(al).double() * torch.clamp(want, max=0) + torch.as_tensor(res).double()
    ymax = float(want.abs().max())
    xd, wd, bd, ad, rd = (torch.as_tensor(a).cuda() for a in (x, w, b, al, res))
    out = {}
    for mode in ("f32", "split", "split16"):
        ops.CONV3D_SPLIT = mode != "f32"
        ops.WINO_GEMM = mode
        pw = ops.pack_conv(wd)
        with torch.no_grad():
            y = ops.conv3d(xd, pw, bd)
            yp = ops.conv3d(xd, pw, bd, ad, rd)
        out[mode] = (float((y.cpu().double() - want).abs().max()) / ymax, float((yp.cpu().double() - wantp).abs().max()) / ymax, y)
    ops.CONV3D_SPLIT, ops.WINO_GEMM = None, "f32"
    print("B=%d %dx%dx%d: f32 %.2e / %.2e   split %.2e / %.2e   split16 %.2e / %.2e   (x max|y| = %.3g)"
          % (B, H, W, D, out["f32"][0], out["f32"][1], out["split"][0], out["split"][1], out["split16"][0], out["split16"][1], ymax), flush=True)


def timing(B, iters):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, 64, 64, 32, 32), device="cuda", generator=g)
    w = torch.randn((3, 3, 3, 32, 32), device="cuda", generator=g) * 0.05
    b = torch.randn(32, device="cuda", generator=g) * 0.1
    al = torch.rand(32, device="cuda", generator=g) * 0.25
    from rendernet_amd import _lib as L
    import ctypes
    ax = torch.zeros(1, dtype=torch.int32, device="cuda")          # max|x| as the producing layer would have left it (format H2)
    L.check(L.lib().rn_absmax(L.ptr(x), x.numel(), ctypes.c_void_p(ax.data_ptr()), L.stream_ptr()), "rn_absmax")
    x._rn_amax = (ax, x._version)
    for mode in ("f32", "split", "split16"):
        ops.CONV3D_SPLIT = mode != "f32"
        ops.WINO_GEMM = mode
        pw = ops.pack_conv(w)
        with torch.no_grad():
            for _ in range(3):
                ops.conv3d(x, pw, b, al)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv3d(x, pw, b, al)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
        fl = 2.0 * B * 64 * 64 * 32 * 32 * 32 * 12          # F(2x2,3x3) over (H,W) x 3 depth taps: 12 products per output
        print("B=%d res1 layer, %-7s %.3f ms  (%.1f TFLOP/s fp32-equivalent executed)" % (B, mode, best, fl / best / 1e9), flush=True)
    ops.CONV3D_SPLIT, ops.WINO_GEMM = None, "f32"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    args = ap.parse_args()
    if not args.no_accuracy:
        accuracy(1, 4, 32, 2)
        accuracy(1, 8, 32, 3, 1)
        accuracy(2, 16, 64, 5, 2)
        accuracy(1, 64, 64, 32, 3)
        accuracy(3, 10, 20, 4, 4)            # ragged: H % 4 != 0, W % 32 != 0
    if not args.no_timing:
        timing(args.batch, args.iters)
