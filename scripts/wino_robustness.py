#!/usr/bin/env python
"""Rounding error of the Winograd schemes on hostile statistics: F(6x6,3x3) against F(4x4,3x3), the
fused F(2x2,3x3) kernel and the direct implicit-GEMM kernel, all through the C ABI, each measured against a FLOAT64 CPU
conv of the same fp32 operands (the rounding-free answer), on the res2 layer shape (3x3, C -> C, 64x64 map).

Input statistics (scripts/robust_util.py: hostile_inputs): N(0,1) (what round 2 tested); |N(0,1)| + 3 (post-PReLU-like
positive mean); log-normal per-channel gains (heavy tails); filters scaled so the outputs reach +-8 and beyond; and a stack
of 21 convs (10 res-blocks + skip, RenderNet_Shader.py:71-84) whose activations are produced by the scheme under test itself.
Prints one markdown table.    usage: python scripts/wino_robustness.py [--channels 1024] [--out file.md]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts import robust_util as robust  # noqa: E402


def conv_f64(x, w, b=None):
    """SAME 3x3 conv in float64 on the CPU: x [B,H,W,C] fp32 values, w [3,3,Cin,Cout]."""
    xn = torch.as_tensor(x).double().permute(0, 3, 1, 2)
    y = F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1), None, 1, 1).permute(0, 2, 3, 1)
    if b is not None:
        y = y + torch.as_tensor(b).double()
    return y.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--stack-hw", type=int, default=24)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    C, hw = args.channels, args.hw
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    lines = ["| case | max abs y | direct | F(2x2,3x3) | F(4x4,3x3) | F(6x6,3x3) | F63 / F43 |",
             "|---|---|---|---|---|---|---|"]
    rng = np.random.default_rng(20260926)
    for name, x, w, b in robust.hostile_inputs(rng, 1, hw, hw, C, C):
        want = conv_f64(x, w, b)
        ymax = float(want.abs().max())
        errs = {}
        for scheme in robust.SCHEMES:
            got = robust.conv_with_scheme(torch.as_tensor(x).cuda(), torch.as_tensor(w).cuda(), torch.as_tensor(b).cuda(), scheme)
            errs[scheme] = float((got.cpu().double() - want).abs().max()) / ymax
        lines.append("| %s | %.3g | %.2e | %.2e | %.2e | %.2e | %.1f |" % (
            name, ymax, errs["direct"], errs["f22"], errs["f43"], errs["f63"], errs["f63"] / max(errs["f43"], 1e-30)))
        print(lines[-1], flush=True)
    # 21 stacked convs: 10 res-blocks + the skip conv, activations produced by the scheme under test (errors compound)
    hs = args.stack_hw
    net = robust.res_stack_weights(rng, C, n_blocks=10)
    x0 = (np.abs(rng.standard_normal((1, hs, hs, C))) + 0.5).astype(np.float32)
    want = robust.res_stack_f64(x0, net, conv_f64)
    ymax = float(want.abs().max())
    errs = {}
    for scheme in robust.SCHEMES:
        got = robust.res_stack_gpu(torch.as_tensor(x0).cuda(), net, scheme)
        errs[scheme] = float((got.cpu().double() - want).abs().max()) / ymax
    lines.append("| 21 stacked convs (10 res-blocks + skip), %dx%d map | %.3g | %.2e | %.2e | %.2e | %.2e | %.1f |" % (
        hs, hs, ymax, errs["direct"], errs["f22"], errs["f43"], errs["f63"], errs["f63"] / max(errs["f43"], 1e-30)))
    print(lines[-1], flush=True)
    out = "errors are max|got - float64 conv| / max|y|; C = %d, map %dx%d, batch 1\n\n" % (C, hw, hw) + "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
