#!/usr/bin/env python
"""Markdown table of the rounding errors of every conv route -- direct, F(2x2,3x3), F(4x4,3x3), F(6x6,3x3), and the last two with
the split (bf16x3) multiply stage -- against a float64 CPU conv on the hostile statistics of tests/test_gpu_wino_robust.py
(same seeds, same cases).   python scripts/robust_report.py > gpurun_out/rNN_wino_robustness.md     (GPU box)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts import robust_util as RU  # noqa: E402

C = 1024
NAMES = {"direct": "direct", "f22": "F(2x2,3x3)", "f43": "F(4x4,3x3)", "f63": "F(6x6,3x3)", "f43s": "F(4x4,3x3) split", "f63s": "F(6x6,3x3) split",
         "f43h": "F(4x4,3x3) split16", "f63h": "F(6x6,3x3) split16"}


def conv_f64(x, w, b=None):
    xn = torch.as_tensor(x).double().permute(0, 3, 1, 2)
    y = F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1), None, 1, 1).permute(0, 2, 3, 1)
    return (y + torch.as_tensor(b).double() if b is not None else y).contiguous()


def main():
    print("errors are max|got - float64 conv| / max|y|; C = %d, map 64x64, batch 1; `split` = the multiply stage on the bf16 pipe" % C)
    print("(three bf16 pieces per fp32 operand, six piece products, fp32 accumulation); bars: 3e-5 (1e-4 for F(6x6,3x3)), the same for both\n")
    print("| case | max abs y | " + " | ".join(NAMES[s] for s in RU.SCHEMES) + " | split / exact (F63) | split16 / exact (F63) |")
    print("|---|---|" + "---|" * (len(RU.SCHEMES) + 2))
    rng = np.random.default_rng(20260926)
    for name, x, w, b in RU.hostile_inputs(rng, 1, 64, 64, C, C):
        want = conv_f64(x, w, b)
        ymax = float(want.abs().max())
        xd, wd, bd = (torch.as_tensor(a).cuda() for a in (x, w, b))
        e = {s: float((RU.conv_with_scheme(xd, wd, bd, s).cpu().double() - want).abs().max()) / ymax for s in RU.SCHEMES}
        print("| %s | %.3g | " % (name, ymax) + " | ".join("%.2e" % e[s] for s in RU.SCHEMES) + " | %.2f | %.2f |" % (e["f63s"] / e["f63"], e["f63h"] / e["f63"]), flush=True)
    rng = np.random.default_rng(7)
    net = RU.res_stack_weights(rng, C, n_blocks=10)
    x0 = (np.abs(rng.standard_normal((1, 24, 24, C))) + 0.5).astype(np.float32)
    want = RU.res_stack_f64(x0, net, conv_f64)
    ymax = float(want.abs().max())
    e = {s: float((RU.res_stack_gpu(torch.as_tensor(x0).cuda(), net, s).cpu().double() - want).abs().max()) / ymax for s in RU.SCHEMES}
    print("| 21 stacked convs (10 res-blocks + skip), 24x24 map | %.3g | " % ymax + " | ".join("%.2e" % e[s] for s in RU.SCHEMES)
          + " | %.2f | %.2f |" % (e["f63s"] / e["f63"], e["f63h"] / e["f63"]))


if __name__ == "__main__":
    main()
