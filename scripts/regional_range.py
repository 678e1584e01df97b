#!/usr/bin/env python
"""Dynamic range ACROSS REGIONS of one activation tensor: the res2 conv (3x3, C -> C, 64x64 map;
tools/layer_util.py:91-105 -- no normalisation layer bounds the range between a bright region and a dim one) fed a unit floor with ONE 8x8
patch scaled by 2^10 / 2^15 / 2^20, through every three-launch route -- exact fp32, bf16x3 split, fp16x2 split -- against a float64 conv of
the same fp32 operands.  Two numbers per route: the max-norm error max|err| / max|y| over the whole map (what profiles/*wino_robustness.md
reports), and the REGIONAL error: max|err| over the outputs at least 10 pixels away from the patch (no conv tap and no Winograd tile of
theirs touches it) divided by the largest output magnitude IN THAT REGION.  Exact fp32 and bf16x3 work tile-locally, so their regional
error cannot depend on the patch; fp16x2 scales the whole tensor by ONE power of two taken from max|x|, and values 2^-18 below the scaled
maximum lose relative precision.  Prints a markdown table.   usage: python scripts/regional_range.py [--channels 1024] [--out file.md]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts import robust_util as robust  # noqa: E402

ROUTES = (("f63", "F(6x6,3x3) exact"), ("f63s", "F(6x6,3x3) bf16x3"), ("f63h", "F(6x6,3x3) fp16x2"),
          ("f43", "F(4x4,3x3) exact"), ("f43s", "F(4x4,3x3) bf16x3"), ("f43h", "F(4x4,3x3) fp16x2"))


def conv_f64(x, w, b):
    xn = torch.as_tensor(x).double().permute(0, 3, 1, 2)
    y = F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1), None, 1, 1).permute(0, 2, 3, 1)
    return (y + torch.as_tensor(b).double()).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    C, hw = args.channels, args.hw
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rng = np.random.default_rng(20260927)
    w = robust.xavier(rng, (3, 3, C, C))
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    base = (np.abs(rng.standard_normal((1, hw, hw, C))) + 0.5).astype(np.float32)          # the unit floor (post-PReLU-like)
    p0 = 24                                                                                   # patch rows / columns p0 .. p0 + 7
    far = np.ones((hw, hw), bool)
    far[max(0, p0 - 10):p0 + 8 + 10, max(0, p0 - 10):p0 + 8 + 10] = False
    head = "| patch gain | " + " | ".join("%s: whole map / far region" % n for _, n in ROUTES) + " |"
    lines = [head, "|---|" + "---|" * len(ROUTES)]
    worst = {}
    for gain_log2 in (0, 10, 15, 20):
        x = base.copy()
        x[0, p0:p0 + 8, p0:p0 + 8, :] *= np.float32(2.0 ** gain_log2)
        want = conv_f64(x, w, b)[0].numpy()
        ymax, yfar = np.abs(want).max(), np.abs(want[far]).max()
        cells = []
        for scheme, _ in ROUTES:
            got = robust.conv_with_scheme(torch.as_tensor(x).cuda(), torch.as_tensor(w).cuda(), torch.as_tensor(b).cuda(), scheme)
            err = np.abs(got[0].cpu().double().numpy() - want)
            eg, ef = err.max() / ymax, err[far].max() / yfar
            worst[(scheme, gain_log2)] = (eg, ef)
            cells.append("%.1e / %.1e" % (eg, ef))
        lines.append("| 2^%d | " % gain_log2 + " | ".join(cells) + " |")
        print(lines[-1], flush=True)
    ratio = {s: worst[(s, 20)][1] / worst[("f63" if s.startswith("f63") else "f43", 20)][1] for s, _ in ROUTES}
    note = ("far-region error at gain 2^20 relative to the exact route of the same scheme: " +
            ", ".join("%s %.1fx" % (n, ratio[s]) for s, n in ROUTES if s[-1] in "sh"))
    out = ("res2 shape: 3x3, C = %d, %dx%d map, batch 1; unit floor |N(0,1)| + 0.5 with one 8x8 patch x gain; errors vs a float64 conv of the same fp32 "
           "operands: max|err| / max|y| over the whole map, and max|err| over the outputs >= 10 pixels from the patch / max|y| of THAT region\n\n"
           % (C, hw, hw) + "\n".join(lines) + "\n\n" + note + "\n")
    if args.out:
        with open(args.out, "w") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
