#!/usr/bin/env python
"""Per-layer timing of the Phong-shader path at batch B on one GPU (HIP events), with achieved
TFLOP/s (convs, MAC counting rule of SURVEY.md §8d) or GB/s (resampler).  Development tool."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import ops  # noqa: E402


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--no-dense", action="store_true", help="resampler: skip the random-dense worst case")
    args = ap.parse_args()
    B = args.batch
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape):
        return torch.randn(shape, device=dev, generator=g)

    rows = []

    def conv_case(name, kind, xshape, wshape, stride, count):
        if args.only and args.only not in name:
            return
        x = rnd(*xshape)
        w = rnd(*wshape) * 0.05
        if kind in ("conv3d", "conv2d"):
            pw = ops.pack_conv(w)
            cout = wshape[-1]
        else:
            pw = ops.pack_conv_transpose(w, stride[0])
            cout = wshape[-2]
        b = rnd(cout) * 0.1
        al = torch.rand(cout, device=dev, generator=g) * 0.25
        fn = {"conv3d": lambda: ops.conv3d(x, pw, b, al, stride=stride),
              "conv2d": lambda: ops.conv2d(x, pw, b, al, stride=stride),
              "convT": lambda: ops.conv2d_transpose(x, pw, b, al, stride=stride)}[kind]
        y = fn()
        ms = timeit(fn, args.iters)
        taps = int(np.prod(wshape[:-2]))
        if kind == "convT":
            macs = int(np.prod(xshape[:-1])) * taps * wshape[-1] * wshape[-2]
        else:
            macs = int(np.prod(y.shape[:-1])) * taps * wshape[-2] * wshape[-1]
        tf = 2 * macs / (ms * 1e-3) / 1e12
        rows.append((name, ms, tf, count, ms * count))
        print("%-12s %9.3f ms  %7.2f TFLOP/s  x%-3d = %9.2f ms" % (name, ms, tf, count, ms * count), flush=True)
        del x, w, y

    # resampler
    if not args.only or "resample" in args.only:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import synthetic_batch
        vox_np, pose_np = synthetic_batch(B)                 # the 5 shipped fixtures, bench poses
        vox = torch.as_tensor(vox_np).cuda()
        pose = torch.as_tensor(pose_np).cuda()
        ms = timeit(lambda: ops.resample(vox, pose, 128), max(args.iters, 20))
        gb = B * 9437184 / (ms * 1e-3) / 1e9
        rows.append(("resample", ms, gb, 1, ms))
        print("%-12s %9.3f ms  %7.1f GB/s algorithmic (fixtures)" % ("resample", ms, gb), flush=True)
        if not args.no_dense:
            voxd = (torch.rand((B, 64, 64, 64, 1), device=dev, generator=g) < 0.2).float()
            msd = timeit(lambda: ops.resample(voxd, pose, 128), max(args.iters, 20))
            print("%-12s %9.3f ms  %7.1f GB/s algorithmic (20%% random-dense volume: worst case, every cell occupied)"
                  % ("resample-dense", msd, B * 9437184 / (msd * 1e-3) / 1e9), flush=True)

    conv_case("e_conv1", "conv3d", (B, 128, 128, 128, 1), (5, 5, 5, 1, 8), (2, 2, 2), 1)
    if args.only and "tex" in args.only:          # texture net (RenderNet_Texture_Face_Normal.py): 5 input channels, 16-wide trunk
        conv_case("tex_e_conv1", "conv3d", (B, 128, 128, 128, 5), (5, 5, 5, 5, 8), (2, 2, 2), 1)
        conv_case("tex_res1", "conv3d", (B, 64, 64, 32, 16), (3, 3, 3, 16, 16), (1, 1, 1), 21)
    conv_case("e_conv2", "conv3d", (B, 64, 64, 64, 8), (3, 3, 3, 8, 16), (1, 1, 2), 1)
    conv_case("e_conv3", "conv3d", (B, 64, 64, 32, 16), (3, 3, 3, 16, 32), (1, 1, 1), 1)
    conv_case("res1", "conv3d", (B, 64, 64, 32, 32), (3, 3, 3, 32, 32), (1, 1, 1), 21)
    conv_case("proj1x1", "conv2d", (B, 64, 64, 1024), (1, 1, 1024, 1024), (1, 1), 1)
    conv_case("res2", "conv2d", (B, 64, 64, 1024), (3, 3, 1024, 1024), (1, 1), 21)
    conv_case("e_conv5", "conv2d", (B, 64, 64, 1024), (4, 4, 1024, 512), (1, 1), 1)
    conv_case("res3", "conv2d", (B, 64, 64, 512), (3, 3, 512, 512), (1, 1), 11)
    conv_case("e_conv6", "conv2d", (B, 64, 64, 512), (4, 4, 512, 256), (1, 1), 1)
    conv_case("e_conv7", "convT", (B, 64, 64, 256), (4, 4, 128, 256), (2, 2), 1)
    conv_case("e_conv7_1", "convT", (B, 128, 128, 128), (4, 4, 128, 128), (1, 1), 1)
    conv_case("e_conv8", "convT", (B, 128, 128, 128), (4, 4, 64, 128), (2, 2), 1)
    conv_case("e_conv9", "convT", (B, 256, 256, 64), (4, 4, 32, 64), (2, 2), 1)
    conv_case("e_conv10", "convT", (B, 512, 512, 32), (4, 4, 16, 32), (1, 1), 1)
    conv_case("e_conv11", "convT", (B, 512, 512, 16), (4, 4, 1, 16), (1, 1), 1)
    total = sum(r[4] for r in rows)
    print("sum over layers: %.2f ms per batch of %d  ->  %.2f frames/s" % (total, B, B / total * 1e3))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/layer_bench.json", "w") as f:
        json.dump([{"layer": r[0], "ms": r[1], "rate": r[2], "count": r[3]} for r in rows], f, indent=1)


if __name__ == "__main__":
    main()
