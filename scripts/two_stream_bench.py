#!/usr/bin/env python
"""Render bench with the batch cut into N parts that run on N HIP streams (frames are independent): the HBM-bound stages of
one part (Winograd transforms, stem, tail) can run in the shadow of the MFMA-bound GEMM stage of another.  Development tool.
usage: python scripts/two_stream_bench.py --parts 1,2,3 [--steps 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--parts", type=str, default="1,2,3")
    ap.add_argument("--mode", type=str, default="render")
    args = ap.parse_args()
    wl = bench.build_workload(args.mode, "cuda:0")
    vox_np, aux_np, poses_np = wl["inputs"](args.batch)
    vox, poses = torch.as_tensor(vox_np).cuda(), torch.as_tensor(poses_np).cuda()
    aux = torch.as_tensor(aux_np).cuda() if aux_np is not None else None
    main_stream = torch.cuda.current_stream()
    with torch.no_grad():
        ref = wl["render"](vox, aux, poses)               # builds every pack on the main stream
        torch.cuda.synchronize()
        for n in [int(v) for v in args.parts.split(",")]:
            streams = [torch.cuda.Stream() for _ in range(n)]
            cuts = [(args.batch * i) // n for i in range(n + 1)]

            def step():
                outs = []
                for i, s in enumerate(streams):
                    s.wait_stream(main_stream)
                    with torch.cuda.stream(s):
                        a, b = cuts[i], cuts[i + 1]
                        outs.append(wl["render"](vox[a:b], aux[a:b] if aux is not None else None, poses[a:b]))
                for s in streams:
                    main_stream.wait_stream(s)
                return outs
            for _ in range(2):
                outs = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                outs = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            got = torch.cat(outs)
            print("parts=%d  %.2f ms/step  %.1f frames/s  bit-equal to the single call: %s  max|d| %.3g"
                  % (n, dt * 1e3, args.batch / dt, bool(torch.equal(got, ref)), float((got - ref).abs().max())), flush=True)


if __name__ == "__main__":
    main()
