#!/usr/bin/env python
"""Two half batches in lockstep on two HIP streams, the GEMM stages strictly alternating (a baton passed through ops.STAGE_HOOK): while
one half's GEMM stage runs on RN_WINO_BF3_GRID CUs, the other half's HBM-bound launches (output transform, next input transform, 3-D
convs ...) take the rest of the chip.  The GEMM stage is power-bound (profiles/r05_gemm_cu_scaling.txt), so CUs taken from it cost less
than their share.  Development tool / measurement.   RN_WINO_BF3_GRID=192 python scripts/pipelined_bench.py [--steps 8]"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rendernet_amd import ops  # noqa: E402


class Baton:
    """GEMM launches of the two workers alternate: worker i may issue its n-th GEMM only after worker 1 - i has issued its
    (n - 1 + i)-th, and the launch waits (on the device) for that GEMM to finish."""

    def __init__(self, streams):
        self.cv = threading.Condition()
        self.turn = 0
        self.last = None                      # event behind the most recent GEMM of either worker
        self.streams = streams
        self.ids = {}

    def hook(self, stage, tkn):
        if stage != "gemm":
            return None
        i = self.ids.get(threading.get_ident())
        if i is None:
            return None
        baton = self

        class Start:
            def record(self_inner):
                with baton.cv:
                    while baton.turn != i:
                        baton.cv.wait()
                    if baton.last is not None:
                        baton.streams[i].wait_event(baton.last)

        class End:
            def record(self_inner):
                ev = torch.cuda.Event()
                ev.record(baton.streams[i])
                with baton.cv:
                    baton.last = ev
                    baton.turn = 1 - i
                    baton.cv.notify_all()
        return Start(), End()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    wl = bench.build_workload("render", "cuda:0")
    vox_np, _, poses_np = wl["inputs"](args.batch)
    vox, poses = torch.as_tensor(vox_np).cuda(), torch.as_tensor(poses_np).cuda()
    h = args.batch // 2
    with torch.no_grad():
        ref = wl["render"](vox, None, poses)
        for sl in (slice(0, h), slice(h, None)):                    # every launch plan of a half batch once (packs, LDS attributes)
            wl["render"](vox[sl], None, poses[sl])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl["render"](vox, None, poses)
        torch.cuda.synchronize()
        single = (time.perf_counter() - t0) / args.steps
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    main_stream = torch.cuda.current_stream()
    outs = [None, None]

    def step():
        baton = Baton(streams)
        ops.STAGE_HOOK = baton.hook

        def work(i):
            baton.ids[threading.get_ident()] = i
            sl = slice(0, h) if i == 0 else slice(h, None)
            with torch.cuda.stream(streams[i]), torch.no_grad():
                outs[i] = wl["render"](vox[sl], None, poses[sl])
        for s in streams:
            s.wait_stream(main_stream)
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for s in streams:
            main_stream.wait_stream(s)
        ops.STAGE_HOOK = None

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    got = torch.cat(outs)
    print("grid %s: single stream %.2f ms/step (%.1f frames/s)   two half batches, alternating GEMM stages %.2f ms/step (%.1f frames/s)   bit-equal: %s"
          % (os.environ.get("RN_WINO_BF3_GRID", "256"), single * 1e3, args.batch / single, dt * 1e3, args.batch / dt, bool(torch.equal(got, ref))), flush=True)


if __name__ == "__main__":
    main()
