#!/usr/bin/env python
"""rn_epilogue_bwd (bias + PReLU backward: dz, dbias, dalpha) on the training bench's largest layers.  Development tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    for (M, C) in ((24 * 32 * 32, 1024), (24 * 16 * 16, 512), (24 * 32 * 32 * 16, 32), (24 * 128 * 128, 64)):
        g = torch.Generator(device="cuda").manual_seed(0)
        dy = torch.randn((M, C), device="cuda", generator=g)
        z = torch.randn((M, C), device="cuda", generator=g)
        al = torch.rand(C, device="cuda", generator=g) * 0.25
        dz = torch.empty_like(dy)
        db, da = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        f = lambda: lib.rn_epilogue_bwd(L.ptr(dy), L.ptr(z), None, L.ptr(al), L.ptr(dz), L.ptr(db), L.ptr(da), M, C, L.RN_ACT_PRELU, L.stream_ptr())
        for _ in range(3):
            L.check(f(), "epi")
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(f(), "epi")
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("M=%d C=%d  %.4f ms  %.2f TB/s (3 passes of %.0f MB)" % (M, C, best, 3 * M * C * 4 / best / 1e9, M * C * 4 / 1e6), flush=True)


if __name__ == "__main__":
    main()
