#!/usr/bin/env python
"""Does an HBM-bound Winograd transform launch run in the shadow of the MFMA-bound GEMM stage when the two sit on different
HIP streams?  Times (wall clock around a synchronize) 5 GEMM stages on stream 1, N transforms on stream 2, and both at once.
Development tool (results: profiles/r02e_stream_overlap.txt).  usage: python scripts/overlap_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import _lib as L, ops  # noqa: E402


def main():
    lib = L.lib()
    B, hw, c = 12, 64, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    scheme, nxi, m = L.RN_WINO_F63, 64, 6
    T = B * (-(-hw // m)) ** 2
    w = torch.randn((3, 3, c, c), device="cuda", generator=g) * 0.02
    u = ops.pack_conv(w).wino63
    bufs = []
    for _ in range(2):
        x = torch.randn((B, hw, hw, c), device="cuda", generator=g)
        ws = torch.empty(nxi * T * 2 * c, device="cuda")
        y = torch.empty_like(x)
        bufs.append((x, ws, y))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def gemm(k, st):
        x, ws, y = bufs[k]
        L.check(lib.rn_winograd_gemm(scheme, L.ptr(ws), L.ptr(u), ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * c), T, c, c, st), "gemm")

    def xin(k, st):
        x, ws, y = bufs[k]
        L.check(lib.rn_winograd_input_transform(scheme, L.ptr(x), L.ptr(ws), B, hw, hw, c, 1, st), "input")

    def xout(k, st):
        x, ws, y = bufs[k]
        L.check(lib.rn_winograd_output_transform(scheme, ctypes.c_void_p(ws.data_ptr() + 4 * nxi * T * c), None, None, None, L.ptr(y), None,
                                                 B, hw, hw, c, 0, st), "output")

    def wall(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    p1, p2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
    def tcopy(k, st):
        with torch.cuda.stream(s2):
            bufs[k][2].copy_(bufs[k][0])                   # a torch elementwise copy: a handful of registers per thread

    xs = torch.randn((B, hw, hw, c), device="cuda", generator=g)
    ws43 = torch.empty(36 * (B * 256) * c, device="cuda")

    def xin43(k, st):
        L.check(lib.rn_winograd_input_transform(L.RN_WINO_F43, L.ptr(xs), L.ptr(ws43), B, hw, hw, c, 1, st), "input43")
    for name, tr in (("input transform", xin), ("output transform", xout), ("F43 input transform", xin43), ("torch copy", tcopy)):
        for _ in range(2):
            gemm(0, p1); tr(1, p2)
        NG, NT = 5, 40
        a = min(wall(lambda: [gemm(0, p1) for _ in range(NG)]) for _ in range(3))
        b = min(wall(lambda: [tr(1, p2) for _ in range(NT)]) for _ in range(3))

        def both():
            for i in range(NG):
                gemm(0, p1)
                for _ in range(NT // NG):
                    tr(1, p2)
        cc = min(wall(both) for _ in range(3))
        print("%s: %d GEMM stages alone %.3f ms, %d transforms alone %.3f ms, together %.3f ms (sum %.3f, max %.3f)"
              % (name, NG, a, NT, b, cc, a + b, max(a, b)), flush=True)


if __name__ == "__main__":
    main()
