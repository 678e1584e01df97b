"""Resampler timing and bit check on one MI355X: rn_resample_fwd on the bench batch (24 x 64^3 -> 128^3, C = 1, image layout),
the stress batch (8 x 128^3 -> 256^3), a C = 4 batch and a crop window; HIP events over back-to-back calls, and a SHA-1 of every
output so that two runs (RN_RS_V1=1: the three-launch form; default: the two-launch form) can be compared bit for bit.
    python scripts/resample_bench.py [--iters 50]"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rendernet_amd import ops  # noqa: E402


def run(name, vox, poses, n, window, iters):
    import ctypes
    from rendernet_amd import _lib as L
    v = torch.as_tensor(vox[..., None] if vox.ndim == 4 else vox).cuda().contiguous()
    p = torch.as_tensor(poses).cuda()
    B, S, C = v.shape[0], v.shape[1], v.shape[4]
    h0, w0, ph, pw = window if window is not None else (0, 0, n, n)
    lib = L.lib()
    out = torch.empty((B, ph, pw, n, C), device="cuda")
    nws = int(lib.rn_resample_workspace_bytes(B, S, C))
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")

    def call():
        L.check(lib.rn_resample_fwd(L.ptr(v), L.ptr(p), L.ptr(out), B, S, n, C, h0, w0, ph, pw, 1, ctypes.c_void_p(ws.data_ptr()), nws,
                                    L.stream_ptr()), "rn_resample_fwd")

    out.fill_(7.0)                                             # every element must be written by the call
    call()
    torch.cuda.synchronize()
    assert torch.equal(out, ops.resample(v, p, n, window=window))
    sha = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    # the C entry called back to back (ctypes + two or three launches per call: the host stays ahead of a ~50 us call)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(5):
        call()
    ev[0].record()
    for i in range(iters):
        call()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    nbytes = 4.0 * (v.numel() + out.numel())
    med = ts[len(ts) // 2]
    print("%-34s out %-22s median %7.1f us  min %7.1f us  %6.0f GB/s (%.3f of 8 TB/s)  nonzero %.4f  sha1 %s"
          % (name, tuple(out.shape), med, ts[0], nbytes / med / 1e3, nbytes / med / 1e3 / 8000.0, float((out != 0).float().mean()), sha))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    print("# scripts/resample_bench.py, RN_RS_V1=%s" % os.environ.get("RN_RS_V1", "(unset: two launches)"))
    vox, poses = bench.synthetic_batch(24)
    run("bench batch 24 x 64^3 -> 128^3", vox, poses, 128, None, a.iters)
    run("crop 64 window (31, 17)", vox, poses, 128, (31, 17, 64, 64), a.iters)
    rng = np.random.default_rng(0)
    v4 = (vox[:6].reshape(6, 64, 64, 64, 1) * rng.uniform(0.2, 1.0, (6, 1, 1, 1, 4))).astype(np.float32)
    run("C = 4, 6 x 64^3 -> 128^3", v4, poses[:6], 128, None, a.iters)
    vox2, poses2 = bench.synthetic_batch(8, upsample=2)
    run("stress 8 x 128^3 -> 256^3", vox2, poses2, 256, None, max(10, a.iters // 5))
    dense = rng.uniform(0.0, 1.0, (4, 64, 64, 64, 1)).astype(np.float32)
    run("dense random 4 x 64^3 -> 128^3", dense, poses[:4], 128, None, a.iters)


if __name__ == "__main__":
    main()
