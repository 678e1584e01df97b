#!/usr/bin/env python
"""Single-frame render latency (BASELINE config 1 shape: one 64^3 binvox -> 512x512x3), eager launches vs hipGraph
replay (Renderer.capture).  Development tool."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights  # noqa: E402


def main():
    spec = ShaderSpec(out_ch=3).check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234))
    for B in (1, 2, 4):
        vox, poses = synthetic_batch(B)
        vox_d, pose_d = torch.as_tensor(vox).cuda(), torch.as_tensor(poses).cuda()
        with torch.no_grad():
            for _ in range(3):
                r.render(vox_d, pose_d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                r.render(vox_d, pose_d)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 10
        replay = r.capture(B)
        for _ in range(3):
            replay(vox_d, pose_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            replay(vox_d, pose_d)
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 10
        print("B=%d  eager %.2f ms  hipGraph %.2f ms  (%.1f / %.1f frames/s)" % (B, eager * 1e3, graph * 1e3, B / eager, B / graph), flush=True)


if __name__ == "__main__":
    main()
