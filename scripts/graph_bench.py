#!/usr/bin/env python
"""Eager launches vs hipGraph replay (Renderer.capture) of the render step at batch B.  Development tool."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = 5
    spec = ShaderSpec().check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True))
    vox_np, poses_np = synthetic_batch(B)
    vox, poses = torch.as_tensor(vox_np).cuda(), torch.as_tensor(poses_np).cuda()
    with torch.no_grad():
        for _ in range(2):
            out = r.render(vox, poses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = r.render(vox, poses)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / steps
        replay = r.capture(B)
        g = replay(vox, poses)
        torch.cuda.synchronize()
        assert torch.equal(g, out)
        t0 = time.perf_counter()
        for _ in range(steps):
            replay.graph.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / steps
    print("B=%d  eager %.3f ms/step (%.2f fps)   hipGraph replay %.3f ms/step (%.2f fps)" % (B, eager * 1e3, B / eager, graph * 1e3, B / graph))


if __name__ == "__main__":
    main()
