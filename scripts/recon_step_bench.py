#!/usr/bin/env python
"""One inverse-rendering step at the reference sizes (Reconstruct_RenderNet_Face.py:334-413: five hypotheses, 64^3 -> 128^3 -> 512^2), timed; run it
under rocprofv3 --kernel-trace --stats for the per-kernel table.  Development tool (GPU box)."""
import math, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import reconstruct as RC

rng = np.random.default_rng(1)
rec = RC.Reconstructor(batch_size=5)
rec.assign(vector=np.full((5, 200), 0.5, np.float32), param=RC.create_param_center(5, 270, 60, 90, 30),
           texture=rng.standard_normal((5, 199)).astype(np.float32), light=(np.linspace(230, 320, num=5) * math.pi / 180.0)[:, None])
target = torch.from_numpy(rng.uniform(0, 1, (5, 512, 512, 3)).astype(np.float32)).cuda()
for _ in range(2):
    rec.step(target)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = rec.step(target)
torch.cuda.synchronize()
print("inverse-rendering step, 5 hypotheses: %.2f ms" % ((time.perf_counter() - t0) / n * 1e3), loss.cpu().numpy()[:2])
