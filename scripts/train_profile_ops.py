#!/usr/bin/env python
"""Which torch ops (and from which Python lines) put small kernels / copies into the training step?  torch.profiler over two steps, grouped by op and
by the innermost rendernet_amd / tools frame.  Development tool (GPU box)."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from rendernet_amd.shader import ShaderSpec, init_shader_weights
from rendernet_amd.train import Trainer

spec = ShaderSpec().check()
tr = Trainer(spec, init_shader_weights(spec, seed=1234, perturb=True), device="cuda:0")
vox_np, poses_np = bench.synthetic_batch(24)
vox, poses = torch.as_tensor(vox_np).cuda(), torch.as_tensor(poses_np).cuda()
targets = torch.rand((24, 512, 512, spec.out_ch), device="cuda")
for i in range(2):
    tr.step(vox, poses, targets, patch_size=64, start_point=(10, 20))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(vox, poses, targets, patch_size=64, start_point=(10, 20))
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::index", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense"):
        frame = next((f for f in (ev.stack or []) if "rendernet_amd" in f or "/tools/" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        shp = str(ev.input_shapes)[:60]
        cnt[(ev.name, frame.strip()[:110], shp)] += 1
for (name, frame, shp), n in sorted(cnt.items(), key=lambda x: -x[1])[:40]:
    print("%4d  %-14s %-60s %s" % (n, name, shp, frame))
