#!/usr/bin/env python
"""Inverse-rendering step at the reference sizes (Reconstruct_RenderNet_Face.py:334-413, five hypotheses, 64^3 -> 128^3 -> 512^2): losses and
latent gradients in every multiply-stage mode against the exact-fp32 mode, and -- as the yardstick for what fp32 rounding alone does to these
heavily cancelling sums -- the exact-fp32 mode with F(4x4,3x3) instead of F(6x6,3x3) (two exact routes of the same convs).  Development tool;
the bars of tests/test_gpu_reconstruct.py::test_full_size_inverse_rendering_gradients_agree_across_multiply_stage_modes come from here."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rendernet_amd import ops  # noqa: E402
from rendernet_amd import reconstruct as RC  # noqa: E402


def run(mode, gain=None):
    old_mode, old_gain = ops.WINO_GEMM, ops.WINO63_MIN_GAIN
    ops.WINO_GEMM = mode
    if gain is not None:
        ops.WINO63_MIN_GAIN = gain
    try:
        rng = np.random.default_rng(1)
        rec = RC.Reconstructor(batch_size=5)
        rec.assign(vector=np.full((5, 200), 0.5, np.float32), param=RC.create_param_center(5, 270, 60, 90, 30),
                   texture=rng.standard_normal((5, 199)).astype(np.float32),
                   light=(np.linspace(230, 320, num=5) * math.pi / 180.0)[:, None])
        rec.etas.update(vector=0.0, param=0.0, texture=0.0, light=0.0)
        target = torch.from_numpy(rng.uniform(0, 1, (5, 512, 512, 3)).astype(np.float32)).cuda()
        loss = rec.step(target).cpu().numpy()
        grads = {k: v.grad.cpu().numpy().astype(np.float64) for k, v in rec.latents.items()}
        del rec
        torch.cuda.empty_cache()
        return loss, grads
    finally:
        ops.WINO_GEMM, ops.WINO63_MIN_GAIN = old_mode, old_gain


def main():
    l0, g0 = run("f32")
    print("losses (exact):", l0)
    for name, (l, g) in (("exact, F(4x4,3x3) only", run("f32", 2.0)), ("split (bf16x3)", run("split")), ("split16 (fp16x2)", run("split16")),
                         ("exact again (determinism)", run("f32"))):
        print("%-28s loss rel %.2e | " % (name, np.abs(l - l0).max() / np.abs(l0).max()) +
              "  ".join("%s %.2e (max|g| %.2e)" % (k, np.abs(g[k] - g0[k]).max() / np.abs(g0[k]).max(), np.abs(g0[k]).max()) for k in g0), flush=True)


if __name__ == "__main__":
    main()
