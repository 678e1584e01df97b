#!/usr/bin/env python
"""One-off parity record of the WHOLE benched batch: all 24 frames of bench.py's config-2 batch rendered by the HIP path in one
call vs the CPU oracle (frames go through the oracle in groups of 4; ~1 min of CPU).  Prints one JSON object with the per-frame
max |gpu - oracle| (tolerance 1e-3, BASELINE.json north_star).  usage (GPU box): python scripts/parity_full_batch.py > out.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import rendernet as ON  # noqa: E402
from oracle import resample as OR  # noqa: E402


def main():
    B = 24
    torch.set_num_threads(bench.pick_threads()[0])
    wl = bench.build_workload("render", "cuda:0")
    vox, _, poses = wl["inputs"](B)
    with torch.no_grad():
        got = wl["render"](torch.as_tensor(vox).cuda(), None, torch.as_tensor(poses).cuda()).cpu().numpy()
    errs = []
    for s in range(0, B, 4):
        want = ON.rendernet_forward(OR.net_input(vox[s:s + 4], poses[s:s + 4], 64, 128), wl["weights"])
        errs += [float(np.abs(got[s + i] - want[i]).max()) for i in range(want.shape[0])]
    print(json.dumps({"config": "BASELINE configs[1]: 24 frames, 64^3 -> 512x512, the batch of bench.py", "tol": 1e-3,
                      "max_abs_err_per_frame": errs, "max_abs_err": max(errs), "ok": max(errs) <= 1e-3,
                      "csrc_digest": bench.kernel_sources_digest(), "git": os.environ.get("GIT_REV", "unknown")}))


if __name__ == "__main__":
    main()
