import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rendernet_amd import ops
def conv_f64(x, w):
    xn = F.pad(torch.as_tensor(x).double().permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1))
    return F.conv3d(xn, torch.as_tensor(w).double().permute(4, 3, 0, 1, 2)).permute(0, 2, 3, 4, 1).contiguous()
for (B, H, W, D) in ((1, 2, 32, 8), (1, 2, 32, 8), (1, 2, 32, 5)):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 32, 32)) * 0.05).astype(np.float32)
    want = conv_f64(x, w)
    ops.CONV3D_SPLIT = True
    xd = torch.as_tensor(x).cuda()
    pw = ops.pack_conv(torch.as_tensor(w).cuda())
    ys = [ops.conv3d(xd, pw, None).cpu().double() for _ in range(3)]
    for y in ys:
        e = (y - want).abs()
        bad = (e > 1e-4).nonzero()
        print(B, H, W, D, "max err", float(e.max()), "n bad", len(bad))
        for i in bad[:6].tolist():
            print("   ", i, "got", float(y[tuple(i)]), "want", float(want[tuple(i)]))
    # which taps are missing?  contributions per depth tap at the first bad position
    if len(bad):
        i = bad[0].tolist()
        for dz in range(3):
            wz = np.zeros_like(w); wz[:, :, dz] = w[:, :, dz]
            print("    tap", dz, float(conv_f64(x, wz)[tuple(i)]))
