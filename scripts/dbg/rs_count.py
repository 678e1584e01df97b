import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from bench import synthetic_batch
from rendernet_amd import _lib as L
B = 24
vox_np, pose_np = synthetic_batch(B)
lib = L.lib()
for name, vox in (("fixtures", torch.as_tensor(vox_np).float().cuda().contiguous()), ("dense", (torch.rand((B,64,64,64,1), device='cuda') < 0.2).float())):
    pose = torch.as_tensor(pose_np).float().cuda()
    out = torch.empty((B,128,128,128,1), device='cuda')
    nws = int(lib.rn_resample_workspace_bytes(B, 64, 1))
    ws = torch.zeros(nws, dtype=torch.uint8, device='cuda')
    print(vox.dtype, vox.is_contiguous(), pose.dtype, pose.is_contiguous(), out.dtype, flush=True)
    for _ in range(3):
        L.check(lib.rn_resample_fwd(L.ptr(vox), L.ptr(pose), L.ptr(out), B, 64, 128, 1, 0, 0, 128, 128, 1, ctypes.c_void_p(ws.data_ptr()), nws, L.stream_ptr()), "rs")
    torch.cuda.synchronize()
    base = (ws.data_ptr() + 127) // 128 * 128 - ws.data_ptr()
    off = B * 28 * 4 + B * 16 * 16 * 4 + B * 64 * 64 * 2 * 4
    off = base + (off + 127) // 128 * 128
    c = ws[off:off + B * 128].view(torch.int32).view(B, 32)
    cnt = int(c[:, 0].sum().item())
    print(name, 'tiles surviving tile-level test', int(c[0, 1]), 'hit samples', int(c[0, 2]), 'waves with hits', int(c[0, 3]))
    nz = (out != 0).float().mean().item()
    print(name, "non-empty tiles", cnt, "of", B * 16 * 16 * 16, "nonzero output fraction %.4f" % nz)
