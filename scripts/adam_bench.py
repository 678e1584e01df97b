import os, sys, torch, ctypes, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rendernet_amd import _lib as L
n = 237270428
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
lib = L.lib()
def run():
    L.check(lib.rn_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 1.0, L.stream_ptr()), "adam")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("RN_ADAM_WGS=%s  %.3f ms  %.2f TB/s" % (os.environ.get("RN_ADAM_WGS", "4096"), ms, 7 * 4 * n / ms * 1e-9))
