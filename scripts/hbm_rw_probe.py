#!/usr/bin/env python
"""What one MI355X sustains for a pure read stream, a pure write stream and a copy (torch elementwise kernels on 1 GiB tensors): the yardstick for
the write-heavy launches of the path (input transform: 74 % writes; resampler: 89 % writes).  Development tool."""
import torch


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    n = 1 << 28                                   # 1 GiB of float32
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    gb = n * 4 / 1e9
    t = timeit(lambda: y.fill_(1.5))
    print("write  (fill_, %.2f GB):            %.3f ms  %.2f TB/s" % (gb, t, gb / t))
    t = timeit(lambda: y.zero_())
    print("write  (zero_, %.2f GB):            %.3f ms  %.2f TB/s" % (gb, t, gb / t))
    t = timeit(lambda: x.sum())
    print("read   (sum, %.2f GB):              %.3f ms  %.2f TB/s" % (gb, t, gb / t))
    t = timeit(lambda: y.copy_(x))
    print("copy   (copy_, %.2f GB in + out):   %.3f ms  %.2f TB/s" % (2 * gb, t, 2 * gb / t))
    t = timeit(lambda: torch.add(x, 1.0, out=y))
    print("r + w  (add, %.2f GB in + out):     %.3f ms  %.2f TB/s" % (2 * gb, t, 2 * gb / t))


if __name__ == "__main__":
    main()
