"""Achievable HBM streaming rates on this GPU (reference points for the roofline fractions)."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (80, 400, 2000):
    n = mb * 1000 * 1000 // 4
    x = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
    y = torch.empty_like(x)
    t = timeit(lambda: x.sum())
    print("read  %5d MB: %.1f us  %.2f TB/s" % (mb, t * 1e6, mb * 1e6 / t / 1e12))
    t = timeit(lambda: y.copy_(x))
    print("copy  %5d MB: %.1f us  %.2f TB/s (read + write)" % (mb, t * 1e6, 2 * mb * 1e6 / t / 1e12))
    t = timeit(lambda: y.fill_(1))
    print("write %5d MB: %.1f us  %.2f TB/s" % (mb, t * 1e6, mb * 1e6 / t / 1e12))
