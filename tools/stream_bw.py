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

# the library's own streaming read (XOR hash over a key array, 16 B per lane)
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

ctx = capi.Context(0)
g = capi.geom(23, 14)
for size in (10_000_000, 100_000_000):
    km = synth_torch.phylogeny_sets(23, 1, size, 5, dev)[0]
    s = synth_torch.device_set(g, km)
    del km
    t = timeit(lambda: ctx.set_hash(s), n=10)
    print("ksh_set_hash %4d MB: %.1f us  %.2f TB/s (includes one stream sync + 8-byte read-back)" % (
        s.n_keys * 4 // 1000000, t * 1e6, s.n_keys * 4 / t / 1e12))
