"""Time of one SPSS encode by set size (the loop's dominant stage); quoted in DESIGN.md."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

ctx = capi.Context(0)
for k, size in ((23, 10_000_000), (23, 100_000_000), (31, 100_000_000)):
    g = capi.geom(k, 14)
    d = synth_torch.device_set(g, synth_torch.phylogeny_sets(k, 1, size, 4, ctx.device)[0])
    ctx.spss_encode(d, mode=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sp = ctx.spss_encode(d, mode=0)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print("k=%d n=%d: encode %.2f ms (%.2f G k-mers/s), %d strings" % (k, d.n_keys, t * 1e3, d.n_keys / t / 1e9, sp.n_strings))
    del d, sp
