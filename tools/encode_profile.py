"""One SPSS encode of a 1e8-k-mer set, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = capi.Context(0)
g = capi.geom(23, 14)
fam = synth_torch.phylogeny_sets(23, 2, size, 4, ctx.device)
a, b = (synth_torch.device_set(g, x) for x in fam)
del fam
inter, amb, bma = ctx.pair_algebra(a, b)
for s in (a, amb):          # one whole genome (few long unitigs) and a difference set (many short ones)
    for _ in range(2):
        sp = ctx.spss_encode(s, mode=0)
    torch.cuda.synchronize()
    print("n", s.n_keys, "strings", sp.n_strings, "unitigs", ctx.spss_encode_stats().get("unitigs"))
