"""Per-kernel times of ONE kind of set of the loop, for rocprofv3 --kernel-trace --stats: argv[1] =
genome | difference | intersection (of two 1e8 genomes of the phylogeny family)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "difference"
size = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
ctx = capi.Context(0)
g = capi.geom(23, 14)
fam = synth_torch.phylogeny_sets(23, 2, size, 4, ctx.device)
a, b = (synth_torch.device_set(g, x) for x in fam)
del fam
inter, amb, bma = ctx.pair_algebra(a, b)
s = {"genome": a, "difference": amb, "intersection": inter}[which]
for _ in range(3):
    sp = ctx.spss_encode(s, mode=0)
torch.cuda.synchronize()
print("n", s.n_keys, "strings", sp.n_strings, "unitigs", ctx.spss_encode_stats().get("unitigs"))
