"""SPSS encode of a set with bubbles -- the union of two genomes that differ by substitutions, whose
path cover stitches 10^5..10^6 unitigs into few strings -- next to a whole-genome set of the same
size (few long unitigs) and a fragmented difference set: wall times per encode, unitigs, strings.
Run under rocprofv3 --kernel-trace --stats for the per-kernel times."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = capi.Context(0)
g = capi.geom(23, 14)
fam = synth_torch.phylogeny_sets(23, 2, size, 4, ctx.device)
union = torch.unique(torch.cat(fam))           # sorted
sets = {"genome": synth_torch.device_set(g, fam[0]), "union_of_two_genomes": synth_torch.device_set(g, union)}
# a genome with tips: every `every`-th k-mer gets a second successor ending in T (where the genome goes
# on with A, C or G): the greedy sweep takes the genome's edge first (lower base), the tips stay
# single, and the path cover stitches the unitigs between the tips into ONE string per stretch
sets["genome_with_tips"] = synth_torch.device_set(g, synth_torch.genome_with_tips(23, size, 4, ctx.device, every=400))
a, b = (synth_torch.device_set(g, x) for x in fam)
del fam, union
_inter, amb, _bma = ctx.pair_algebra(a, b)
sets["difference"] = amb
out = {}
for name, s in sets.items():
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp = ctx.spss_encode(s, mode=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = ctx.spss_encode_stats()
    out[name] = {"kmers": s.n_keys, "encode_ms": best * 1e3, "ns_per_kmer": best * 1e9 / max(s.n_keys, 1),
                 "unitigs": st["unitigs"], "strings": st["strings"], "matching_rounds": st["rounds"],
                 "unitigs_per_string": st["unitigs"] / max(st["strings"], 1)}
print(json.dumps(out))
