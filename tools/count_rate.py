"""Rates of the FASTA -> counted k-mer set path and of the SPSS text kernels (numbers quoted in
DESIGN.md; not part of the test suite)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

ctx = capi.Context(0)
dev = ctx.device
k, n = 23, 14
g = capi.geom(k, n)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


for n_reads, genome_len in ((1_000_000, 2_000_000), (10_000_000, 30_000_000)):
    gen = torch.Generator(device="cpu").manual_seed(11)
    genome = torch.randint(0, 4, (genome_len,), generator=gen, dtype=torch.int64)
    starts = torch.randint(0, genome.numel() - 100, (n_reads,), generator=gen, dtype=torch.int64)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
    rec = torch.empty((n_reads, 103), dtype=torch.uint8)
    rec[:, 0], rec[:, 1], rec[:, 102] = ord(">"), ord("\n"), ord("\n")
    for j in range(100):
        rec[:, 2 + j] = letters[genome[starts + j]]
    text = rec.reshape(-1).to(dev)
    t_frag, frags = timed(lambda: ctx.fasta_fragments(g, text))
    t_cnt, (s, n_cut) = timed(lambda: ctx.kmer_count(frags, 2))
    kmers = n_reads * (100 - k + 1)
    print("FASTA %d MB, %d reads: fragments %.2f ms (%.1f GB/s of text), count+cutoff %.2f ms "
          "(%.1f G k-mer occurrences/s); %d k-mers kept, %d cut" % (
              text.numel() // 1000000, n_reads, t_frag * 1e3, text.numel() / t_frag / 1e9, t_cnt * 1e3,
              kmers / t_cnt / 1e9, s.n_keys, n_cut))
    del text, frags, s

km = synth_torch.phylogeny_sets(k, 1, 50_000_000, 4, dev)[0]
d = synth_torch.device_set(g, km)
sp = ctx.spss_encode(d, mode=0)
t_txt, text = timed(lambda: ctx.spss_to_text(sp))
t_back, back = timed(lambda: ctx.spss_from_text(g, text))
print("SPSS of %d k-mers: %d strings, %d bases; to text %.2f ms (%.1f GB/s of text), from text %.2f ms (%.1f GB/s)" % (
    d.n_keys, sp.n_strings, sp.n_bases, t_txt * 1e3, text.numel() / t_txt / 1e9, t_back * 1e3,
    text.numel() / t_back / 1e9))
