"""GPU parity tests for the step before the loop: FASTA text -> counted k-mers -> KmerSet
(KmerCounter::FromFASTA / FromReads / ToKmerSet, lib/core/kmer_counter.h), through the C ABI
(ksh_fasta_plan / ksh_fasta_write / ksh_spss_decode_plan / ksh_kmer_count_write) against the
oracle's restatement and the reference's known answers.  Bit-exact.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from kmersets import capi

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")


@pytest.fixture(scope="module")
def ctx(gpu):
    c = capi.Context(0)
    yield c
    c.close()


def to_device(ctx, raw):
    import torch

    if not raw:
        return torch.zeros(0, dtype=torch.uint8, device=ctx.device)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(ctx.device)


def fasta_of(reads):
    return "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)).encode()


def check_counts(ctx, k, n, kb, text, canonical, cutoffs):
    g = capi.geom(k, n)
    frags = ctx.fasta_fragments(g, to_device(ctx, text))
    for cutoff in cutoffs:
        oc = ol.Counter(k, n, kb)
        assert oc.from_fasta(text, canonical=canonical) == 0
        want_set, want_cut = oc.to_set(cutoff)
        got, n_cut = ctx.kmer_count(frags, cutoff, canonical=canonical)
        assert got.n_keys == want_set.size()
        assert n_cut == want_cut
        assert np.array_equal(got.kmers(), want_set.kmers())
        if cutoff <= 1:
            assert got.n_keys == oc.size() and n_cut == 0
    return frags


def test_reference_known_answers(ctx):
    gold = json.load(open(GOLDEN))["kmer_counter"]
    k, n = 5, 3
    g = capi.geom(k, n)
    fr = gold["from_reads"]
    frags = ctx.fasta_fragments(g, to_device(ctx, fasta_of(fr["reads"])))
    assert frags.to_strings() == fr["reads"]
    for cutoff in (1, 2, 3):
        got, n_cut = ctx.kmer_count(frags, cutoff, canonical=fr["canonical"])
        want = sorted(ol.kmer(s) for s, c in fr["want"].items() if c >= cutoff)
        assert [int(x) for x in got.kmers()] == want
        assert n_cut == sum(1 for c in fr["want"].values() if c < cutoff)
    # ToKmerSet's vector: counts 3, 1, 2, 4 as repeated reads, cutoff 3
    tk = gold["to_kmer_set"]
    reads = [s for s, v in tk["adds"] for _ in range(v)]
    frags = ctx.fasta_fragments(g, to_device(ctx, fasta_of(reads)))
    got, n_cut = ctx.kmer_count(frags, tk["cutoff"], canonical=False)
    assert n_cut == tk["want_cutoff_count"]
    assert sorted(ol.kmer_str(int(x), k) for x in got.kmers()) == sorted(tk["want_set"])


@pytest.mark.parametrize("geom", [(5, 3, 1), (15, 14, 2), (23, 14, 4), (31, 14, 8)])
@pytest.mark.parametrize("canonical", [True, False])
def test_random_fasta_vs_oracle(ctx, geom, canonical):
    k, n, kb = geom
    rng = np.random.default_rng(100 * k + int(canonical))
    genome = rng.integers(0, 4, size=6000 if k > 5 else 300)
    reads = []
    for _ in range(400 if k > 5 else 60):                       # overlapping reads: counts > 1
        a = int(rng.integers(0, genome.size - 10))
        ln = int(rng.integers(1, 260))
        seq = np.array(list("ACGT"))[genome[a:a + ln]]
        seq = seq.copy()
        for _ in range(int(rng.integers(0, 3))):                # sprinkle N's, sometimes in runs
            p = int(rng.integers(0, seq.size))
            seq[p:p + int(rng.integers(1, 4))] = "N"
        reads.append("".join(seq))
    reads += ["", "N", "NNNN", "ACGT"[:k - 1] if k > 4 else "", "A" * (k - 1) + "N" + "C" * k]
    text = fasta_of(reads)
    frags = check_counts(ctx, k, n, kb, text, canonical, (0, 1, 2, 3, 5))
    # the fragments are the maximal ACGT runs of length >= K, in file order
    want = [f for r in reads for f in r.split("N") if len(f) >= k]
    assert frags.to_strings() == want
    check_counts(ctx, k, n, kb, text[:-1], canonical, (1, 2))   # no final newline


def test_edge_shapes(ctx):
    k, n, kb = 9, 10, 4
    g = capi.geom(k, n)
    assert ctx.fasta_fragments(g, to_device(ctx, b"")).n_strings == 0
    for text in (b">only header\n\n", b">a\nACGTACG\n>b\nNNNNNNNNNNNN\n"):    # nothing long enough
        fr = ctx.fasta_fragments(g, to_device(ctx, text))
        assert fr.n_strings == 0 and fr.n_bases == 0
        got, n_cut = ctx.kmer_count(fr, 1)
        assert got.n_keys == 0 and n_cut == 0 and int(got.offsets[-1]) == 0
    # one line much longer than a workgroup's span, with chunk-aligned and unaligned N's
    rng = np.random.default_rng(3)
    line = np.array(list("ACGT"))[rng.integers(0, 4, size=200000)]
    for p in (63, 64, 65, 16383, 16384, 16385, 70000, 199999):
        line[p] = "N"
    text = b">chr with spaces and > signs\n" + "".join(line).encode() + b"\n"
    check_counts(ctx, k, n, kb, text, True, (1, 2))
    # saturation: the reference's counts stop at 255; a cutoff of 255 keeps a k-mer seen 300
    # times and cuts one seen 254 times
    reads = ["ACGTACGTT"] * 300 + ["GGGTTTAAC"] * 254 + ["CCCCCCCCA"]
    frags = ctx.fasta_fragments(g, to_device(ctx, fasta_of(reads)))
    got, n_cut = ctx.kmer_count(frags, 255, canonical=False)
    assert [int(x) for x in got.kmers()] == [ol.kmer("ACGTACGTT")] and n_cut == 2
    oc = ol.Counter(k, n, kb)
    assert oc.from_fasta(fasta_of(reads), canonical=False) == 0
    assert oc.get(ol.kmer("ACGTACGTT")) == 255 and oc.to_set(255)[1] == 2


def test_invalid_fasta(ctx):
    g = capi.geom(5, 3)
    cases = [(b">r1\nACGTA\n>r2\n", "even number of lines"), (b"r1\nACGTA\n", "invalid FASTA"),
             (b"\nACGTA\n", "invalid FASTA"), (b">r1\nACGTa\n", "invalid FASTA"),
             (b">r1\nACGTA\r\n", "invalid FASTA"), (b">r1\nACGT>\n", "invalid FASTA")]
    for raw, what in cases:
        with pytest.raises(capi.KshError) as e:
            ctx.fasta_fragments(g, to_device(ctx, raw))
        assert e.value.code == 9 and what in str(e.value)
        k, n, kb = 5, 3, 1
        assert ol.Counter(k, n, kb).from_fasta(raw) == (1 if "even" in what else 2)


def test_counting_at_scale(ctx):
    """10^6 reads of 100 bases over a 2 x 10^6-base genome (k = 23): sizes against torch on the
    same k-mers, and ToKmerSet(1) == the set of all k-mers."""
    import torch

    k, n = 23, 14
    g = capi.geom(k, n)
    dev = ctx.device
    gen = torch.Generator(device="cpu").manual_seed(11)
    genome = torch.randint(0, 4, (2_000_000,), generator=gen, dtype=torch.int64)
    starts = torch.randint(0, genome.numel() - 100, (1_000_000,), generator=gen, dtype=torch.int64)
    idx = (starts[:, None] + torch.arange(100)[None, :]).reshape(-1)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8)[genome[idx]].reshape(-1, 100)
    rec = torch.empty((letters.shape[0], 103), dtype=torch.uint8)          # ">\n" + read + "\n"
    rec[:, 0], rec[:, 1], rec[:, 102] = ord(">"), ord("\n"), ord("\n")
    rec[:, 2:102] = letters
    text = rec.reshape(-1).to(dev)
    frags = ctx.fasta_fragments(g, text)
    assert frags.n_strings == 1_000_000 and frags.n_bases == 100_000_000
    # reference counts with torch: canonical k-mers of every window, sorted, run lengths
    codes = genome[idx].reshape(-1, 100).to(dev)
    fw = torch.zeros((codes.shape[0], 100 - k + 1), dtype=torch.int64, device=dev)
    rc = torch.zeros_like(fw)
    for j in range(k):
        fw = (fw << 2) | codes[:, j:j + 100 - k + 1]
        rc = rc | ((3 - codes[:, j:j + 100 - k + 1]) << (2 * j))
    canon = torch.minimum(fw, rc).reshape(-1)
    uniq, counts = torch.unique(canon, return_counts=True)
    for cutoff in (1, 2, 4):
        got, n_cut = ctx.kmer_count(frags, cutoff)
        assert got.n_keys == int((counts >= cutoff).sum()) and n_cut == int((counts < cutoff).sum())
        assert np.array_equal(got.kmers(), uniq[counts >= cutoff].cpu().numpy().astype(np.uint64))
