"""GPU parity tests for the SPSS container: decode (KmerSetCompact::ToKmerSet), encode
(KmerSetCompact::FromKmerSet = GetSPSSCanonical fast, and GetUnitigsCanonical) and
Size/Weight, through the C ABI, against the oracle on the same seeded inputs.
Bit-exact: the strings must equal the oracle's, in order."""
import numpy as np
import pytest

import oracle_lib as ol
from kmersets import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    c = capi.Context(0)
    yield c
    c.close()


def dev_set(ctx, k, n, kmers):
    return capi.DeviceSet.from_kmers(capi.geom(k, n), np.asarray(kmers, dtype=np.uint64), ctx.device)


def kmers_of(seq, k):
    return synth.canonical_set_of_bases(synth.bases_of_string(seq), k)


# ------------------------------------------------------------------------------ decode
@pytest.mark.parametrize("geom", [(5, 3, 1), (9, 10, 1), (15, 14, 2), (19, 10, 4), (23, 14, 4), (31, 14, 8)])
def test_decode_oracle_spss(ctx, geom):
    """ToKmerSet(FromKmerSet(S)) == S  (test/kmer_set_compact.cc:71-90), decode side."""
    k, n, kb = geom
    kmers = synth.random_read_kmers(k, 150 if k == 5 else 6000, seed=k + 1, canonical=True)
    oset = ol.Set.from_kmers(k, n, kb, kmers)
    strings = oset.spss()
    sp = capi.DeviceSpss.from_strings(capi.geom(k, n), strings, ctx.device)
    assert ctx.spss_size(sp) == oset.size() == oset.compact().size()
    assert sp.weight() == oset.compact().weight()
    got = ctx.spss_decode(sp)
    assert got.n_keys == oset.size()
    assert np.array_equal(got.kmers(), oset.kmers())
    off, keys = got.to_numpy()
    w_off, w_keys = synth.to_bucketed(oset.kmers(), k, n, got.g.key_bytes)
    assert np.array_equal(off, w_off) and np.array_equal(keys, w_keys)


def test_decode_arbitrary_lines(ctx):
    """GetKmerSetFromSPSS on lines that are not an SPSS: repeated k-mers collapse, strings
    of exactly K bases, string boundaries at every offset of the 32-base words."""
    k, n, kb = 9, 10, 1
    g = synth.random_genome(8000, 77)
    text = synth.string_of_bases(g)
    lines = []
    at = 0
    for i in range(120):
        ln = k + (i * 7) % 40
        lines.append(text[at:at + ln])
        at += ln - (i % 5)            # overlaps -> repeated k-mers across lines
    lines += [text[:k], text[:k], text[5:5 + k], "ACGTACGTACGTACGTACGT", "A" * 40]
    want = ol.Set.from_spss(lines, k, n, kb)
    sp = capi.DeviceSpss.from_strings(capi.geom(k, n), lines, ctx.device)
    got = ctx.spss_decode(sp)
    assert got.n_keys == want.size()
    assert np.array_equal(got.kmers(), want.kmers())
    assert ctx.spss_size(sp) == sum(len(x) - k + 1 for x in lines)
    # non-canonical decode keeps the forward k-mers
    want_fw = ol.Set.from_spss(lines, k, n, kb, canonical=False)
    got_fw = ctx.spss_decode(sp, canonical=False)
    assert np.array_equal(got_fw.kmers(), want_fw.kmers())


def test_decode_empty_and_tiny(ctx):
    k, n = 23, 14
    g = capi.geom(k, n)
    sp = capi.DeviceSpss.from_strings(g, [], ctx.device)
    got = ctx.spss_decode(sp)
    assert got.n_keys == 0 and int(got.offsets[-1]) == 0
    one = "ACGTTGCATGCATGACTGACTGA"
    sp = capi.DeviceSpss.from_strings(g, [one], ctx.device)
    got = ctx.spss_decode(sp)
    assert got.n_keys == 1
    assert int(got.kmers()[0]) == int(ol.lib().ko_canonical(ol.kmer(one), k))


def test_decode_two_level_scatter(gpu):
    """The decode's two-level run-wise scatter (k_decode_l1 / k_decode_l2: what inputs of 2^20 k-mers and
    more take) on small inputs (KSH_DECODE_L2_MIN=1024), against the oracle: SPSS of phylogeny sets at the
    three key widths, arbitrary lines with repeated k-mers and string ends at every offset of a word,
    a k-mer count with a cutoff; and the direct scatter (KSH_DECODE_SCATTER=direct) on the same."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from kmersets import capi, synth\n"
        "ctx = capi.Context(0)\n"
        "for (k, n, kb) in ((15, 14, 2), (23, 14, 4), (31, 14, 8), (23, 10, 8)):\n"
        "    for kmers in synth.phylogeny_sets(k, 2, 50000, seed=k) + [synth.random_read_kmers(k, 3000, seed=k, canonical=True)]:\n"
        "        o = ol.Set.from_kmers(k, n, kb, kmers)\n"
        "        sp = capi.DeviceSpss.from_strings(capi.geom(k, n, kb), o.spss(), ctx.device)\n"
        "        got = ctx.spss_decode(sp)\n"
        "        assert got.n_keys == o.size() and np.array_equal(got.kmers(), o.kmers()), (k, n)\n"
        "k, n, kb = 23, 14, 4\n"
        "text = synth.string_of_bases(synth.random_genome(60000, 5))\n"
        "lines, at = [], 0\n"
        "for i in range(900):\n"
        "    ln = k + (i * 7) %% 90\n"
        "    lines.append(text[at:at + ln]); at += ln - (i %% 5)\n"
        "want = ol.Set.from_spss(lines, k, n, kb)\n"
        "got = ctx.spss_decode(capi.DeviceSpss.from_strings(capi.geom(k, n), lines, ctx.device))\n"
        "assert got.n_keys == want.size() and np.array_equal(got.kmers(), want.kmers())\n"
        "c = ol.Counter(k, n, kb); c.from_reads(lines + lines[:300]); wset, w_cut = c.to_set(2)\n"
        "gset, n_cut = ctx.kmer_count(capi.DeviceSpss.from_strings(capi.geom(k, n), lines + lines[:300], ctx.device), 2)\n"
        "assert np.array_equal(gset.kmers(), wset.kmers()) and n_cut == w_cut\n"
        "print('decode routes ok')\n"
    ) % (os.path.join(here, "..", "kmer-sets-compression_amd"), here)
    for env_add in ({"KSH_DECODE_L2_MIN": "1024"}, {"KSH_DECODE_SCATTER": "direct"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_add), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "decode routes ok" in r.stdout, str(env_add) + r.stdout[-2000:] + r.stderr[-4000:]


# ------------------------------------------------------------------------------ encode
def check_encode(ctx, k, n, kb, kmers):
    oset = ol.Set.from_kmers(k, n, kb, kmers)
    d = dev_set(ctx, k, n, oset.kmers())
    unitigs = ctx.spss_encode(d, mode=1)
    assert unitigs.to_strings() == oset.unitigs()
    spss = ctx.spss_encode(d, mode=0)
    stats = ctx.spss_encode_stats()
    want = oset.spss()
    assert spss.to_strings() == want
    assert spss.n_strings == len(want) and spss.n_bases == sum(len(x) for x in want)
    assert spss.weight() == oset.compact().weight()
    # and back: ToKmerSet(FromKmerSet(S)) == S on device
    back = ctx.spss_decode(spss)
    assert back.n_keys == d.n_keys and ctx.set_diff(back, d) == 0
    return stats


def test_encode_known_answers(ctx):
    """SURVEY.md 3.2 known answers of the reference, and the small special shapes."""
    k, n, kb = 5, 3, 1
    d = dev_set(ctx, k, n, kmers_of("AACCGTTAGCAT", k))
    assert ctx.spss_encode(d, mode=0).to_strings() == ["ATGCTAACGGTT"]
    d = dev_set(ctx, k, n, kmers_of("ACGTACGTACG", k))
    assert ctx.spss_encode(d, mode=0).to_strings() == ["GTACGT"]
    for seq in ["AAAAAAAAA", "AACCGGTT", "ACGTTGCAACGT", "ATATATATAT", "GATTACAGATTACAGATTACA",
                "CCCCCCGGGGGG"]:
        check_encode(ctx, k, n, kb, kmers_of(seq, k))
    e = dev_set(ctx, k, n, np.zeros(0, dtype=np.uint64))
    sp = ctx.spss_encode(e, mode=0)
    assert sp.n_strings == 0 and sp.n_bases == 0 and sp.to_strings() == []


@pytest.mark.parametrize("seed", range(8))
def test_encode_random_reads(ctx, seed):
    """test/spss.cc:43-66,127-153 shapes (K=9, N=10), exact strings instead of invariants."""
    k = [5, 7, 9, 9, 11, 15, 9, 9][seed]
    size = [50, 300, 2000, 20000, 3000, 3000, 65536, 1][seed]
    km = synth.random_read_kmers(k, min(size, 4 ** k // 3), seed=seed, canonical=True)
    check_encode(ctx, k, min(10, 2 * k - 4), 4, km)


def test_encode_loops_and_loop_cuts(ctx):
    """Non-branching loops (spss.h:585-610) and loops of the path cover (spss.h:1578-1644)."""
    n_loops = 0
    for seed in range(60):
        k = [5, 7, 9, 11][seed % 4]
        km = synth.circular_with_tails(k, 20 + (seed * 7) % 150, seed % 5, 1 + seed % 4, seed)
        check_encode(ctx, k, min(10, 2 * k - 4), 4, km)
        n_loops += 1
    assert n_loops == 60


@pytest.mark.parametrize("geom", [(15, 14, 2), (19, 10, 4), (23, 14, 4), (31, 14, 8)])
def test_encode_families(ctx, geom):
    """The CLI geometries on a correlated family and on its algebra results (what one
    KmerSetSet iteration encodes: A&B, A\\B, B\\A)."""
    k, n, kb = geom
    sets = synth.phylogeny_sets(k, 3, 30000, seed=k)
    stats = check_encode(ctx, k, n, kb, sets[0])
    assert stats["unitigs"] >= 1
    check_encode(ctx, k, n, kb, np.intersect1d(sets[0], sets[1]))
    check_encode(ctx, k, n, kb, np.setdiff1d(sets[0], sets[1]))
    check_encode(ctx, k, n, kb, np.setdiff1d(sets[2], sets[0]))


def test_encode_many_unitigs(ctx):
    """150 000 unrelated k-mers: about as many unitigs, nearly all of one kind, so the packed
    two-counter prefix sums over the unitigs pass 2^48 on arrays of 10^5 elements (the
    single-launch scan has to carry full 64-bit sums), and a mixed set on the same path."""
    k, n, kb = 15, 14, 2
    lone, other = synth.uniform_pair(k, 150000, 0.5, seed=77)
    stats = check_encode(ctx, k, n, kb, lone)
    assert stats["unitigs"] > 100000
    reads = synth.random_read_kmers(k, 200000, seed=78, canonical=True)
    check_encode(ctx, k, n, kb, np.union1d(reads, other[:80000]))


def test_encode_roundtrip_large(ctx):
    """Size-independent properties at 2 x 10^6 k-mers (k = 23): every k-mer exactly once,
    decode(encode(S)) == S, weight = |S| + (K-1) * strings."""
    k, n = 23, 14
    sets = synth.phylogeny_sets(k, 2, 2_000_000, seed=5)
    a, b = dev_set(ctx, k, n, sets[0]), dev_set(ctx, k, n, sets[1])
    inter, amb, _ = ctx.pair_algebra(a, b)
    for s in (a, inter, amb):
        sp = ctx.spss_encode(s, mode=0)
        assert ctx.spss_size(sp) == s.n_keys               # sum(len - K + 1) == |S|: no k-mer twice
        assert sp.n_bases == s.n_keys + (k - 1) * sp.n_strings
        back = ctx.spss_decode(sp)
        assert back.n_keys == s.n_keys and ctx.set_diff(back, s) == 0
        un = ctx.spss_encode(s, mode=1)
        assert un.n_strings >= sp.n_strings
        assert ctx.set_diff(ctx.spss_decode(un), s) == 0


def test_streamvbyte_0124(ctx):
    """kmer_set_compact.h:257-265,272: device pack/unpack of the lengths vs the oracle's codec,
    byte for byte, and the compact's in-memory lengths of an encoded set."""
    L = ol.lib()
    rng = np.random.default_rng(5)
    for n in (0, 1, 3, 4, 5, 1000, 100003):
        v = rng.choice(np.array([0, 1, 7, 255, 256, 65535, 65536, 2**32 - 1], dtype=np.uint32), size=n)
        v = np.ascontiguousarray(v, dtype=np.uint32)
        want = np.zeros(max(1, L.ko_svb_max_compressed_bytes(n)), dtype=np.uint8)
        size = L.ko_svb_encode_0124(v if n else np.zeros(1, np.uint32), n, want)
        got = ctx.svb_encode(v)
        assert got.size == size and np.array_equal(got, want[:size])
        back, used = ctx.svb_decode(got, n)
        assert used == size and np.array_equal(back, v)
    k, n_bits, kb = 23, 14, 4
    kmers = synth.phylogeny_sets(k, 1, 20000, seed=8)[0]
    oset = ol.Set.from_kmers(k, n_bits, kb, np.setdiff1d(kmers, kmers[::7]))
    sp = ctx.spss_encode(dev_set(ctx, k, n_bits, oset.kmers()), mode=0)
    lens = sp.lens[: sp.n_strings].cpu().numpy().view(np.uint32)
    assert np.array_equal(ctx.svb_encode(lens), oset.compact().lengths_compressed())


@pytest.mark.parametrize("geom", [(13, 2, 4), (19, 2, 8), (13, 1, 4)])
def test_decode_oversize_buckets(ctx, geom):
    """Buckets larger than the LDS sort capacity (what 5 x 10^8 keys per set produce at N = 14)
    go through the partition path; forced here with very few buckets.  Repeated k-mers across
    lines must still collapse."""
    k, n, kb = geom
    kmers = synth.random_read_kmers(k, 60000, seed=31 + k, canonical=True)
    oset = ol.Set.from_kmers(k, n, kb, kmers)
    strings = oset.spss()
    strings = strings + strings[:50]          # duplicates
    sp = capi.DeviceSpss.from_strings(capi.geom(k, n), strings, ctx.device)
    got = ctx.spss_decode(sp)
    assert got.n_keys == oset.size()
    assert np.array_equal(got.kmers(), oset.kmers())
    # and the encoder on such a set (probes through big buckets)
    d = dev_set(ctx, k, n, oset.kmers())
    assert ctx.spss_encode(d, mode=0).to_strings() == oset.spss()


# ------------------------------------------------------------------------------ text form
def _text_of(strings):
    return "".join(x + "\n" for x in strings).encode()


def _roundtrip_text(ctx, g, strings):
    import torch

    sp = capi.DeviceSpss.from_strings(g, strings, ctx.device)
    text = ctx.spss_to_text(sp)
    want = _text_of(strings)
    assert bytes(text.cpu().numpy().tobytes()) == want
    for raw in (want, want[:-1] if want else want):          # with and without the final newline
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(ctx.device) if raw else \
            torch.zeros(0, dtype=torch.uint8, device=ctx.device)
        back = ctx.spss_from_text(g, t)
        assert back.n_strings == len(strings) and back.n_bases == sum(len(x) for x in strings)
        assert back.to_strings() == strings
        n_words = (back.n_bases + 31) // 32
        assert bool(torch.equal(back.words[:n_words], sp.words[:n_words]))
        assert bool(torch.equal(back.lens[: back.n_strings], sp.lens[: sp.n_strings]))


def test_text_form_small(ctx):
    """KmerSetCompact::Dump / Load text (kmer_set_compact.h:62-87): exact bytes both ways."""
    g = capi.geom(5, 3)
    _roundtrip_text(ctx, g, [])
    _roundtrip_text(ctx, g, ["ACGTA"])
    _roundtrip_text(ctx, g, ["ATGCTAACGGTT"])                                  # reference known answer
    _roundtrip_text(ctx, g, ["ACGTA", "CCCCCC", "GATTACAGATTACAGATTACA", "TTTTT"])
    rng = np.random.default_rng(5)
    for n in (31, 32, 33, 63, 64, 65, 500):                                   # word / chunk edges
        strings = ["".join("ACGT"[c] for c in rng.integers(0, 4, size=int(m)))
                   for m in rng.integers(5, 40, size=n)]
        _roundtrip_text(ctx, g, strings)
    # one string longer than a workgroup's span, then many strings of exactly K bases
    long_one = "".join("ACGT"[c] for c in rng.integers(0, 4, size=40000))
    _roundtrip_text(ctx, g, ["ACGTA", long_one, "GGGGG"])
    _roundtrip_text(ctx, g, ["".join("ACGT"[c] for c in rng.integers(0, 4, size=5)) for _ in range(9000)])


def test_text_form_of_an_encoded_set(ctx):
    """Dump text of a device-encoded SPSS == the oracle's strings joined by newlines, and the
    text loads back to the same set."""
    import torch

    k, n, kb = 23, 14, 4
    kmers = synth.phylogeny_sets(k, 1, 300000, seed=12)[0]
    oset = ol.Set.from_kmers(k, n, kb, kmers)
    d = dev_set(ctx, k, n, kmers)
    sp = ctx.spss_encode(d, mode=0)
    text = ctx.spss_to_text(sp)
    assert bytes(text.cpu().numpy().tobytes()) == _text_of(oset.spss())
    back = ctx.spss_from_text(sp.g, text)
    assert back.n_strings == sp.n_strings and back.n_bases == sp.n_bases
    assert ctx.set_diff(ctx.spss_decode(back), d) == 0


def test_text_form_rejects_bad_input(ctx):
    import torch

    g = capi.geom(5, 3)
    for raw in (b"ACGTN\n", b"ACGTA\r\n", b"ACG\n", b"ACGTA\n\nACGTA\n", b"acgta\n"):
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(ctx.device)
        with pytest.raises(capi.KshError) as e:
            ctx.spss_from_text(g, t)
        assert e.value.code == 3


def test_parallel_disjoint_set_vs_oracle(ctx):
    """ksh_dsu_components (the device ParallelDisjointSet) gives the partition of the oracle's serial
    union-find (test/parallel_disjoint_set.cc:14-118: same partition, any representatives): random
    graphs from very sparse to one giant component, a long chain, a ring, self pairs, no pairs."""
    rng = np.random.default_rng(5)
    cases = [(1000, rng.integers(0, 1000, 300), rng.integers(0, 1000, 300)),
             (50000, rng.integers(0, 50000, 20000), rng.integers(0, 50000, 20000)),
             (50000, rng.integers(0, 50000, 200000), rng.integers(0, 50000, 200000)),
             (200000, np.arange(199999), np.arange(1, 200000)),                        # one chain
             (4096, np.arange(4096), (np.arange(4096) + 1) % 4096),                    # a ring
             (100, np.arange(100), np.arange(100)),                                    # self pairs
             (77, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))]
    for n, xs, ys in cases:
        got = ctx.dsu_components(n, xs, ys)
        d = ol.lib().ko_dsu_new(n)
        for a, b in zip(xs.tolist(), ys.tolist()):
            ol.lib().ko_dsu_unite(d, a, b)
        want = np.array([ol.lib().ko_dsu_find(d, i) for i in range(n)])
        ol.lib().ko_dsu_free(d)
        assert np.array_equal(got[got], got)                         # representatives are roots
        # same partition: the map got-root -> want-root is a bijection on components
        fwd, bwd = {}, {}
        for g_, w_ in zip(got.tolist(), want.tolist()):
            assert fwd.setdefault(g_, w_) == w_ and bwd.setdefault(w_, g_) == g_


def test_encode_branching_sets(ctx):
    """Sets whose unitig graph branches at every few hundred k-mers -- a genome with one-k-mer tips, and
    the union of two diverged genomes (a bubble per substitution) -- so that the path cover has
    thousands of multi-unitig strings, competing edges at every junction and both orientations of
    every walk: the strings are the oracle's, in order (walks ranked by pointer jumping, loops found
    by the parallel union-find)."""
    k, n, kb = 23, 14, 4
    tips = synth.genome_with_tips(k, 150000, seed=7, every=150)
    a, b = synth.phylogeny_sets(k, 2, 60000, seed=77, rate=0.004)
    for name, kmers in (("tips", tips), ("bubbles", np.union1d(a, b))):
        d = capi.DeviceSet.from_kmers(capi.geom(k, n), kmers, ctx.device)
        for mode in (0, 2):
            sp = ctx.spss_encode(d, mode=mode)
            st = ctx.spss_encode_stats()
            o = ol.Set.from_kmers(k, n, kb, kmers)
            assert sp.to_strings() == (o.spss() if mode == 0 else o.spss_slow()), (name, mode)
            assert st["unitigs"] > 700 and st["strings"] < st["unitigs"]


def test_encode_skewed_rc_groups(ctx):
    """k_adj_rc1 takes the groups whose reverse-complement records fit its LDS -- sized from the AVERAGE group -- and
    k_adj_rc the others, in the same encode: a genome's k-mers (a dozen records per group) plus 6000 k-mers that
    carry one fixed 7-base block where the group id is read (bases k-8 .. k-2 of x, i.e. 1 .. 7 of rc(x)): those
    that stay canonical crowd one group far beyond the window.  Strings == the oracle's, both routes reported.
    (With KSH_RC1=batched that group is k_adj_rc1's too, a dozen batches deep: test_encode_alternative_paths has the
    same set.)"""
    k, n, kb = 23, 14, 4
    rng = np.random.default_rng(23)
    base = synth.phylogeny_sets(k, 1, 200000, seed=19)[0]
    x = rng.integers(0, 4 ** k, size=6000, dtype=np.uint64)
    block = np.uint64(0b10_01_11_00_01_10_11)  # G C T A C G T
    x = (x & ~np.uint64(0xFFFC)) | (block << np.uint64(2))
    kmers = np.unique(np.concatenate([base, synth.canonical(x, k)]))
    d = capi.DeviceSet.from_kmers(capi.geom(k, n), kmers, ctx.device)
    sp = ctx.spss_encode(d, mode=0)
    routes = ctx.spss_encode_routes()
    assert {"probe_staged", "rc1_streamed", "rc_marks_groups", "fwd_targets"} <= routes, sorted(routes)
    want = ol.Set.from_kmers(k, n, kb, kmers).spss()
    assert sp.to_strings() == want
    back = ctx.spss_decode(sp)
    assert back.n_keys == d.n_keys and ctx.set_diff(back, d) == 0
    # a set without the crowd takes k_adj_rc1 alone
    d2 = capi.DeviceSet.from_kmers(capi.geom(k, n), base, ctx.device)
    sp2 = ctx.spss_encode(d2, mode=0)
    assert "rc_marks_groups" not in ctx.spss_encode_routes()
    assert sp2.to_strings() == ol.Set.from_kmers(k, n, kb, base).spss()


def test_encode_repeat_rich(ctx):
    """A repeat-rich family (synth.plant_repeats: stretches of 50..500 bases copied to several loci): the unitig
    graph branches at both ends of every copy -- tens of thousands of unitigs, junctions with several candidate
    edges, a matching of several rounds, loops of the path cover to cut
    (lib/core/spss.h:1445-1644).  A set, the intersection with its sibling and a difference set: strings ==
    the oracle's, all three SPSS constructions."""
    k, n, kb = 23, 14, 4
    a, b = synth.phylogeny_sets(k, 2, 400_000, seed=31, rate=0.004, repeats=(1500, 6))
    rounds = []
    for name, kmers in (("set", a), ("intersection", np.intersect1d(a, b)), ("difference", np.setdiff1d(a, b))):
        d = capi.DeviceSet.from_kmers(capi.geom(k, n), kmers, ctx.device)
        o = ol.Set.from_kmers(k, n, kb, kmers)
        for mode in (0, 1, 2):
            sp = ctx.spss_encode(d, mode=mode)
            st = ctx.spss_encode_stats()
            want = o.spss() if mode == 0 else (o.unitigs() if mode == 1 else o.spss_slow())
            assert sp.to_strings() == want, (name, mode)
            if mode == 0:
                rounds.append(st["rounds"])
                assert st["unitigs"] > 2000 and st["strings"] < st["unitigs"], (name, st)
    assert max(rounds) >= 3  # (a random genome's matching is done after one or two rounds)


def test_u16_keys_medium_and_large_sets(ctx):
    """(15, 14, uint16_t) beyond the sizes of the small parity cases: the 256-thread and 512-thread variants of
    k_adj_rc with 2-byte keys (a window of an odd number of 2-byte keys put the 4-byte marks behind it on a
    2-byte boundary and faulted the kernel's compare-and-swap: `bench_loop.py --k 15 --sets 8 --size 3e7`, round
    3).  One set against the oracle string for string; larger ones, odd window sizes among them, through
    encode -> decode == the set."""
    from kmersets import synth_torch

    k, n, kb = 15, 14, 2
    g = capi.geom(k, n)
    kmers = synth.phylogeny_sets(k, 1, 4_400_001, seed=11)[0]
    d = capi.DeviceSet.from_kmers(g, kmers, ctx.device)
    assert ctx.spss_encode(d, mode=0).to_strings() == ol.Set.from_kmers(k, n, kb, kmers).spss()
    odd = 0
    for i, size in enumerate((10_000_001, 13_333_337, 17_000_003, 23_456_789, 30_000_001)):
        ks = synth_torch.phylogeny_sets(k, 1, size, 20 + i, ctx.device)[0]
        d = synth_torch.device_set(g, ks)
        window = (d.n_keys // (1 << n)) * 5 // 4 + 256  # what the host asks for before it rounds to 8 keys
        odd += window % 2
        back = ctx.spss_decode(ctx.spss_encode(d, mode=0))
        assert back.n_keys == d.n_keys and ctx.set_hash(back) == ctx.set_hash(d), size
        del ks, d, back
    assert odd >= 1  # (the sizes above are chosen so that the case that faulted is among them)


@pytest.mark.parametrize("knob", ["KSH_RANK=stamp", "KSH_EMIT=walk", "KSH_L2_MIN=4096", "KSH_ADJACENCY=probe", "KSH_FWD=probe", "KSH_FWD=staged", "KSH_RC1=marks", "KSH_RC1=batched", "KSH_RC_SCATTER=direct", "KSH_RC_GROUPS=half", "KSH_RANK_PHASES=1"])
def test_encode_alternative_paths(gpu, knob):
    """The encoder's other routes give the oracle's strings too: the stamping ranking walks with
    k_choose / k_emit per k-mer (what a set with a non-branching loop falls back to), the strings
    written by a second walk instead of from the ranking walks' logs, the two-level pointer jumping on
    sets of any size (by default it starts at 2^20 ruler records), the in-place
    neighbour probe, the in-place forward half, the one-pass record scatter, half-bucket groups, the
    ranking walks of large sets (all walkers in one launch) on small ones.  The switches are read
    once per process, so each runs in a process of its own."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from kmersets import capi, synth\n"
        "ctx = capi.Context(0)\n"
        "def check(k, n, kb, kmers, modes=(0,)):\n"
        "    d = capi.DeviceSet.from_kmers(capi.geom(k, n), kmers, ctx.device)\n"
        "    o = ol.Set.from_kmers(k, n, kb, kmers)\n"
        "    for mode in modes:\n"
        "        want = o.spss() if mode == 0 else (o.unitigs() if mode == 1 else o.spss_slow())\n"
        "        assert ctx.spss_encode(d, mode=mode).to_strings() == want, (k, n, mode)\n"
        "for (k, n, kb) in ((15, 14, 2), (23, 14, 4), (31, 14, 8)):\n"
        "    s = synth.phylogeny_sets(k, 3, 30000, seed=k)\n"
        "    check(k, n, kb, s[0], (0, 1)); check(k, n, kb, np.intersect1d(s[0], s[1])); check(k, n, kb, np.setdiff1d(s[0], s[1]))\n"
        "check(23, 14, 4, synth.genome_with_tips(23, 150000, seed=7, every=150), (0, 2))\n"
        "big = synth.phylogeny_sets(23, 2, 400000, seed=5); check(23, 14, 4, big[0]); check(23, 14, 4, np.intersect1d(big[0], big[1]))\n"
        "a, b = synth.phylogeny_sets(23, 2, 60000, seed=77, rate=0.004); check(23, 14, 4, np.union1d(a, b))\n"
        "x = np.random.default_rng(23).integers(0, 4 ** 23, size=6000, dtype=np.uint64)\n"
        "x = (x & ~np.uint64(0xFFFC)) | (np.uint64(0b10011100011011) << np.uint64(2))  # one reverse-complement group crowded\n"
        "check(23, 14, 4, np.unique(np.concatenate([big[1][:200000], synth.canonical(x, 23)])))\n"
        "for seed in range(24):\n"
        "    k = [5, 7, 9, 11][seed %% 4]\n"
        "    check(k, min(10, 2 * k - 4), 4, synth.circular_with_tails(k, 20 + (seed * 7) %% 150, seed %% 5, 1 + seed %% 4, seed))\n"
        "for seed in range(6):\n"
        "    k = [5, 7, 9, 9, 11, 15][seed]\n"
        "    check(k, min(10, 2 * k - 4), 4, synth.random_read_kmers(k, min([50, 300, 2000, 20000, 3000, 3000][seed], 4 ** k // 3), seed=seed, canonical=True))\n"
        "print('alternative path ok')\n"
    ) % (os.path.join(here, "..", "kmer-sets-compression_amd"), here)
    name, value = knob.split("=")
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "alternative path ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
