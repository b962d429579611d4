"""GPU parity tests: the HIP path, called through the C ABI (include/kmersets_hip.h),
against the oracle (CPU restatement) on the same seeded inputs, against the
reference's known answers (tests/golden), and -- at BASELINE config-2 size --
through size-independent properties.  Bit-exact: integer / index work only.

Run with `pytest -m gpu` on an MI355X.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from kmersets import capi, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")
GEOMS = [(5, 3, 1), (9, 10, 1), (15, 14, 2), (19, 10, 4), (23, 14, 4), (31, 14, 8)]


@pytest.fixture(scope="module")
def ctx(gpu):
    c = capi.Context(0)
    yield c
    c.close()


def dev_set(ctx, k, n, kmers):
    return capi.DeviceSet.from_kmers(capi.geom(k, n), np.asarray(kmers, dtype=np.uint64), ctx.device)


def check_algebra(ctx, k, n, kb, ka, kbm, use_oracle=True):
    """A&B, A\\B, B\\A, Diff, Hash of one pair: GPU vs numpy set algebra (+ oracle)."""
    a, b = dev_set(ctx, k, n, ka), dev_set(ctx, k, n, kbm)
    i_d, amb_d, bma_d = ctx.pair_algebra(a, b)
    want_i = np.intersect1d(ka, kbm)
    want_amb = np.setdiff1d(ka, kbm)
    want_bma = np.setdiff1d(kbm, ka)
    for got, want in ((i_d, want_i), (amb_d, want_amb), (bma_d, want_bma)):
        assert got.n_keys == want.size
        off, keys = got.to_numpy()
        w_off, w_keys = synth.to_bucketed(want, k, n, got.g.key_bytes)
        assert np.array_equal(off, w_off)
        assert np.array_equal(keys, w_keys)
    # the one-call form (upper-bound buffers) must give the same three sets
    for got, want in zip(ctx.pair_algebra_onepass(a, b), (want_i, want_amb, want_bma)):
        assert got.n_keys == want.size
        off, keys = got.to_numpy()
        w_off, w_keys = synth.to_bucketed(want, k, n, got.g.key_bytes)
        assert np.array_equal(off, w_off)
        assert np.array_equal(keys, w_keys)
    assert ctx.set_diff(a, b) == want_amb.size + want_bma.size
    for d, want in ((a, ka), (b, kbm), (i_d, want_i)):
        h = 0
        if want.size:
            h = int(np.bitwise_xor.reduce(want.astype(np.uint64)))
        assert ctx.set_hash(d) == h
    if use_oracle:
        oa, ob = ol.Set.from_kmers(k, n, kb, ka), ol.Set.from_kmers(k, n, kb, kbm)
        oi = oa.intersection(ob)
        assert oi.size() == i_d.n_keys and oi.hash() == ctx.set_hash(i_d)
        assert np.array_equal(oi.kmers(), i_d.kmers())
        assert np.array_equal(oa.copy().sub_set(oi).kmers(), amb_d.kmers())
        assert np.array_equal(ob.copy().sub_set(oi).kmers(), bma_d.kmers())
        assert oa.diff(ob) == ctx.set_diff(a, b)
        assert oa.hash() == ctx.set_hash(a)


def test_reference_known_answers(ctx):
    """test/kmer_set.cc:72-124 through the HIP path."""
    g = json.load(open(GOLDEN))
    op = g["kmer_set_operators"]
    k, n = op["k"], op["n"]
    s1 = np.array(sorted(ol.kmer(x) for x in op["set1"]), dtype=np.uint64)
    s2 = np.array(sorted(ol.kmer(x) for x in op["set2"]), dtype=np.uint64)
    a, b = dev_set(ctx, k, n, s1), dev_set(ctx, k, n, s2)
    i_d, amb_d, bma_d = ctx.pair_algebra(a, b)
    assert i_d.n_keys == op["intersection_size"]
    assert amb_d.n_keys == op["sub12_size"] and bma_d.n_keys == op["sub21_size"]
    assert a.n_keys + bma_d.n_keys == op["add_size"]
    eq = g["kmer_set_equals"]
    sets = [np.array(sorted(ol.kmer(x) for x in eq[nm]), dtype=np.uint64) for nm in ("set1", "set2", "set3")]
    d = [dev_set(ctx, k, n, s) for s in sets]
    assert ctx.set_diff(d[0], d[1]) == 0 and ctx.set_diff(d[1], d[0]) == 0
    assert ctx.set_diff(d[0], d[2]) != 0 and ctx.set_diff(d[2], d[0]) != 0
    for x in d:
        assert ctx.set_diff(x, x) == 0
    sv = g["survey_known_answers"]
    for case in sv["cases"]:
        seq = synth.bases_of_string(case["sequence"])
        km = synth.canonical_set_of_bases(seq, sv["k"])
        s = dev_set(ctx, sv["k"], sv["n"], km)
        assert s.n_keys == case["size"] and ctx.set_hash(s) == case["hash"]


@pytest.mark.parametrize("geom", GEOMS)
def test_pair_algebra_vs_oracle(ctx, geom):
    k, n, kb = geom
    size = 150 if k == 5 else 20000
    sets = synth.phylogeny_sets(k, 4, size, seed=k)
    for (x, y) in [(0, 1), (0, 3), (2, 2)]:
        check_algebra(ctx, k, n, kb, sets[x], sets[y])
    ua, ub = synth.uniform_pair(k, min(size, 4 ** k // 4), 0.5, seed=3 * k)
    check_algebra(ctx, k, n, kb, ua, ub)


@pytest.mark.parametrize("geom", GEOMS)
def test_contains_and_find_vs_oracle(ctx, geom):
    """KmerSet::Contains (batched) and Find(n_workers) through the C ABI (ksh_set_contains,
    ksh_set_kmers) against the oracle's hash-bucket set: members, near misses (a member's four
    Next / Prev candidates, as the unitig walk asks), values in empty buckets, and the empty set."""
    k, n, kb = geom
    size = 150 if k == 5 else 20000
    kmers = synth.phylogeny_sets(k, 2, size, seed=k + 40)[0]
    d = dev_set(ctx, k, n, kmers)
    o = ol.Set.from_kmers(k, n, kb, kmers)
    mask = np.uint64((1 << (2 * k)) - 1)
    some = kmers[:: max(1, kmers.size // 500)]
    nexts = np.concatenate([((some << np.uint64(2)) & mask) | np.uint64(c) for c in range(4)])
    prevs = np.concatenate([(some >> np.uint64(2)) | (np.uint64(c) << np.uint64(2 * k - 2)) for c in range(4)])
    rnd = synth.mix64(np.arange(2000, dtype=np.uint64) + np.uint64(k)) & mask
    queries = np.concatenate([some, nexts, prevs, rnd, np.array([0, int(mask)], dtype=np.uint64)])
    got = ctx.set_contains(d, queries)
    want = np.array([bool(o.contains(int(q))) for q in queries])
    assert np.array_equal(got, want)
    assert got[: some.size].all() and not got.all()
    assert np.array_equal(ctx.set_kmers(d), o.kmers())
    empty = dev_set(ctx, k, n, np.zeros(0, dtype=np.uint64))
    assert not ctx.set_contains(empty, queries[:50]).any() and ctx.set_kmers(empty).size == 0
    assert ctx.set_contains(d, np.zeros(0, dtype=np.uint64)).size == 0


@pytest.mark.parametrize("geom", [(23, 14, 4), (31, 14, 8)])
def test_pair_algebra_edge_cases(ctx, geom):
    k, n, kb = geom
    empty = np.zeros(0, dtype=np.uint64)
    some = synth.phylogeny_sets(k, 1, 5000, seed=1)[0]
    check_algebra(ctx, k, n, kb, empty, empty)
    check_algebra(ctx, k, n, kb, empty, some)
    check_algebra(ctx, k, n, kb, some, empty)
    check_algebra(ctx, k, n, kb, some, some)
    check_algebra(ctx, k, n, kb, some[::2].copy(), some[1::2].copy())   # disjoint, interleaved
    # one heavy bucket, far more keys than one tile, with every kind of tile boundary:
    # dense runs of common keys so that tile splits land inside "a, b" pairs.
    kbits = 2 * k - n
    base = np.uint64(5) << np.uint64(kbits)
    dense = base + np.arange(0, 40000, dtype=np.uint64)
    check_algebra(ctx, k, n, kb, dense, dense, use_oracle=False)
    check_algebra(ctx, k, n, kb, dense[dense % np.uint64(3) != 0], dense[dense % np.uint64(5) != 0],
                  use_oracle=False)
    check_algebra(ctx, k, n, kb, dense[:30000], dense[10000:], use_oracle=False)
    # max key values in the last bucket
    top = (np.uint64(1) << np.uint64(2 * k)) - np.uint64(1)
    hi = top - np.arange(0, 3000, dtype=np.uint64)[::-1]
    check_algebra(ctx, k, n, kb, hi, hi[::3].copy(), use_oracle=False)
    # the batch form (all pairs tiled together, one count and one write launch) on a mix of
    # empty, identical, heavy-bucket and ordinary pairs, against the pair-at-a-time form
    other = synth.phylogeny_sets(k, 2, 5000, seed=1)[1]
    members = [empty, some, other, dense[:30000], dense[10000:], hi]
    d = [dev_set(ctx, k, n, x) for x in members]
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (3, 4), (4, 3), (5, 1), (2, 5)]
    got = ctx.pair_algebra_batch([(d[i], d[j]) for i, j in pairs])
    for (i, j), trio in zip(pairs, got):
        want = (np.intersect1d(members[i], members[j]), np.setdiff1d(members[i], members[j]),
                np.setdiff1d(members[j], members[i]))
        for s_, w in zip(trio, want):
            assert s_.n_keys == w.size
            off, keys = s_.to_numpy()
            w_off, w_keys = synth.to_bucketed(w, k, n, s_.g.key_bytes)
            assert np.array_equal(off, w_off) and np.array_equal(keys, w_keys)


def test_pair_weights_vs_oracle(ctx):
    """GetEdgeWeight over sampled buckets (kmer_set_set.h:158-219): the initial weight
    table of the oracle's KmerSetSet == ksh_pair_weights on resident sets."""
    k, n, kb = 15, 14, 2
    sets = synth.phylogeny_sets(k, 6, 30000, seed=21)
    ids = synth.sample_bucket_ids(n, seed=4)
    osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
    kss = ol.KmerSetSet([s.compact() for s in osets], ids, max_iterations=0)
    want = kss.initial_weights(len(sets))
    dsets = [dev_set(ctx, k, n, s) for s in sets]
    pairs = [(i, j) for i in range(len(sets)) for j in range(i + 1, len(sets))]
    got = ctx.pair_weights(dsets, ids, pairs)
    assert np.array_equal(got, want)
    assert got.sum() > 0
    # a pair list with repeats, self pairs and reversed order
    pairs2 = [(3, 3), (5, 0), (0, 5), (1, 2)]
    got2 = ctx.pair_weights(dsets, ids, pairs2)
    want_self = sum(np.count_nonzero((sets[3] >> np.uint64(2 * k - n)) == np.uint64(b)) for b in ids)
    assert got2[0] == want_self and got2[1] == got2[2] == want[pairs.index((0, 5))]
    assert got2[3] == want[pairs.index((1, 2))]


def test_config2_properties(ctx):
    """BASELINE config 2 at its full size (4 x 10^7 canonical k=23 sets, all 6 pairs) through
    size-independent properties; KMERSETS_TEST_SIZE sizes it down."""
    import torch

    k, n = 23, 14
    size = int(float(os.environ.get("KMERSETS_TEST_SIZE", "1e7")))
    sets = synth.phylogeny_sets(k, 4, size, seed=2)
    d = [dev_set(ctx, k, n, s) for s in sets]
    hashes = [ctx.set_hash(x) for x in d]
    for i in range(4):
        assert hashes[i] == int(np.bitwise_xor.reduce(sets[i]))
    for i in range(4):
        for j in range(i + 1, 4):
            a, b = d[i], d[j]
            inter, amb, bma = ctx.pair_algebra(a, b)
            i1, a1, b1 = ctx.pair_algebra_onepass(a, b)
            i2, a2, b2 = ctx.pair_algebra_batch([(a, b), (b, a)])[1]     # batch form, swapped pair
            assert ctx.set_diff(i2, inter) == 0 and ctx.set_diff(a2, bma) == 0 and ctx.set_diff(b2, amb) == 0
            assert ctx.set_diff(i1, inter) == 0 and ctx.set_diff(a1, amb) == 0 and ctx.set_diff(b1, bma) == 0
            assert bool(torch.equal(i1.offsets, inter.offsets)) and bool(torch.equal(b1.offsets, bma.offsets))
            assert inter.n_keys + amb.n_keys == a.n_keys
            assert inter.n_keys + bma.n_keys == b.n_keys
            assert ctx.set_hash(inter) ^ ctx.set_hash(amb) == hashes[i]
            assert ctx.set_hash(inter) ^ ctx.set_hash(bma) == hashes[j]
            assert ctx.set_diff(a, b) == amb.n_keys + bma.n_keys
            assert ctx.set_diff(amb, bma) == amb.n_keys + bma.n_keys      # disjoint
            assert ctx.set_diff(inter, amb) == inter.n_keys + amb.n_keys  # disjoint
            # idempotence: (A & B) & B == A & B, (A \ B) \ B == A \ B
            i2, r, _ = ctx.pair_algebra(inter, b)
            assert i2.n_keys == inter.n_keys and r.n_keys == 0
            assert ctx.set_diff(i2, inter) == 0
            _, amb2, _ = ctx.pair_algebra(amb, b)
            assert ctx.set_diff(amb2, amb) == 0
            # sortedness inside every bucket and consistent offsets
            for s in (inter, amb, bma):
                off = s.offsets
                assert int(off[0]) == 0 and int(off[-1]) == s.n_keys
                assert bool(torch.all(off[1:] >= off[:-1]))
                km = torch.from_numpy(s.kmers().astype(np.int64))
                assert bool(torch.all(km[1:] > km[:-1]))
            # exact membership against numpy on this pair
            assert np.array_equal(inter.kmers(), np.intersect1d(sets[i], sets[j]))


def test_large_pair_properties(ctx):
    """Two device-generated k=23 sets of 6.5e7 k-mers: more than 2^17 tiles, so the tile-count
    prefix takes the chained scan with more than 64 workgroups and the multi-launch scan is
    used for nothing; several tiles per bucket.  Checked through identities that need no host
    copy of the sets."""
    import torch
    from kmersets import synth_torch

    k, n = 23, 14
    size = int(float(os.environ.get("KMERSETS_TEST_LARGE", "6.5e7")))
    g = capi.geom(k, n)
    ka, kb = synth_torch.phylogeny_sets(k, 2, size, seed=9, device=ctx.device)
    a, b = synth_torch.device_set(g, ka), synth_torch.device_set(g, kb)
    # expected sizes and hashes from torch on the sorted k-mer tensors
    both = torch.cat([ka, kb]).sort().values
    dup = both[1:] == both[:-1]
    want_i = both[1:][dup]
    xor = lambda t: int(np.bitwise_xor.reduce(t.cpu().numpy().astype(np.uint64))) if t.numel() else 0
    h_a, h_b, h_i = xor(ka), xor(kb), xor(want_i)
    del both, dup
    assert ctx.set_hash(a) == h_a and ctx.set_hash(b) == h_b
    inter, amb, bma = ctx.pair_algebra(a, b)
    assert inter.n_keys == want_i.numel() and ctx.set_hash(inter) == h_i
    assert inter.n_keys + amb.n_keys == a.n_keys and inter.n_keys + bma.n_keys == b.n_keys
    assert ctx.set_hash(amb) == h_a ^ h_i and ctx.set_hash(bma) == h_b ^ h_i
    assert ctx.set_diff(a, b) == amb.n_keys + bma.n_keys
    i2 = ctx.pair_algebra_batch([(b, a)])[0]
    assert ctx.set_diff(i2[0], inter) == 0 and ctx.set_diff(i2[1], bma) == 0 and ctx.set_diff(i2[2], amb) == 0
    key_bits = 2 * k - n
    for s_, want in ((inter, want_i), (amb, None), (bma, None)):
        off = s_.offsets
        assert int(off[0]) == 0 and int(off[-1]) == s_.n_keys and bool(torch.all(off[1:] >= off[:-1]))
        keys = s_.keys[: s_.n_keys * 4].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        bucket = torch.repeat_interleave(torch.arange(1 << n, device=keys.device), off[1:] - off[:-1])
        km = (bucket << key_bits) | keys
        assert bool(torch.all(km[1:] > km[:-1]))
        if want is not None:
            assert bool(torch.equal(km, want))
    # union through the same tiles: |A u B| and its hash
    u = ctx.set_union(a, b)
    assert u.n_keys == a.n_keys + bma.n_keys and ctx.set_hash(u) == h_a ^ h_b ^ h_i


def test_device_generator_matches_host(ctx):
    """bench.py builds its large inputs on the GPU; same sets as the numpy generator."""
    from kmersets import synth_torch

    for k, n in ((23, 14), (31, 14), (15, 14)):
        host = synth.phylogeny_sets(k, 4, 20000, seed=6)
        dev = synth_torch.phylogeny_sets(k, 4, 20000, seed=6, device=ctx.device)
        g = capi.geom(k, n)
        for h, d in zip(host, dev):
            assert np.array_equal(d.cpu().numpy().astype(np.uint64), h)
            ds = synth_torch.device_set(g, d)
            assert ds.n_keys == h.size and np.array_equal(ds.kmers(), h)
            assert ctx.set_hash(ds) == int(np.bitwise_xor.reduce(h))
