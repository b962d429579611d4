"""Pins the oracle (CPU restatement under oracle/) to the reference's own
known-answer tests and to the reference's property tests, replayed with seeded
inputs.  CPU only.

Reference tests mirrored (file:line in each test's docstring):
  test/kmer.cc, test/kmer_set.cc, test/spss.cc, test/kmer_set_compact.cc,
  test/kmer_set_set.cc, test/parallel_disjoint_set.cc, test/range.cc.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from kmersets import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def kmers_of(seq, k, canonical=True):
    L = ol.lib()
    out = []
    for i in range(len(seq) - k + 1):
        b = ol.kmer(seq[i:i + k])
        out.append(int(L.ko_canonical(b, k)) if canonical else b)
    return np.array(sorted(set(out)), dtype=np.uint64)


# ----------------------------------------------------------------------------- kmer.h
def test_kmer_string_roundtrip(golden):
    """test/kmer.cc:8-12"""
    for s in golden["kmer_string_roundtrip"]["cases"]:
        assert ol.kmer_str(ol.kmer(s), len(s)) == s


def test_kmer_canonical(golden):
    """test/kmer.cc:14-19"""
    for s, want in golden["kmer_canonical"]["cases"]:
        assert ol.kmer_str(ol.lib().ko_canonical(ol.kmer(s), len(s)), len(s)) == want


def test_kmer_complement(golden):
    """test/kmer.cc:21-24"""
    for s, want in golden["kmer_complement"]["cases"]:
        assert ol.kmer_str(ol.lib().ko_complement(ol.kmer(s), len(s)), len(s)) == want


def test_kmer_next_prev(golden):
    """test/kmer.cc:26-34"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for s, c, want in golden["kmer_next"]["cases"]:
        assert ol.kmer_str(ol.lib().ko_next(ol.kmer(s), len(s), code[c]), len(s)) == want
    for s, c, want in golden["kmer_prev"]["cases"]:
        assert ol.kmer_str(ol.lib().ko_prev(ol.kmer(s), len(s), code[c]), len(s)) == want


def test_string_complement(golden):
    """test/spss.cc:13"""
    import ctypes as C

    for s, want in golden["string_complement"]["cases"]:
        buf = C.create_string_buffer(s.encode())
        ol.lib().ko_complement_string(buf, len(s))
        assert buf.value.decode() == want


def test_complement_matches_bit_parallel():
    """kmer.h:103-129 (K-step loop) vs the bit-parallel form the device uses."""
    rng = np.random.default_rng(7)
    for k in (5, 9, 15, 19, 23, 31):
        x = rng.integers(0, 1 << (2 * k), size=2000, dtype=np.uint64)
        want = np.zeros_like(x)
        ol.lib().ko_complement_many(x, x.size, k, want)
        assert np.array_equal(synth.revcomp(x, k), want)


# ------------------------------------------------------------------------- kmer_set.h
def test_bucket_and_key_roundtrip(golden):
    """test/kmer_set.cc:10-23"""
    import ctypes as C

    g = golden["kmer_set_bucket_key"]
    for s in g["cases"]:
        b, key = C.c_int64(), C.c_uint64()
        ol.lib().ko_bucket_and_key(g["k"], g["n"], ol.kmer(s), C.byref(b), C.byref(key))
        back = ol.lib().ko_kmer_from_bucket_and_key(g["k"], g["n"], b.value, key.value)
        assert ol.kmer_str(back, g["k"]) == s
        assert b.value == ol.kmer(s) >> (2 * g["k"] - g["n"])


def test_kmer_set_add_remove():
    """test/kmer_set.cc:25-46"""
    s = ol.Set(5, 3, 1)
    a = ol.kmer("AAAAA")
    assert s.size() == 0 and not s.contains(a)
    s.add([a])
    assert s.size() == 1 and s.contains(a)
    s.remove([a])
    assert s.size() == 0 and not s.contains(a)


def test_kmer_set_find():
    """test/kmer_set.cc:48-70 (Find with a predicate == filter of all k-mers)."""
    s = ol.Set.from_kmers(5, 3, 1, [ol.kmer("AAAAA"), ol.kmer("CCCCC")])
    all_kmers = [ol.kmer_str(x, 5) for x in s.kmers()]
    assert [x for x in all_kmers if x[0] == "A"] == ["AAAAA"]
    assert [x for x in all_kmers if x[1] == "C"] == ["CCCCC"]


def test_kmer_set_operators(golden):
    """test/kmer_set.cc:72-94"""
    g = golden["kmer_set_operators"]
    s1 = ol.Set.from_kmers(g["k"], g["n"], g["key_bytes"], [ol.kmer(x) for x in g["set1"]])
    s2 = ol.Set.from_kmers(g["k"], g["n"], g["key_bytes"], [ol.kmer(x) for x in g["set2"]])
    assert s1.copy().add_set(s2).size() == g["add_size"]
    assert s1.copy().sub_set(s2).size() == g["sub12_size"]
    assert s2.copy().sub_set(s1).size() == g["sub21_size"]
    assert s2.intersection(s1).size() == g["intersection_size"]
    assert s1.intersection(s2).size() == g["intersection_size"]


def test_kmer_set_equals_and_diff(golden):
    """test/kmer_set.cc:96-124"""
    g = golden["kmer_set_equals"]
    s = [ol.Set.from_kmers(g["k"], g["n"], g["key_bytes"], [ol.kmer(x) for x in g[name]])
         for name in ("set1", "set2", "set3")]
    for x in s:
        assert x.equals(x)
    assert s[0].equals(s[1]) and s[1].equals(s[0])
    assert not s[0].equals(s[2]) and not s[2].equals(s[0])
    assert s[0].diff(s[2]) == 3


def test_survey_known_answers(golden):
    """outputs of the reference's own headers at n_workers=1."""
    g = golden["survey_known_answers"]
    k, n, kb = g["k"], g["n"], g["key_bytes"]
    for case in g["cases"]:
        kmers = kmers_of(case["sequence"], k)
        s = ol.Set.from_kmers(k, n, kb, kmers)
        assert s.size() == case["size"]
        assert s.hash() == case["hash"]
        if "kmers" in case:
            assert [ol.kmer_str(x, k) for x in s.kmers()] == case["kmers"]
        if "spss" in case:
            assert s.spss() == case["spss"]
        if "weight" in case:
            assert s.compact().weight() == case["weight"]


# ----------------------------------------------------------------------------- spss.h
def check_spss_invariants(strings, k, n, kb, want_set, canonical=True):
    """test/spss.cc:29-40,113-124,141-152: every string >= K long, no k-mer twice,
    union equals the input."""
    L = ol.lib()
    seen = set()
    for s in strings:
        assert len(s) >= k
        for i in range(len(s) - k + 1):
            c = ol.kmer(s[i:i + k])
            if canonical:
                c = int(L.ko_canonical(c, k))
            assert c not in seen
            seen.add(c)
    got = ol.Set.from_kmers(k, n, kb, np.array(sorted(seen), dtype=np.uint64))
    assert want_set.equals(got)


@pytest.mark.parametrize("seed", range(6))
def test_unitigs_and_spss_random(seed):
    """test/spss.cc:43-66 (GetUnitigsCanonicalRandom), :127-153 (GetSPSSCanonicalFastRandom),
    :172-189 (GetKmerSetFromSPSSCanonicalRandom); K=9, N=10, uint8 keys."""
    k, n, kb = 9, 10, 1
    size = [1, 17, 300, 4000, 20000, 65536][seed]
    kmers = synth.random_read_kmers(k, size, seed=100 + seed, canonical=True)
    s = ol.Set.from_kmers(k, n, kb, kmers)
    assert s.size() == kmers.size
    check_spss_invariants(s.unitigs(), k, n, kb, s)
    spss = s.spss()
    check_spss_invariants(spss, k, n, kb, s)
    assert ol.Set.from_spss(spss, k, n, kb).equals(s)


@pytest.mark.parametrize("seed", range(6))
def test_noncanonical_and_slow_random(seed):
    """test/spss.cc:15-40 (GetUnitigsRandom), :71-96 (GetSPSSRandom), :155-170
    (GetKmerSetFromSPSSRandom) on non-canonical sets, and :99-124 (GetSPSSCanonicalRandom,
    fast = false); K=9, N=10, uint8 keys."""
    k, n, kb = 9, 10, 1
    size = [1, 17, 300, 4000, 20000, 65536][seed]
    fw = synth.random_read_kmers(k, size, seed=200 + seed, canonical=False)
    f = ol.Set.from_kmers(k, n, kb, fw)
    check_spss_invariants(f.unitigs_directed(), k, n, kb, f, canonical=False)
    spss = f.spss_directed()
    check_spss_invariants(spss, k, n, kb, f, canonical=False)
    assert ol.Set.from_spss(spss, k, n, kb, canonical=False).equals(f)
    c = f.compact(canonical=False)
    assert c.size() == f.size() and c.weight() == sum(len(x) for x in spss)
    assert c.to_set(canonical=False).equals(f)
    cs = ol.Set.from_kmers(k, n, kb, synth.random_read_kmers(k, size, seed=300 + seed, canonical=True))
    slow = cs.spss_slow()
    check_spss_invariants(slow, k, n, kb, cs)
    assert cs.compact(fast=False).weight() == sum(len(x) for x in slow)
    assert ol.Set.from_spss(slow, k, n, kb).equals(cs)


def test_noncanonical_and_slow_dense_and_special():
    """Dense random graphs (k = 5: every branching pattern, loops in the path cover) and the
    special shapes, both constructions."""
    rng = np.random.default_rng(4)
    k, n, kb = 5, 3, 1
    for trial in range(20):
        fw = np.unique(rng.integers(0, 4 ** k, size=int(rng.integers(1, 1500)), dtype=np.uint64))
        f = ol.Set.from_kmers(k, n, kb, fw)
        check_spss_invariants(f.unitigs_directed(), k, n, kb, f, canonical=False)
        check_spss_invariants(f.spss_directed(), k, n, kb, f, canonical=False)
        cs = ol.Set.from_kmers(k, n, kb, np.unique(synth.canonical(fw, k)))
        check_spss_invariants(cs.spss_slow(), k, n, kb, cs)
    for seq in ["ACGTACGTACG", "AAAAAAAAA", "AACCGGTT", "ATATATATAT", "GATTACAGATTACAGATTACA"]:
        fw = np.unique(synth.kmers_of_bases(synth.bases_of_string(seq), k))
        f = ol.Set.from_kmers(k, n, kb, fw)
        check_spss_invariants(f.spss_directed(), k, n, kb, f, canonical=False)
    # a forward walk spells its k-mers in order: one string for a sequence without repeats
    f = ol.Set.from_kmers(k, n, kb, np.unique(synth.kmers_of_bases(synth.bases_of_string("AACCGTTAGCAT"), k)))
    assert f.spss_directed() == ["AACCGTTAGCAT"] and f.unitigs_directed() == ["AACCGTTAGCAT"]
    empty = ol.Set(k, n, kb)
    assert empty.unitigs_directed() == [] and empty.spss_directed() == [] and empty.spss_slow() == []


def test_kmer_set_set_noncanonical():
    """KmerSetSet(..., canonical = false, ...): Get(i) gives the inputs back
    (test/kmer_set_set.cc:30-34 with the flag the other way)."""
    k, n, kb = 15, 14, 2
    genomes = synth.phylogeny_genomes(4, 3000 + k - 1, seed=21)
    sets = [np.unique(synth.kmers_of_bases(g, k)) for g in genomes]
    compacts = [ol.Set.from_kmers(k, n, kb, s).compact(canonical=False) for s in sets]
    kss = ol.KmerSetSet(compacts, synth.sample_bucket_ids(n, seed=2), canonical=False)
    assert kss.size() > len(sets)
    for i, s in enumerate(sets):
        assert np.array_equal(kss.get(i).kmers(), s)


@pytest.mark.parametrize("geom", [(5, 3, 1), (15, 14, 2), (19, 10, 4), (23, 14, 4), (31, 14, 8)])
def test_spss_other_geometries(geom):
    """The CLI instantiations (src/kmerset-multiple-compress.cc:149-162) + (31,14,u64)."""
    k, n, kb = geom
    kmers = synth.random_read_kmers(k, 3000 if k > 5 else 200, seed=k, canonical=True)
    s = ol.Set.from_kmers(k, n, kb, kmers)
    spss = s.spss()
    check_spss_invariants(spss, k, n, kb, s)
    c = s.compact()
    assert c.size() == s.size()
    assert c.weight() == sum(len(x) for x in spss)
    assert c.to_set().equals(s)


def test_spss_special_shapes():
    """Loops, hairpins, isolated k-mers, the empty set (SURVEY.md 8c fixtures list)."""
    k, n, kb = 5, 3, 1
    for seq in ["ACGTACGTACG", "AAAAAAAAA", "AACCGGTT", "ACGTTGCAACGT", "ATATATATAT",
                "GATTACAGATTACAGATTACA", "CCCCCCGGGGGG"]:
        s = ol.Set.from_kmers(k, n, kb, kmers_of(seq, k))
        check_spss_invariants(s.unitigs(), k, n, kb, s)
        check_spss_invariants(s.spss(), k, n, kb, s)
    empty = ol.Set(k, n, kb)
    assert empty.unitigs() == [] and empty.spss() == []
    assert empty.compact().size() == 0 and empty.compact().weight() == 0


# ------------------------------------------------------------------- kmer_set_compact.h
def test_compact_dump_load_size_sampled(tmp_path):
    """test/kmer_set_compact.cc:15-129; n = 100000 there, 20000 here (same shape)."""
    k, n, kb = 9, 10, 1
    kmers = synth.random_read_kmers(k, 20000, seed=5, canonical=True)
    s = ol.Set.from_kmers(k, n, kb, kmers)
    c = s.compact()
    path = str(tmp_path / "x.txt")
    c.dump(path)
    with open(path) as f:
        lines = f.read().split("\n")
    assert lines[-1] == "" and set("".join(lines)) <= set("ACGT")
    assert ol.Compact.load(path, k, n, kb).to_set().equals(s)
    assert c.size() == s.size()
    assert c.to_set().equals(s)
    ids = np.arange(1 << n, dtype=np.int32)[::-1].copy()
    offsets, keys = c.sampled(ids)
    rebuilt = []
    for i, b in enumerate(ids):
        seg = keys[offsets[i]:offsets[i + 1]]
        assert np.all(seg[:-1] < seg[1:])
        rebuilt.append((np.uint64(b) << np.uint64(2 * k - n)) + seg)
    assert np.array_equal(np.sort(np.concatenate(rebuilt)), s.kmers())


def test_streamvbyte_0124_roundtrip():
    """kmer_set_compact.h:257-265,272 call sites; format restated in oracle/ko_compact.h."""
    L = ol.lib()
    rng = np.random.default_rng(3)
    for n in (0, 1, 3, 4, 5, 1000):
        v = rng.choice(np.array([0, 1, 255, 256, 65535, 65536, 2**32 - 1], dtype=np.uint32), size=n)
        v = np.ascontiguousarray(v, dtype=np.uint32)
        buf = np.zeros(max(1, L.ko_svb_max_compressed_bytes(n)), dtype=np.uint8)
        size = L.ko_svb_encode_0124(v if n else np.zeros(1, np.uint32), n, buf)
        want = (n + 3) // 4 + int(np.sum((v > 0) * 1 + (v > 255) * 1 + (v > 65535) * 2))
        assert size == want
        back = np.zeros(max(n, 1), dtype=np.uint32)
        assert L.ko_svb_decode_0124(buf, back, n) == size
        assert np.array_equal(back[:n], v)
    # a hand-checked vector: values 0, 1, 256, 65536
    v = np.array([0, 1, 256, 65536], dtype=np.uint32)
    buf = np.zeros(32, dtype=np.uint8)
    size = L.ko_svb_encode_0124(v, 4, buf)
    assert size == 1 + 0 + 1 + 2 + 4
    assert buf[0] == (0 | (1 << 2) | (2 << 4) | (3 << 6))
    assert list(buf[1:8]) == [1, 0, 1, 0, 0, 1, 0]


# ----------------------------------------------------------------------- kmer_set_set.h
def make_family(k, n, kb, n_sets, size, seed):
    sets = synth.phylogeny_sets(k, n_sets, size, seed=seed)
    return [ol.Set.from_kmers(k, n, kb, x) for x in sets]


def test_kmer_set_set_get_dump_load(tmp_path):
    """test/kmer_set_set.cc:15-123 (10 sets x 10000, K=9 there; a correlated family
    here, because independent random sets have nothing to merge)."""
    k, n, kb = 9, 10, 1
    sets = make_family(k, n, kb, 6, 3000, seed=11)
    compacts = [s.compact() for s in sets]
    ids = synth.sample_bucket_ids(n, seed=1)
    kss = ol.KmerSetSet(compacts, ids)
    assert kss.size() >= len(sets)
    for i, s in enumerate(sets):
        assert kss.get(i).equals(s)
    d = str(tmp_path / "out")
    kss.dump(d, "txt")
    meta = open(os.path.join(d, "meta.txt")).read().split("\n")
    assert meta[0] == kss.meta() and int(meta[1]) == kss.size()
    loaded = ol.KmerSetSet.load(d, "txt", k, n, kb)
    assert loaded.size() == kss.size()
    for i in range(kss.size()):
        assert loaded.get(i).equals(kss.get(i))


def test_kmer_set_set_merges_and_trace():
    k, n, kb = 15, 14, 2
    sets = make_family(k, n, kb, 8, 20000, seed=3)
    ids = synth.sample_bucket_ids(n, seed=2)
    kss = ol.KmerSetSet([s.compact() for s in sets], ids)
    it = kss.iterations()
    assert len(it) > 0, "a correlated family must merge at least once"
    # every merge takes the arg-max pair and shrinks the stored k-mer total
    assert np.all(it[:, 4] < 0)
    assert kss.stat(1) == kss.stat(0) + int(it[:, 4].sum())
    assert kss.stat(3) == kss.stat(0) + int(it[:, 3].sum())
    for i, s in enumerate(sets):
        assert kss.get(i).equals(s)


# ------------------------------------------------------------ parallel_disjoint_set.h
def test_disjoint_set_matches_serial():
    """test/parallel_disjoint_set.cc:14-180: same partition as a naive serial DSU,
    unions issued from 8 threads."""
    L = ol.lib()
    rng = np.random.default_rng(9)
    n = 2000
    xs = rng.integers(0, n, size=1500).astype(np.int32)
    ys = rng.integers(0, n, size=1500).astype(np.int32)
    d = L.ko_dsu_new(n)
    L.ko_dsu_unite_parallel(d, xs, ys, xs.size, 8)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in zip(xs, ys):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[ra] = rb
    fwd, bwd = {}, {}
    for i in range(n):
        a, b = L.ko_dsu_find(d, i), find(i)
        assert fwd.setdefault(a, b) == b and bwd.setdefault(b, a) == a
    for _ in range(2000):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        assert bool(L.ko_dsu_is_same(d, a, b)) == (find(a) == find(b))
    L.ko_dsu_free(d)


# -------------------------------------------------------------------------------- range.h
def test_range_split_exhaustive():
    """test/range.cc: Split covers [begin, end) contiguously for all small cases."""
    L = ol.lib()
    for begin in range(0, 12):
        for end in range(begin + 1, 40, 3):
            for n in range(1, 20):
                b = np.zeros(n, dtype=np.int64)
                e = np.zeros(n, dtype=np.int64)
                L.ko_range_split(begin, end, n, b, e)
                assert b[0] == begin and e[-1] == end
                assert np.array_equal(b[1:], e[:-1])
                sizes = e - b
                assert sizes.max() - sizes.min() <= 1 and np.all(np.diff(sizes) >= 0)


# ---- k-mer counter (the step before the loop): test/kmer_counter.cc ------------------------------
def test_kmer_counter_known_answers():
    g = json.load(open(GOLDEN))["kmer_counter"]
    k, n, kb = 5, 3, 1
    c = ol.Counter(k, n, kb)
    c.add(ol.kmer("AAAAA"), g["add_with_max"]["x"])
    c.add(ol.kmer("AAAAA"), g["add_with_max"]["y"])
    assert c.get(ol.kmer("AAAAA")) == g["add_with_max"]["want"]
    c = ol.Counter(k, n, kb)
    for s, v in g["add_and_get"]["adds"]:
        c.add(ol.kmer(s), v)
    for s, want in g["add_and_get"]["want"].items():
        assert c.get(ol.kmer(s)) == want
    c = ol.Counter(k, n, kb)
    for s, v in g["to_kmer_set"]["adds"]:
        c.add(ol.kmer(s), v)
    s_, cut = c.to_set(g["to_kmer_set"]["cutoff"])
    assert cut == g["to_kmer_set"]["want_cutoff_count"]
    assert sorted(ol.kmer_str(x, k) for x in s_.kmers()) == sorted(g["to_kmer_set"]["want_set"])
    c = ol.Counter(k, n, kb)
    c.from_reads(g["from_reads"]["reads"], canonical=g["from_reads"]["canonical"])
    for s, want in g["from_reads"]["want"].items():
        assert c.get(ol.kmer(s)) == want
    assert c.size() == len(g["from_reads"]["want"])


def test_kmer_counter_fasta_rules():
    """FromFASTA's checks (kmer_counter.h:157-190) and the 'N' split (:79)."""
    k, n, kb = 5, 3, 1
    ok = b">r1\nAACCGTTNNAACCGTA\n>r2 some text\nACGTNACGTACGT\n"
    c = ol.Counter(k, n, kb)
    assert c.from_fasta(ok, canonical=False) == 0
    assert c.get(ol.kmer("AACCG")) == 2 and c.get(ol.kmer("CCGTA")) == 1
    assert c.get(ol.kmer("ACGTA")) == 1 and c.get(ol.kmer("ACGTN".replace("N", "A"))) == 1
    assert c.get(ol.kmer("GTTAA")) == 0                      # windows never cross an N
    assert ol.Counter(k, n, kb).from_fasta(ok[:-1], canonical=True) == 0      # no final newline
    assert ol.Counter(k, n, kb).from_fasta(b"", canonical=True) == 0          # no lines at all
    assert ol.Counter(k, n, kb).from_fasta(b">r1\nACGTA\n>r2\n", canonical=True) == 1
    assert ol.Counter(k, n, kb).from_fasta(b"r1\nACGTA\n", canonical=True) == 2
    assert ol.Counter(k, n, kb).from_fasta(b"\nACGTA\n", canonical=True) == 2
    assert ol.Counter(k, n, kb).from_fasta(b">r1\nACGTa\n", canonical=True) == 2
    assert ol.Counter(k, n, kb).from_fasta(b">r1\nACGTA\r\n", canonical=True) == 2
    assert ol.Counter(k, n, kb).from_fasta(b">r1\n\n", canonical=True) == 0   # an empty read is fine
