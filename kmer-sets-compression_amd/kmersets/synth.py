"""Seeded synthetic inputs for tests and bench.py (host-side plumbing, numpy only).

Everything is counter-based (splitmix64 finaliser over (seed, index)), so the same
arrays come out here, on the GPU box and in later rounds, independent of numpy's
Generator streams.

* random_read_kmers    -- the shape of the reference's test generator
                          (lib/random.h:37-110: reads of 1..100 random k-mers,
                          doubled half of the time to create loops).
* phylogeny_sets       -- SURVEY.md 8(d): ancestor genome, balanced binary tree,
                          independent substitutions per tree edge; leaf i's set is
                          the canonical k-mers of its genome, deduplicated.
* uniform_pair         -- the unstructured stress case of config 2: i.i.d. uniform
                          canonical k-mers, a chosen fraction shared.
* sample_bucket_ids    -- explicit, seeded replacement for GetRandomInts
                          (lib/core/random.h:12-41, kmer_set_set.h:123-124).
"""
import numpy as np

U = np.uint64
_M1, _M2, _G = U(0xBF58476D1CE4E5B9), U(0x94D049BB133111EB), U(0x9E3779B97F4A7C15)


def mix64(x):
    """splitmix64 finaliser, vectorised over uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=U) + _G
        x = (x ^ (x >> U(30))) * _M1
        x = (x ^ (x >> U(27))) * _M2
        return x ^ (x >> U(31))


def revcomp(x, k):
    """Bit-parallel reverse complement of 2-bit packed k-mers (uint64 array)."""
    x = ~np.asarray(x, dtype=U)
    x = ((x >> U(2)) & U(0x3333333333333333)) | ((x & U(0x3333333333333333)) << U(2))
    x = ((x >> U(4)) & U(0x0F0F0F0F0F0F0F0F)) | ((x & U(0x0F0F0F0F0F0F0F0F)) << U(4))
    x = x.byteswap()
    return x >> U(64 - 2 * k)


def canonical(x, k):
    return np.minimum(np.asarray(x, dtype=U), revcomp(x, k))


def kmers_of_bases(bases, k):
    """All k-mers (forward strand, in order) of a uint8 base array (values 0..3)."""
    n = bases.size - k + 1
    if n <= 0:
        return np.zeros(0, dtype=U)
    out = np.zeros(n, dtype=U)
    for j in range(k):
        out = (out << U(2)) | bases[j:j + n].astype(U)
    return out


def canonical_set_of_bases(bases, k):
    return np.unique(canonical(kmers_of_bases(bases, k), k))


def bases_of_string(s):
    return np.frombuffer(s.encode().translate(bytes.maketrans(b"ACGT", b"\x00\x01\x02\x03")),
                         dtype=np.uint8)


def string_of_bases(b):
    return np.asarray(b, dtype=np.uint8).tobytes().translate(
        bytes.maketrans(b"\x00\x01\x02\x03", b"ACGT")).decode()


def random_genome(length, seed):
    idx = np.arange(length, dtype=U)
    return (mix64(mix64(U(seed)) + idx) & U(3)).astype(np.uint8)


def mutate(bases, rate, edge_seed):
    """Independent substitutions with probability `rate` per base."""
    idx = np.arange(bases.size, dtype=U)
    with np.errstate(over="ignore"):
        r = mix64(mix64(np.array([edge_seed], dtype=U) * U(0xD6E8FEB86659FD93) + U(1)) + idx)
    hit = (r >> U(11)).astype(np.float64) < rate * float(1 << 53)
    shift = (U(1) + (r & U(0xFFFF)) % U(3)).astype(np.uint8)
    out = bases.copy()
    out[hit] = (bases[hit] + shift[hit]) & 3
    return out


def plant_repeats(bases, n_segments, copies, seed, min_len=50, max_len=500):
    """A repeat-rich ancestor: `n_segments` source stretches of min_len .. max_len bases (short ones likelier: the
    length is min_len + (max_len - min_len) * u^3) are each copied to `copies` other places of the genome, later
    copies over earlier ones where they meet.  Every copy puts the segment's k-mers at one more locus, so the
    compacted de Bruijn graph branches at both ends of every copy: about 2 * n_segments * copies more unitigs and
    junctions with several candidate edges for the greedy path cover (lib/core/spss.h:1445-1644), which a random
    genome hardly has.  Counter-based hashes of (seed, segment, copy): the torch twin plants the same bases."""
    n = bases.size
    out = bases.copy()
    seg = np.arange(n_segments, dtype=U)
    with np.errstate(over="ignore"):
        h0 = mix64(mix64(U(seed) * U(0xA24BAED4963EE407) + U(7)) + seg * U(3))
        h1 = mix64(h0 + U(1))
    u = (h1 >> U(11)).astype(np.float64) / float(1 << 53)
    length = (min_len + (max_len - min_len) * u ** 3).astype(np.int64)
    src = (h0 % U(max(1, n - max_len))).astype(np.int64)
    for c in range(copies):
        with np.errstate(over="ignore"):
            dst = (mix64(h0 + U(0x9E3779B97F4A7C15) * U(c + 2)) % U(max(1, n - max_len))).astype(np.int64)
        # all of a round's copies read the genome as it was BEFORE the planting (sources are never rewritten
        # mid-round), and land in ascending segment order
        at = np.repeat(dst - np.cumsum(length) + length, length) + np.arange(int(length.sum()))
        frm = np.repeat(src - np.cumsum(length) + length, length) + np.arange(int(length.sum()))
        out[at] = bases[frm]
    return out


def phylogeny_genomes(n_sets, length, seed, rate=0.002, repeats=None):
    genomes = [random_genome(length, 0x5EED0000 + seed)]
    if repeats:
        genomes[0] = plant_repeats(genomes[0], repeats[0], repeats[1], seed)
    edge = 0
    while len(genomes) < n_sets:
        nxt = []
        for g in genomes:
            for _ in range(2):
                edge += 1
                nxt.append(mutate(g, rate, (seed << 20) + edge))
        genomes = nxt
    return genomes[:n_sets]


def phylogeny_sets(k, n_sets, size, seed, rate=0.002, repeats=None):
    """n_sets sorted arrays of canonical k-mers (uint64); |S_i| a little under `size`.  repeats = (segments,
    copies): the ancestor genome carries planted repeats (plant_repeats)."""
    return [canonical_set_of_bases(g, k)
            for g in phylogeny_genomes(n_sets, size + k - 1, seed, rate, repeats)]


def genome_with_tips(k, size, seed, every=400):
    """Canonical k-mers of a random genome plus one-k-mer tips (the k-mer at every `every`-th position
    gets a second successor ending in T where the genome goes on with A, C or G): the greedy path cover
    takes the genome's own edge first, so the unitigs between the tips are stitched into long strings."""
    bases = random_genome(size + k - 1, 0x5EED0000 + seed)
    fwd = kmers_of_bases(bases, k)
    pos = np.arange(0, fwd.size - 1, every)
    keep = bases[pos + k] != 3
    mask = U((1 << (2 * k)) - 1)
    tips = ((fwd[pos[keep]] << U(2)) & mask) | U(3)
    return np.unique(canonical(np.concatenate([fwd, tips]), k))


def uniform_pair(k, size, shared_fraction, seed):
    """Two sorted canonical k-mer sets of about `size` keys, `shared_fraction` in common."""
    n_shared = int(size * shared_fraction)
    n_own = size - n_shared
    total = n_shared + 2 * n_own
    x = canonical(mix64(mix64(U(seed)) + np.arange(total, dtype=U)) >> U(64 - 2 * k), k)
    shared, a_own, b_own = x[:n_shared], x[n_shared:n_shared + n_own], x[n_shared + n_own:]
    return np.unique(np.concatenate([shared, a_own])), np.unique(np.concatenate([shared, b_own]))


def random_read_kmers(k, size, seed, canonical=True):
    """`size` distinct k-mers drawn the way lib/random.h:86-110 draws them."""
    seen = set()
    out = []
    ctr = 0
    mask = U((1 << (2 * k)) - 1)
    while len(out) < size:
        ctr += 1
        r = int(mix64(U((seed << 32) + ctr)))
        n_kmers = 1 + r % 100
        doubled = (r >> 20) & 1
        words = mix64(mix64(U((seed << 32) + ctr)) + np.arange(n_kmers, dtype=U)) & mask
        bases = np.zeros(n_kmers * k, dtype=np.uint8)
        for j in range(k):
            bases[j::k] = ((words >> U(2 * (k - 1 - j))) & U(3)).astype(np.uint8)
        if doubled:
            bases = np.concatenate([bases, bases])
        km = kmers_of_bases(bases, k)
        if canonical:
            km = np.minimum(km, revcomp(km, k))
        for v in km.tolist():
            if v not in seen:
                seen.add(v)
                out.append(v)
                if len(out) == size:
                    break
    return np.array(sorted(out), dtype=U)


def sample_bucket_ids(n_bits, seed, divisor=50):
    """First floor(2^N / 50) entries of a seeded Fisher-Yates over [0, 2^N), sorted."""
    n = 1 << n_bits
    m = n // divisor
    perm = list(range(n))
    for i in range(m):
        r = int(mix64(U((seed << 32) + i)))
        j = i + r % (n - i)
        perm[i], perm[j] = perm[j], perm[i]
    return np.array(sorted(perm[:m]), dtype=np.int32)


def to_bucketed(kmers, k, n_bits, key_bytes):
    """Sorted k-mers -> device layout: offsets int64[2^N + 1], keys (u32 or u64)."""
    kmers = np.asarray(kmers, dtype=U)
    key_bits = 2 * k - n_bits
    buckets = (kmers >> U(key_bits)).astype(np.int64)
    offsets = np.zeros((1 << n_bits) + 1, dtype=np.int64)
    np.cumsum(np.bincount(buckets, minlength=1 << n_bits), out=offsets[1:])
    keys = kmers & U((1 << key_bits) - 1)
    return offsets, keys.astype(np.uint16 if key_bytes <= 2 else (np.uint32 if key_bytes <= 4 else np.uint64))


def from_bucketed(offsets, keys, k, n_bits):
    """Inverse of to_bucketed."""
    key_bits = 2 * k - n_bits
    counts = np.diff(offsets)
    buckets = np.repeat(np.arange(counts.size, dtype=U), counts)
    return (buckets << U(key_bits)) | np.asarray(keys, dtype=U)


def circular_with_tails(k, length, n_tails, tail_len, seed, canonical_form=True):
    """Canonical k-mers (or, canonical_form=False, the k-mers as read) of a circular genome with
    short branches hanging off it.

    Loops are what the reference's serial passes exist for (spss.h:585-610 for
    non-branching loops of k-mers, :1578-1644 for loops in the path cover); tails
    turn one loop of k-mers into a loop of several unitigs.
    """
    g = random_genome(length, 0xC1AC0000 + seed)
    circ = np.concatenate([g, g[:k - 1]])
    parts = [kmers_of_bases(circ, k)]
    for t in range(n_tails):
        r = int(mix64(U((seed << 16) + t + 1)))
        p = r % length
        out_going = (r >> 32) & 1
        tail = random_genome(tail_len, 0x7A110000 + (seed << 8) + t)
        anchor = circ[p:p + k - 1]
        seq = np.concatenate([anchor, tail]) if out_going else np.concatenate([tail, anchor])
        parts.append(kmers_of_bases(seq, k))
    allk = np.concatenate(parts)
    return np.unique(canonical(allk, k) if canonical_form else allk)


def pack_strings(strings, k):
    """ACGT strings -> (uint64 words, uint32 len - K): the device SPSS container
    (kmer_set_compact.h:206-266 layout, see include/kmersets_hip.h)."""
    lens = np.array([len(x) - k for x in strings], dtype=np.int64)
    assert np.all(lens >= 0), "every SPSS string holds at least one k-mer"
    total = int(sum(len(x) for x in strings))
    if total == 0:
        return np.zeros(0, dtype=U), lens.astype(np.uint32)
    codes = bases_of_string("".join(strings)).astype(U)
    n_words = (total + 31) // 32
    padded = np.zeros(n_words * 32, dtype=U)
    padded[:total] = codes
    shifts = (U(62) - U(2) * (np.arange(32, dtype=U)))
    words = np.bitwise_or.reduce(padded.reshape(n_words, 32) << shifts[None, :], axis=1)
    return words.astype(U), lens.astype(np.uint32)


def unpack_strings(words, lens, k):
    words = np.asarray(words, dtype=U)
    lens = np.asarray(lens, dtype=np.int64) + k
    total = int(lens.sum())
    if total == 0:
        return ["" for _ in lens]
    shifts = (U(62) - U(2) * (np.arange(32, dtype=U)))
    codes = ((words[:, None] >> shifts[None, :]) & U(3)).reshape(-1)[:total].astype(np.uint8)
    text = string_of_bases(codes)
    out, at = [], 0
    for ln in lens:
        out.append(text[at:at + int(ln)])
        at += int(ln)
    return out
