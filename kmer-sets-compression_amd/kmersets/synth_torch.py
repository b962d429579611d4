"""Device-side twin of synth.py's phylogeny generator (plumbing for bench.py at sizes
where numpy on the host is too slow).  Same counter-based hashes, same arrays: tests
check it bit for bit against synth.phylogeny_sets at small sizes."""
import torch

from . import capi

_MASK64 = (1 << 64) - 1


def _c(x):
    """uint64 constant as the int64 with the same bit pattern."""
    x &= _MASK64
    return x - (1 << 64) if x >= (1 << 63) else x


_G, _M1, _M2 = _c(0x9E3779B97F4A7C15), _c(0xBF58476D1CE4E5B9), _c(0x94D049BB133111EB)


def lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x):
    x = x + _G
    x = (x ^ lsr(x, 30)) * _M1
    x = (x ^ lsr(x, 27)) * _M2
    return x ^ lsr(x, 31)


def _mix64_scalar(v):
    v = (v + 0x9E3779B97F4A7C15) & _MASK64
    v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & _MASK64
    return v ^ (v >> 31)


def random_genome(length, seed, device):
    idx = torch.arange(length, dtype=torch.int64, device=device)
    return mix64(idx + _c(_mix64_scalar(seed))) & 3


def mutate(bases, rate, edge_seed):
    idx = torch.arange(bases.numel(), dtype=torch.int64, device=bases.device)
    s = _mix64_scalar((edge_seed * 0xD6E8FEB86659FD93 + 1) & _MASK64)
    r = mix64(idx + _c(s))
    hit = lsr(r, 11).double() < rate * float(1 << 53)
    shift = 1 + (r & 0xFFFF) % 3
    return torch.where(hit, (bases + shift) & 3, bases)


def revcomp(x, k):
    x = ~x
    m2, m4 = _c(0x3333333333333333), _c(0x0F0F0F0F0F0F0F0F)
    x = (lsr(x, 2) & m2) | ((x & m2) << 2)
    x = (lsr(x, 4) & m4) | ((x & m4) << 4)
    m8, m16 = _c(0x00FF00FF00FF00FF), _c(0x0000FFFF0000FFFF)
    x = (lsr(x, 8) & m8) | ((x & m8) << 8)
    x = (lsr(x, 16) & m16) | ((x & m16) << 16)
    x = lsr(x, 32) | (x << 32)
    return lsr(x, 64 - 2 * k)


def canonical_set_of_bases(bases, k):
    n = bases.numel() - k + 1
    out = torch.zeros(n, dtype=torch.int64, device=bases.device)
    for j in range(k):
        out = (out << 2) | bases[j:j + n]
    out = torch.minimum(out, revcomp(out, k))        # k <= 31: both below 2^62, signed order is fine
    out, _ = torch.sort(out)
    return torch.unique_consecutive(out)


def plant_repeats(bases, n_segments, copies, seed, min_len=50, max_len=500):
    """synth.plant_repeats on the device: the same bases."""
    dev = bases.device
    n = bases.numel()
    seg = torch.arange(n_segments, dtype=torch.int64, device=dev)
    h0 = mix64(seg * 3 + _c(_mix64_scalar((seed * 0xA24BAED4963EE407 + 7) & _MASK64)))
    h1 = mix64(h0 + 1)
    u = lsr(h1, 11).double() / float(1 << 53)
    length = (min_len + (max_len - min_len) * u ** 3).to(torch.int64)
    span = max(1, n - max_len)

    def umod(x, m):          # x as an unsigned 64-bit number, modulo m (m < 2^62)
        return (lsr(x, 1) % m * 2 + (x & 1)) % m

    src = umod(h0, span)
    out = bases.clone()
    total = int(length.sum().item())
    within = torch.arange(total, dtype=torch.int64, device=dev)
    start = torch.cumsum(length, 0) - length
    frm = torch.repeat_interleave(src - start, length) + within
    for c in range(copies):
        dst = umod(mix64(h0 + _c((0x9E3779B97F4A7C15 * (c + 2)) & _MASK64)), span)
        at = torch.repeat_interleave(dst - start, length) + within
        # (numpy's out[at] = v keeps the LAST write to an index; index_put_ on the device keeps an arbitrary
        # one: write in ascending position order through a stable sort and keep each index's last writer)
        order = torch.argsort(at, stable=True)
        at_s, v_s = at[order], bases[frm][order]
        last = torch.ones_like(at_s, dtype=torch.bool)
        last[:-1] = at_s[1:] != at_s[:-1]
        out[at_s[last]] = v_s[last]
    return out


def phylogeny_sets(k, n_sets, size, seed, device, rate=0.002, repeats=None):
    """Same sets as synth.phylogeny_sets, as sorted int64 tensors on `device`."""
    genomes = [random_genome(size + k - 1, 0x5EED0000 + seed, device)]
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
    return [canonical_set_of_bases(g, k) for g in genomes[:n_sets]]


def genome_with_tips(k, size, seed, device, every=400):
    """Canonical k-mers of a random genome plus one-k-mer tips: the k-mer at every `every`-th position
    gets a second successor ending in T where the genome continues with A, C or G.  The greedy path
    cover (spss.h:1445-1499) tries a node's right edges in base order, so the genome's own edge wins
    and the unitigs between tips are stitched into one long string (same sets as synth.genome_with_tips)."""
    bases = random_genome(size + k - 1, 0x5EED0000 + seed, device)
    n = bases.numel() - k + 1
    fwd = torch.zeros(n, dtype=torch.int64, device=device)
    for j in range(k):
        fwd = (fwd << 2) | bases[j:j + n]
    pos = torch.arange(0, n - 1, every, dtype=torch.int64, device=device)
    keep = bases[pos + k] != 3
    mask = (1 << (2 * k)) - 1
    tips = ((fwd[pos[keep]] << 2) & mask) | 3
    allk = torch.cat([fwd, tips])
    allk = torch.minimum(allk, revcomp(allk, k))
    allk, _ = torch.sort(allk)
    return torch.unique_consecutive(allk)


def device_set(g, kmers):
    """Sorted int64 k-mers on device -> capi.DeviceSet (bucketed keys), no host round trip."""
    nb = 1 << g.n_bucket_bits
    key_bits = 2 * g.k - g.n_bucket_bits
    buckets = kmers >> key_bits
    counts = torch.bincount(buckets, minlength=nb)
    offsets = torch.zeros(nb + 1, dtype=torch.int64, device=kmers.device)
    torch.cumsum(counts, 0, out=offsets[1:])
    keys = kmers & ((1 << key_bits) - 1)
    if g.key_bytes == 2:
        raw = keys.to(torch.int16).contiguous().view(torch.uint8)   # (the low 16 bits, as a bit pattern)
    elif g.key_bytes == 4:
        raw = keys.to(torch.int32).contiguous().view(torch.uint8)
    else:
        raw = keys.contiguous().view(torch.uint8)
    if raw.numel() < 16:
        raw = torch.cat([raw, torch.zeros(16 - raw.numel(), dtype=torch.uint8, device=kmers.device)])
    return capi.DeviceSet(g, offsets, raw, kmers.numel())
