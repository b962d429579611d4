"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  The pair graph shards with no data-path
collective: every rank computes the sizes / weights of its share of the pairs and one
all-gather of int64 values puts the whole table on every rank (KB-scale, latency-bound;
it replaces the mutex-guarded `weights[...] = w` updates of
lib/core/kmer_set_set.h:205-218,409-420)."""
import numpy as np


def split_range(begin, end, n):
    """Range::Split (lib/core/range.h:52-77): n chunks, the first n - r of size floor(s / n),
    then r of size floor(s / n) + 1."""
    size = end - begin
    small = size // n
    large_n = size - small * n
    out, at = [], begin
    for i in range(n):
        ln = small if i < n - large_n else small + 1
        out.append((at, at + ln))
        at += ln
    return out


def shard_pairs(pairs, rank, world):
    """The contiguous block of `pairs` this rank owns."""
    lo, hi = split_range(0, len(pairs), world)[rank]
    return pairs[lo:hi], lo, hi


def all_gather_table(local_values, n_total, lo, dist, device="cpu"):
    """Every rank contributes values for pairs [lo, lo + len(local_values)); returns the full
    int64 table of n_total values on every rank (padded all-gather, since shards differ by
    at most one pair)."""
    import torch

    world = dist.get_world_size()
    per = (n_total + world - 1) // world + 1
    buf = torch.full((per + 2,), -1, dtype=torch.int64, device=device)
    buf[0] = lo
    buf[1] = len(local_values)
    if len(local_values):
        buf[2:2 + len(local_values)] = torch.as_tensor(np.asarray(local_values, dtype=np.int64), device=device)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    table = np.full(n_total, -1, dtype=np.int64)
    for g in gathered:
        g = g.cpu().numpy()
        start, cnt = int(g[0]), int(g[1])
        table[start:start + cnt] = g[2:2 + cnt]
    assert np.all(table >= 0), "every pair is owned by exactly one rank"
    return table


def arg_max_pair(pairs, weights):
    """The reference's arg-max (kmer_set_set.h:308-316, strict >) over an ordered table:
    the first maximal pair in lexicographic order; None when every weight is 0."""
    best, best_w = None, 0
    for p, w in sorted(zip(pairs, weights)):
        if w > best_w:
            best, best_w = p, int(w)
    return best, best_w
