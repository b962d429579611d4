"""world_size-2 gloo test of the N > 1 path (CPU): the pair list shards across ranks, each
rank computes its pairs' sizes, one all-gather rebuilds the full table on every rank and
both ranks pick the same arg-max pair as a serial run.  The per-pair arithmetic here is the
oracle's (no GPU in this container); on the GPU box the same plumbing carries the HIP
results (bench.py --gpus N)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from kmersets import dist as kdist
from kmersets import synth


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, n, kb = 15, 14, 2
        sets = synth.phylogeny_sets(k, 5, 8000, seed=13)
        osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
        pairs = [(i, j) for i in range(len(sets)) for j in range(i + 1, len(sets))]
        mine, lo, hi = kdist.shard_pairs(pairs, rank, world)
        local = [osets[i].intersection(osets[j]).size() for (i, j) in mine]
        table = kdist.all_gather_table(local, len(pairs), lo, dist)
        best = kdist.arg_max_pair(pairs, table)
        np.save(os.path.join(out_dir, "table_%d.npy" % rank), table)
        with open(os.path.join(out_dir, "best_%d.txt" % rank), "w") as f:
            f.write(repr(best))
        # weak-scaling shape of bench.py: every rank has its own batch; sizes are gathered
        t = torch.tensor([rank * 100 + x for x in range(6)], dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        assert [int(g[0]) for g in got] == [r * 100 for r in range(world)]
        tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        assert float(tmax) == float(world)
    finally:
        dist.destroy_process_group()


def test_pair_sharding_all_gather(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    k, n, kb = 15, 14, 2
    sets = synth.phylogeny_sets(k, 5, 8000, seed=13)
    osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
    pairs = [(i, j) for i in range(len(sets)) for j in range(i + 1, len(sets))]
    want = np.array([osets[i].intersection(osets[j]).size() for (i, j) in pairs], dtype=np.int64)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("table_%d.npy" % r))), want)
        assert open(str(tmp_path / ("best_%d.txt" % r))).read() == repr(kdist.arg_max_pair(pairs, want))
    assert kdist.arg_max_pair(pairs, want)[1] == int(want.max())


def test_split_range_matches_reference_rule():
    """lib/core/range.h:52-77 (test/range.cc): contiguous, sizes differ by at most one,
    smaller chunks first."""
    for begin in range(0, 5):
        for end in range(begin + 1, 40, 3):
            for n in range(1, 12):
                chunks = kdist.split_range(begin, end, n)
                assert chunks[0][0] == begin and chunks[-1][1] == end
                assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
                sizes = [b - a for a, b in chunks]
                assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes)
