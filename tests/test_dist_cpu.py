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


def _owned_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmersets import owned_model

        k, n, kb = 15, 14, 2
        key_bits = 2 * k - n
        sets = synth.phylogeny_sets(k, 8, 6000, seed=21)
        ids = synth.sample_bucket_ids(n, seed=22)
        # striped owners: siblings on different ranks, so most merges pull a set across
        owners = [i % world for i in range(len(sets))]
        mine = [s if owners[i] == rank else None for i, s in enumerate(sets)]

        def spss_weight(kmers):
            return ol.Set.from_kmers(k, n, kb, kmers).compact().weight()

        res = owned_model.build_owned(mine, owners, key_bits, ids, spss_weight, dist)
        okss = ol.KmerSetSet([ol.Set.from_kmers(k, n, kb, s).compact() for s in sets], ids)
        assert [tuple(int(x) for x in r) for r in okss.iterations()] == [tuple(r) for r in res["trace"]]
        ocp, _ = okss.checkpoints()
        assert [tuple(int(x) for x in r) for r in ocp] == [(a, b, c, int(d)) for a, b, c, d in res["checkpoints"]]
        assert len(res["owner"]) == okss.size()
        held = 0
        for i in range(okss.size()):
            if res["owner"][i] == rank:
                assert np.array_equal(res["sets"][i], okss.node(i).to_set().kmers()), i
                held += 1
            else:
                assert res["sets"][i] is None
        tot = torch.tensor([held, res["sets_received"]], dtype=torch.int64)
        dist.all_reduce(tot)
        assert int(tot[0]) == okss.size() and int(tot[1]) > 0
        with open(os.path.join(out_dir, "owned_%d.txt" % rank), "w") as f:
            f.write("ok %d" % len(res["trace"]))
    finally:
        dist.destroy_process_group()


def test_owned_schedule_world2_gloo(tmp_path):
    """The owner-sharded build's schedule (kmersets/owned_model.py mirrors csrc/ksh_kss.hip build_owned)
    with world_size 2 over gloo: control loop on the all-gathered samples only, merges on the owner of
    j with k sent across, SPSS weights exchanged at the checks -- trace, checkpoints and every node's
    set equal the oracle's single-process KmerSetSet, every node lives on exactly one rank."""
    port = 29600 + os.getpid() % 200
    mp.spawn(_owned_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(os.path.join(str(tmp_path), "owned_%d.txt" % r)).read().startswith("ok")


def test_multi_worker_oracle_matches_single_worker():
    """oracle/ko_mt.h (the reference's n_workers > 1 branches, bench.py's cpu_baseline) against the
    n_workers == 1 oracle: merge sequence, N_proc, every node's set, Get(i)."""
    k, n, kb = 23, 14, 4
    sets = synth.phylogeny_sets(k, 6, 30000, seed=31)
    ids = synth.sample_bucket_ids(n, seed=32)
    oc = [ol.Set.from_kmers(k, n, kb, s).compact() for s in sets]
    one = ol.KmerSetSet(oc, ids, max_iterations=4)
    many = ol.KmerSetSet(oc, ids, max_iterations=4, n_workers=4)
    assert np.array_equal(one.iterations(), many.iterations()) and one.stat(3) == many.stat(3)
    assert one.size() == many.size() and one.meta() == many.meta()
    for i in range(one.size()):
        assert np.array_equal(one.node(i).to_set().kmers(), many.node(i).to_set().kmers())
        strings = many.node(i).strings()          # an SPSS of the same set: every k-mer once
        assert sum(len(x) - k + 1 for x in strings) == one.node(i).size()
    for i in range(len(sets)):
        assert np.array_equal(many.get(i).kmers(), sets[i])
