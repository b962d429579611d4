"""BASELINE.json's configurations at their FULL sizes, through size-independent properties (the oracle
finishes such sizes in hours, so it checks the same code at small sizes in the other files):

* configs[2]: 16 canonical k=23 sets of 10^8 k-mers, the whole KmerSetSet loop on one GPU;
* configs[3]'s workload: 64 canonical k=23 sets of 10^8 k-mers (what bench.py times), one build;
* the same at k=31 (64-bit keys), 8 sets of 5 * 10^8: configs[4]'s geometry at what one GPU's share is.

Checked: Size and XOR Hash of Get(i) == those of the decoded input for every i (the reference's own
--check, src/kmerset-multiple-compress.cc:104-126); DAG invariants (every merge adds one node and two
child links, children have larger indices, the reachable nodes of an input are pairwise disjoint and
their sizes add up); the trace's sizes are consistent with the nodes; total SPSS weight never grows at
a checkpoint that continues; two builds of the same inputs agree in every number (bytes/k-mer included).
"""
import numpy as np
import pytest

from kmersets import capi, synth, synth_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    c = capi.Context(0)
    yield c
    c.close()


def _loop_properties(ctx, k, n_sets, size, seed, builds=2):
    import torch

    g = capi.geom(k, 14)
    kmers = synth_torch.phylogeny_sets(k, n_sets, size, seed, ctx.device)
    sizes = [int(x.numel()) for x in kmers]
    compacts = []
    for i in range(n_sets):
        compacts.append(ctx.spss_encode(synth_torch.device_set(g, kmers[i]), mode=0))
        kmers[i] = None
    del kmers
    torch.cuda.empty_cache()
    ids = synth.sample_bucket_ids(14, seed=seed + 1)
    results = []
    for _ in range(builds):
        kss = capi.DeviceKmerSetSet(ctx, compacts, ids)
        st = kss.stats()
        it, cp, imp = kss.trace()
        n_nodes = kss.size()
        # ---- DAG
        assert n_nodes == n_sets + len(it) and st["nodes"] == n_nodes
        children = [kss.children(i) for i in range(n_nodes)]
        assert sum(len(c) for c in children) == 2 * len(it)
        for t, (j, kk, weight, original, diff) in enumerate(it.tolist()):
            new = n_sets + t
            assert j < kk < new and weight > 0
            assert new in children[j] and new in children[kk]
        assert all(c > i for i, cs in enumerate(children) for c in cs)
        node_sizes = [kss.node_size(i) for i in range(n_nodes)]
        assert sum(node_sizes) == st["final_total_size"]
        assert st["initial_total_size"] == sum(sizes)
        assert st["initial_total_size"] + int(it[:, 4].sum()) == st["final_total_size"]
        assert st["n_processed"] == sum(sizes) + int(it[:, 3].sum())
        # ---- Get(i) == input i (Size, XOR Hash), and the reachable nodes add up
        for i in range(n_sets):
            want = ctx.spss_decode(compacts[i])
            assert want.n_keys == sizes[i]
            assert kss.get_size_and_hash(i) == (want.n_keys, ctx.set_hash(want)), i
            del want
            seen, todo = set(), [i]
            while todo:
                cur = todo.pop()
                if cur not in seen:
                    seen.add(cur)
                    todo.extend(children[cur])
            assert sum(node_sizes[x] for x in seen) == sizes[i]
        # ---- checkpoints: a checkpoint that lets the loop go on saw the weight shrink
        for (iteration, previous, updated, stopped), improvement in zip(cp.tolist(), imp.tolist()):
            assert stopped or updated < previous
            assert abs(improvement - (previous - updated) / previous) < 1e-6
        assert st["final_spss_weight"] <= st["initial_spss_weight"]
        results.append((st["n_processed"], st["final_spss_weight"], st["packed_bytes"], st["length_bytes"], n_nodes,
                        it.tolist(), cp.tolist()))
        kss.close()
    assert all(r == results[0] for r in results)          # deterministic: bytes/k-mer equal across builds
    return results[0]


def test_config3_16x1e8_k23(ctx):
    n_proc, weight, packed, lens, nodes, it, cp = _loop_properties(ctx, 23, 16, int(1e8), seed=3)
    assert len(it) >= 8 and nodes > 16
    assert (packed + lens) / (16 * 1e8) < 0.26          # below the 2 bits per k-mer of the unmerged sets


def test_config4_64x1e8_k23(ctx):
    """configs[3]'s sets (the workload of bench.py's headline number) on one GPU: one whole build and
    the same checks -- every Get(i) against its input by Size and XOR Hash, the DAG, the trace's sizes."""
    n_proc, weight, packed, lens, nodes, it, cp = _loop_properties(ctx, 23, 64, int(1e8), seed=3, builds=1)
    assert len(it) >= 32 and nodes > 64
    assert len(cp) >= 1 and n_proc > 64 * 0.9e8
    assert (packed + lens) / (64 * 1e8) < 0.20


def test_k31_8x5e8(ctx):
    """configs[4]'s geometry (k = 31, 8-byte keys, 5 x 10^8 k-mers per set) at eight sets -- a quarter of one
    rank's 32-set share of the 256 (the full share, 128 GB of resident keys, is the owner-sharded build's; with
    its encode scratch and the 20 GB of samples it has no room for a second build's warm pool on one 288 GB GPU,
    DESIGN.md 7.2) -- and the memory rows of that table checked against the context's counters: the pooled
    buffers (sets, containers, a merge's three results) and the encode scratch per lane."""
    before = ctx.mem_stats(reset_peak=True)
    n_proc, weight, packed, lens, nodes, it, cp = _loop_properties(ctx, 31, 8, int(5e8), seed=5)
    assert len(it) >= 4 and nodes > 8
    m = ctx.mem_stats()
    n, key = 5e8, 8
    sets_bytes = 8 * n * key                         # the resident inputs: 32 GB
    # peak of the pooled buffers: the resident sets, the three results of the largest merge beside its two inputs
    # (<= 3 more sets), the nodes' SPSS containers (< 0.3 B per k-mer), the decode's intermediate keys (10 B per
    # k-mer of one input per lane) -- with the pool's <= 12.5 % rounding
    bound = (sets_bytes + 3 * n * key + 0.3 * 8 * n + 3 * 10 * n) * 1.125 + before["pool_live"]
    assert m["pool_peak"] <= bound, (m, bound)
    assert m["pool_peak"] >= sets_bytes              # (the counter does count: the inputs alone are 32 GB)
    # scratch of one context: the encode slot (~70 B per k-mer + the fine index, with slot_reserve's
    # 12.5 % to spare: 40 GB at 5 x 10^8), the decode slot, the arena and the pair plan -- measured 42.7 GB; a helper
    # lane holds the same plus its own pool (the decode's intermediate keys, an encode's unitig block): 48 GB
    assert m["scratch"] <= 92 * n, m
    # ... and the lanes' scratch together is at most two fifths of the device (run_on_lanes: three encode slots of
    # 39.5 GB at this size); beside it every lane's own pool keeps the decode's temporaries (10 B per k-mer of
    # intermediate keys and bucket ids, 8 B per k-mer of the oversize buckets' scratch copy, the pool's rounding)
    assert m["lanes"] <= 0.4 * 309e9 + m["n_lanes"] * 21 * n and m["n_lanes"] <= 4, m
    assert m["scratch"] >= 70 * n   # (the counters do count)
