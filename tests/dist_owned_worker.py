"""Worker of tests/test_gpu_kmer_set_set.py::test_owned_build: one of N ranks (all on GPU 0; gloo
through host memory is the transport, because RCCL refuses several ranks on one device) building
one KmerSetSet with ksh_kss_build_owned.  KSH_OWNED_BACKEND=nccl (::test_owned_build_rccl_two_gpus, on a
box with two GPUs or more): one GPU per rank and the library's own RCCL transport on device buffers --
ncclSend / ncclRecv pairs, the side communicator, the deferred all-gathers.  Every input is decoded by its owner only; every rank
checks the replicated state (trace, checkpoints, DAG, sizes) against the oracle, and the rank that
owns a node checks its set and its SPSS strings; nobody else can read them."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "kmer-sets-compression_amd"))
import oracle_lib as ol  # noqa: E402
from kmersets import capi, synth  # noqa: E402


def main():
    k, n, kb, n_sets, size, seed = (int(x) for x in sys.argv[1:7])
    layout = sys.argv[7] if len(sys.argv) > 7 else "block"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("KSH_OWNED_BACKEND", "gloo")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        coll_dev = torch.device("cuda", local)
    else:
        local = 0
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
        coll_dev = "cpu"
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = capi.Context(local)
    sets = synth.phylogeny_sets(k, n_sets, size, seed=seed)
    osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
    ocompacts = [s.compact() for s in osets]
    ids = synth.sample_bucket_ids(n, seed=seed + 1)
    okss = ol.KmerSetSet(ocompacts, ids)
    g = capi.geom(k, n)
    # "block": neighbours share a rank (most merges are local); "striped": siblings live on
    # different ranks, so nearly every merge pulls a set across
    owners = capi.block_owners(n_sets, world) if layout == "block" else [i % world for i in range(n_sets)]
    dcompacts = [capi.DeviceSpss.from_strings(g, c.strings(), ctx.device) if owners[i] == rank else None
                 for i, c in enumerate(ocompacts)]
    dkss = capi.OwnedKmerSetSet(ctx, dcompacts, ids, dist, coll_dev, owners=owners)
    transport = dkss.comm.kind
    ranks_seen = dkss.comm.ranks_seen()
    assert ranks_seen == world, (ranks_seen, world)

    it, cp, imp = dkss.trace()
    assert np.array_equal(it, okss.iterations()), (it, okss.iterations())
    ocp, oimp = okss.checkpoints()
    assert np.array_equal(cp, ocp) and np.array_equal(imp, oimp)
    assert dkss.size() == okss.size() and dkss.meta() == okss.meta()
    st = dkss.stats()
    assert st["initial_spss_weight"] == okss.stat(2) and st["n_processed"] == okss.stat(3)
    assert st["initial_total_size"] == okss.stat(0) and st["final_total_size"] == okss.stat(1)
    assert st["final_spss_weight"] == sum(okss.node(i).weight() for i in range(okss.size()))
    assert np.array_equal(dkss.initial_weights(), okss.initial_weights(n_sets))
    held = []
    for i in range(okss.size()):
        node = okss.node(i)
        assert dkss.node_size(i) == node.size()
        holder = dkss.node_holder(i)
        assert 0 <= holder < world
        if holder == rank:
            assert np.array_equal(dkss.node_kmers(i), node.to_set().kmers()), "node %d" % i
            assert dkss.node_strings(i) == node.strings(), "node %d" % i
            held.append(i)
        else:
            for reader in (dkss.node_strings, dkss.node_kmers):    # lives elsewhere: not readable here
                try:
                    reader(i)
                    raise AssertionError("node %d should live on rank %d only" % (i, holder))
                except capi.KshError as e:
                    assert e.code == 9
    # Get(i) through the gathered (size, hash) table == the input, for every input, on every rank
    for i in range(n_sets):
        assert dkss.get_size_and_hash(i) == (osets[i].size(), osets[i].hash()), i
    mine = torch.zeros(okss.size(), dtype=torch.int64, device=coll_dev)
    mine[held] = 1
    dist.all_reduce(mine)
    assert int(mine.min()) == 1 and int(mine.max()) == 1      # every node lives on exactly one rank
    cs = dkss.comm_stats()
    vec = torch.tensor([dkss.stats()["n_encodes"], cs["p2p_sets"], cs["p2p_bytes_sent"], cs["p2p_bytes_received"],
                        len(held), cs["checks_deferred"], cs["rollbacks"], cs["sets_migrated"]], dtype=torch.int64,
                       device=coll_dev)
    allv = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(allv, vec)
    if rank == 0:
        print(json.dumps({"ok": True, "transport": transport, "ranks_seen": ranks_seen,
                          "iterations": int(len(it)), "nodes": okss.size(),
                          "encodes_per_rank": [int(v[0]) for v in allv],
                          "sets_sent_per_rank": [int(v[1]) for v in allv],
                          "bytes_sent": sum(int(v[2]) for v in allv), "bytes_received": sum(int(v[3]) for v in allv),
                          "nodes_per_rank": [int(v[4]) for v in allv],
                          "checks": int(len(cp)), "checks_deferred": int(allv[0][5]), "rollbacks": int(allv[0][6]),
                          "sets_migrated": sum(int(v[7]) for v in allv)}))
    dkss.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
