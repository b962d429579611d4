#!/usr/bin/env python3
"""bench_loop.py -- the whole KmerSetSet loop on one GPU (BASELINE configs[2] shape).

The same timed region as the driver's contract bench (bench.py, which runs the 64 x 10^8 case), with
knobs for other shapes: what kmerset-multiple-compress times around the KmerSetSet constructor
(src/kmerset-multiple-compress.cc:96-101): inputs are KmerSetCompact containers already resident
in HBM, the timed region is ksh_kss_build (decode of the inputs, weight table, every merge
iteration, and the SPSS encodes of the nodes that are stale when the loop reads weights: the
three re-encodes of a merge are deferred to the convergence checks and the end).
Mk-mers/s = N_proc / wall with N_proc as SURVEY.md 8(d) defines it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "kmer-sets-compression_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=23)
    ap.add_argument("--bucket-bits", type=int, default=14)
    ap.add_argument("--sets", type=int, default=16)
    ap.add_argument("--size", type=float, default=1e7)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--max-iterations", type=int, default=-1)
    ap.add_argument("--lanes", type=int, default=0, help="ksh_ctx_set_lanes (0 = default, 1 = one stream)")
    ap.add_argument("--repeats", default="",
                    help="SEGMENTS,COPIES: a repeat-rich family -- the ancestor genome carries SEGMENTS stretches of "
                         "50..500 bases copied to COPIES other places each (synth.plant_repeats): branching unitig "
                         "graphs, deep matchings, loop cuts (lib/core/spss.h:1445-1644) instead of a few long unitigs")
    ap.add_argument("--rate", type=float, default=0.002, help="substitutions per base and tree edge")
    ap.add_argument("--cpu-iterations", type=int, default=0,
                    help="also time the oracle on the first I iterations (0 = skip)")
    ap.add_argument("--cpu-size", type=float, default=0, help="set size for the CPU leg (default: --size)")
    ap.add_argument("--warmup-builds", type=int, default=1,
                    help="untimed builds before the timed one: the first build of a process pays the "
                         "driver's one-off cost of mapping fresh VRAM (10-40 us per MB on this pool, box by "
                         "box), which the context's caching pool then keeps")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed build: Size and XOR Hash of Get(i) against the decoded input, every i "
                         "(the reference's --check, src/kmerset-multiple-compress.cc:104-126), on the device")
    ap.add_argument("--gpus", type=int, default=1,
                    help="strong scaling of the loop: N processes (python -m torch.distributed.run "
                         "--nproc-per-node N bench_loop.py --gpus N ...), every rank runs the loop on "
                         "replicated sets and the SPSS encodes are dealt out (ksh_kss_build_sharded)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from kmersets import capi, synth, synth_torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d (launch with torch.distributed.run)" % (world, args.gpus))
    dist, coll_dev = None, "cpu"
    backend = os.environ.get("KSH_BENCH_BACKEND", "nccl")   # gloo: rehearsal, every rank on cuda:0
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            coll_dev = torch.device("cuda", local_rank)
        else:
            dist.init_process_group(backend)

    k, nbits, n_sets, size = args.k, args.bucket_bits, args.sets, int(args.size)
    g = capi.geom(k, nbits)
    ctx = capi.Context(local_rank)
    ctx.set_lanes(args.lanes)
    dev = ctx.device
    t0 = time.perf_counter()
    repeats = tuple(int(x) for x in args.repeats.split(",")) if args.repeats else None
    kmers = synth_torch.phylogeny_sets(k, n_sets, size, args.seed, dev, rate=args.rate, repeats=repeats)
    compacts = []
    graph = None
    for km in kmers:
        compacts.append(ctx.spss_encode(synth_torch.device_set(g, km), mode=0))
        if graph is None:   # the shape of the first input's unitig graph
            es = ctx.spss_encode_stats()
            graph = {"kmers": int(km.numel()), "unitigs": es["unitigs"], "strings": es["strings"],
                     "matching_rounds": es["rounds"], "routes": sorted(ctx.spss_encode_routes())}
    sizes = [int(km.numel()) for km in kmers]
    del kmers
    torch.cuda.synchronize()
    t_inputs = time.perf_counter() - t0
    ids = synth.sample_bucket_ids(nbits, seed=args.seed + 1)

    first_wall = None
    for _ in range(max(0, args.warmup_builds)):
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        warm = capi.DeviceKmerSetSet(ctx, compacts, ids, max_iterations=args.max_iterations, dist=dist,
                                     dist_device=coll_dev)
        torch.cuda.synchronize()
        if first_wall is None:
            first_wall = time.perf_counter() - w0
        warm.close()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kss = capi.DeviceKmerSetSet(ctx, compacts, ids, max_iterations=args.max_iterations, dist=dist,
                                dist_device=coll_dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    st = kss.stats()
    it, cp, imp = kss.trace()
    encodes = [st["n_encodes"]]
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # the StreamVByte sizes of a node's lengths are known where the node's SPSS is held
        lb = torch.tensor([st["length_bytes"]], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(lb)
        st["length_bytes"] = int(lb.item())
        ne = torch.tensor([st["n_encodes"]], dtype=torch.int64, device=coll_dev)
        parts = [torch.zeros_like(ne) for _ in range(world)]
        dist.all_gather(parts, ne)
        encodes = [int(x.item()) for x in parts]
    out = {
        "metric": "Mk-mers/s processed in kmerset-multiple-compress (KmerSetSet constructor)",
        "value": st["n_processed"] / wall / 1e6,
        "unit": "Mk-mers/s",
        "wall_s": wall,
        "n_processed": st["n_processed"],
        "iterations": int(it.shape[0]),
        "nodes": st["nodes"],
        "sum_input_kmers": sum(sizes),
        "chars_per_kmer_before": st["initial_spss_weight"] / sum(sizes),
        "chars_per_kmer_after": st["final_spss_weight"] / sum(sizes),
        "bytes_per_kmer_after_spss": (st["packed_bytes"] + st["length_bytes"]) / sum(sizes),
        "n_gpus": world,
        "scaling": "strong",
        "encodes_per_rank": encodes,
        "phase_seconds": st["phase_seconds"],
        "encodes": {"n": st["n_encodes"], "kmers": st["n_encoded_kmers"],
                    "ns_per_kmer": st["phase_seconds"]["encodes"] * 1e9 / max(st["n_encoded_kmers"], 1)},
        "first_input_graph": graph,
        "family": "phylogeny, rate %g per base and edge%s" % (args.rate, (", planted repeats: %d segments x %d copies"
                                                                         % repeats) if repeats else ""),
        "first_build_wall_s": first_wall,
        "config": {"workload": "%d canonical k=%d sets of %d k-mers, full KmerSetSet loop" % (n_sets, k, size),
                   "input_build_s": t_inputs,
                   "parallelism": "1 GPU" if world == 1 else
                                  "1 process per GPU, sets replicated, SPSS encodes dealt out by node, "
                                  "all-gather of (n_strings, n_bases) at the convergence checks"},
    }
    if args.verify:
        bad = []
        for i, c in enumerate(compacts):
            want = ctx.spss_decode(c)
            got = kss.get_size_and_hash(i)
            if got != (want.n_keys, ctx.set_hash(want)):
                bad.append(i)
            del want
        out["verified"] = {"sets": len(compacts), "mismatches": bad,
                           "method": "Size and XOR Hash of Get(i) == those of the decoded input i, on the device"}
        if bad:
            raise SystemExit("verification failed for sets %s" % bad)
    if args.cpu_iterations > 0 and rank == 0:
        import oracle_lib as ol

        csize = int(args.cpu_size) if args.cpu_size else size
        host = synth.phylogeny_sets(k, n_sets, csize, seed=args.seed)
        oc = [ol.Set.from_kmers(k, nbits, g.key_bytes, s).compact() for s in host]
        c0 = time.perf_counter()
        okss = ol.KmerSetSet(oc, ids, max_iterations=args.cpu_iterations)
        cw = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": okss.stat(3) / cw / 1e6, "unit": "Mk-mers/s", "cores": 1,
                               "kind": "port",
                               "sample": "oracle KmerSetSet, %d sets of %d k-mers, first %d iterations, %.1f s"
                                         % (n_sets, csize, args.cpu_iterations, cw)}
    if rank == 0:
        print(json.dumps(out))
    kss.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
