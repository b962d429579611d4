#!/usr/bin/env python3
"""bench_loop.py -- the whole KmerSetSet loop on one GPU (BASELINE configs[2] shape).

Not the driver's contract bench (that is bench.py, configs[1]); this one times what
kmerset-multiple-compress times around the KmerSetSet constructor
(src/kmerset-multiple-compress.cc:96-101): inputs are KmerSetCompact containers
already resident in HBM, the timed region is ksh_kss_build (decode of the inputs,
weight table, every merge iteration with its three re-encodes).
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
    ap.add_argument("--cpu-iterations", type=int, default=0,
                    help="also time the oracle on the first I iterations (0 = skip)")
    ap.add_argument("--cpu-size", type=float, default=0, help="set size for the CPU leg (default: --size)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from kmersets import capi, synth, synth_torch

    k, nbits, n_sets, size = args.k, args.bucket_bits, args.sets, int(args.size)
    g = capi.geom(k, nbits)
    ctx = capi.Context(0)
    dev = ctx.device
    t0 = time.perf_counter()
    kmers = synth_torch.phylogeny_sets(k, n_sets, size, args.seed, dev)
    compacts = []
    for km in kmers:
        compacts.append(ctx.spss_encode(synth_torch.device_set(g, km), mode=0))
    sizes = [int(km.numel()) for km in kmers]
    del kmers
    torch.cuda.synchronize()
    t_inputs = time.perf_counter() - t0
    ids = synth.sample_bucket_ids(nbits, seed=args.seed + 1)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kss = capi.DeviceKmerSetSet(ctx, compacts, ids, max_iterations=args.max_iterations)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = kss.stats()
    it, cp, imp = kss.trace()
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
        "config": {"workload": "%d canonical k=%d sets of %d k-mers, full KmerSetSet loop" % (n_sets, k, size),
                   "input_build_s": t_inputs},
    }
    if args.cpu_iterations > 0:
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
    print(json.dumps(out))
    kss.close()
    ctx.close()


if __name__ == "__main__":
    main()
