#!/usr/bin/env python3
"""bench.py -- Mk-mers/s of the k-mer-set hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[1]): 4 synthetic canonical k=23 sets of
10^7 k-mers each (seeded phylogeny family), resident in HBM as
bucketed sorted keys; one step = the pair algebra of all 6 pairs (one
ksh_pair_algebra_batch call: the pairs are independent, so their passes are enqueued back to
back and one stream synchronisation returns all the sizes) -- for each pair
(A, B): A&B, A\\B, B\\A with their bucket offsets and the three counts (what one
KmerSetSet iteration asks for, lib/core/kmer_set_set.h:339-343), Diff = |A\\B| +
|B\\A| derived from them.  Units = sum over pairs of (|A| + |B|) k-mers.

N > 1 (one process per GPU, launched by torch.distributed.run): every rank runs
the same-size batch on its own family (weak scaling), then the per-pair diff
sizes are all-gathered over RCCL (the exchange step north_star names).

Prints ONE JSON line on rank 0.  `roofline` is the write-pass merge kernel
(k_tile_merge<KeyT, 1>): algorithmic bytes (|A| + |B| + |A u B|) * key_bytes per
launch over its HIP-event duration, against the 8 TB/s HBM peak; `traffic` is filled
from profiles/pmc_traffic.json when that file holds a PMC measurement (rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE passes) of the same kernel on the same workload.
`cpu_baseline` times the oracle's restatement of the reference's hash-set algebra on
the host (rank 0, N = 1 only, bounded sample).  `spss` (outside the timed region) runs
the whole KmerSetSet loop once on the same sets and reports the second half of the
metric: bytes/k-mer after SPSS.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--k", type=int, default=23)
    ap.add_argument("--bucket-bits", type=int, default=14)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--size", type=float, default=1e7, help="k-mers per set")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spss", action="store_true", help="skip the untimed KmerSetSet run")
    ap.add_argument("--two-pass", action="store_true",
                    help="use ksh_pair_plan + allocate + ksh_pair_write (exact-size outputs) "
                         "instead of the one-call ksh_pair_algebra (upper-bound outputs)")
    ap.add_argument("--per-pair-sync", action="store_true",
                    help="one ksh_pair_algebra call (and one stream sync) per pair instead of one "
                         "ksh_pair_algebra_batch call per step")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--time-every", type=int, default=5,
                    help="HIP-event pair around every n-th merge launch of the timed region (an event "
                         "pair idles the stream for ~10 us, so timing every launch would cost the "
                         "throughput figure ~15%%)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with "
                             "python -m torch.distributed.run --nproc-per-node %d ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # Rehearsal on a one-GPU box: KSH_BENCH_BACKEND=gloo puts every rank on cuda:0 and runs
    # the collectives on CPU tensors (RCCL refuses two ranks on one device).
    backend = os.environ.get("KSH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    from kmersets import capi, synth

    k, nbits, n_sets, size = args.k, args.bucket_bits, args.sets, int(args.size)
    g = capi.geom(k, nbits)
    ctx = capi.Context(local_rank)
    dev = ctx.device

    # ---- inputs, resident in HBM before anything is timed
    host_sets = synth.phylogeny_sets(k, n_sets, size, seed=args.seed + 1000 * rank)
    sets = [capi.DeviceSet.from_kmers(g, s, dev) for s in host_sets]
    pairs = [(i, j) for i in range(n_sets) for j in range(i + 1, n_sets)]
    units_per_step = sum(sets[i].n_keys + sets[j].n_keys for i, j in pairs)

    # two slots: the all-gather of step s travels while step s + 1 computes
    diff_local = [torch.zeros(len(pairs), dtype=torch.int64, device=coll_dev) for _ in range(2)]
    # pinned staging for the per-step sizes: the copy to the device is enqueued, not waited for (a
    # slot is reused two steps later, after a batch call that ends with a stream synchronisation)
    diff_host = [torch.zeros(len(pairs), dtype=torch.int64) for _ in range(2)]
    if world > 1 and coll_dev.type == "cuda":
        diff_host = [t.pin_memory() for t in diff_host]
    diff_host_np = [t.numpy() for t in diff_host]
    gathered = [[torch.zeros_like(diff_local[0]) for _ in range(world)] for _ in range(2)] if world > 1 else None
    pending = [None]
    step_no = [0]
    algo_bytes = [0.0]

    algebra = ctx.pair_algebra if args.two_pass else ctx.pair_algebra_onepass
    kernel_kind = 0

    def step(record):
        diffs = []
        if args.per_pair_sync or args.two_pass:
            results = [algebra(sets[i], sets[j]) for (i, j) in pairs]
        else:
            results = ctx.pair_algebra_batch([(sets[i], sets[j]) for (i, j) in pairs])
        if hasattr(results, "totals"):      # the batch result: sizes without touching the sets
            sizes = [(int(t[1]), int(t[2])) for t in results.totals]
        else:
            sizes = [(amb.n_keys, bma.n_keys) for (_inter, amb, bma) in results]
        for (i, j), (n_amb, n_bma) in zip(pairs, sizes):
            diffs.append(n_amb + n_bma)
            if record:
                union = sets[i].n_keys + n_bma
                algo_bytes[0] += (sets[i].n_keys + sets[j].n_keys + union) * g.key_bytes
        if world > 1:
            slot = step_no[0] & 1
            step_no[0] += 1
            diff_host_np[slot][:] = diffs
            diff_local[slot].copy_(diff_host[slot], non_blocking=True)
            work = dist.all_gather(gathered[slot], diff_local[slot], async_op=True)
            if pending[0] is not None:
                pending[0].wait()      # the previous step's exchange; this one overlaps the next step
            pending[0] = work
        return diffs

    def fence():
        if world > 1:
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    ctx.enable_timing(max(1, args.time_every))
    ctx.timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        diffs = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    write_ms, write_launches = ctx.timing_read(kernel_kind)
    count_ms, count_launches = ctx.timing_read(1)
    ctx.enable_timing(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([units_per_step], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_units_per_step = int(u.item())
    else:
        total_units_per_step = units_per_step

    value = total_units_per_step * args.steps / elapsed / 1e6

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as ol

        # The reference's algebra for one merge (kmer_set_set.h:339-343) on hash-bucket
        # sets: Intersection(j, k) [two by-value copies + two Sub], then j.Sub(n), k.Sub(n).
        done_units, spent = 0, 0.0
        used = []
        # one more (untimed) batch whose three outputs per pair are compared with the oracle's by
        # size and XOR hash (BASELINE config 2: counts and hashes bit-exact against the CPU path)
        check_res = ctx.pair_algebra_batch([(sets[i], sets[j]) for (i, j) in pairs])
        gpu_sig = [[(s_.n_keys, ctx.set_hash(s_)) for s_ in check_res[idx]] for idx in range(len(pairs))]
        for (i, j) in pairs:
            a = ol.Set.from_kmers(k, nbits, g.key_bytes, host_sets[i])
            b = ol.Set.from_kmers(k, nbits, g.key_bytes, host_sets[j])
            c0 = time.perf_counter()
            n = a.intersection(b)
            a.sub_set(n)
            b.sub_set(n)
            spent += time.perf_counter() - c0
            done_units += host_sets[i].size + host_sets[j].size
            used.append((i, j))
            assert a.size() + b.size() == diffs[pairs.index((i, j))]
            cpu_sig = [(n.size(), n.hash()), (a.size(), a.hash()), (b.size(), b.hash())]
            assert cpu_sig == gpu_sig[pairs.index((i, j))], "pair %s: GPU %s != oracle %s" % (
                (i, j), gpu_sig[pairs.index((i, j))], cpu_sig)
            if spent >= args.cpu_baseline_seconds:
                break
        cpu_baseline = {
            "value": done_units / spent / 1e6,
            "unit": "Mk-mers/s",
            "cores": 1,
            "kind": "port",
            "sample": "oracle hash-set algebra (Intersection + 2 Sub) on pairs %s of the same "
                      "workload, full size, set construction excluded, %.1f s of CPU" % (used, spent),
            "checked": "sizes and XOR hashes of A&B, A\\B, B\\A of those pairs: GPU == oracle",
        }

    spss = None
    if rank == 0 and world == 1 and not args.no_spss:
        compacts = [ctx.spss_encode(s, mode=0) for s in sets]
        ids = synth.sample_bucket_ids(nbits, seed=args.seed + 1)
        torch.cuda.synchronize()
        l0 = time.perf_counter()
        kss = capi.DeviceKmerSetSet(ctx, compacts, ids)
        torch.cuda.synchronize()
        lwall = time.perf_counter() - l0
        st = kss.stats()
        total = sum(s.n_keys for s in sets)
        spss = {
            "bytes_per_kmer": (st["packed_bytes"] + st["length_bytes"]) / total,
            "chars_per_kmer_before": st["initial_spss_weight"] / total,
            "chars_per_kmer_after": st["final_spss_weight"] / total,
            "nodes": st["nodes"], "iterations": int(kss.trace()[0].shape[0]),
            "loop_mkmers_per_s": st["n_processed"] / lwall / 1e6, "loop_wall_ms": lwall * 1e3,
            "note": "whole KmerSetSet constructor on the same %d sets, one run, outside the timed "
                    "region; bytes = sum over nodes of ceil(2 * Weight / 8) + StreamVByte-0124 size of the lengths" % n_sets,
        }
        kss.close()

    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if rank == 0 and os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get("kernel") == "k_tile_merge<KeyT, 1>" and pmc.get("kmers_per_set") == size \
                and pmc.get("k") == k and pmc.get("sets") == n_sets:
            # the counters were collected on batched launches; a per-pair launch moves a
            # proportional share
            per_launch_pairs = 1 if (args.per_pair_sync or args.two_pass) else len(pairs)
            traffic = pmc["bytes_per_launch"] * per_launch_pairs / pmc.get("pairs_per_launch", 1)
            traffic_src = "profiles/pmc_traffic.json"

    if rank == 0:
        # algo_bytes covers every launch of the timed region, write_ms the sampled ones
        # one write-pass launch per pair, or one for the whole batch of a step
        all_launches = args.steps * (len(pairs) if (args.per_pair_sync or args.two_pass) else 1)
        bytes_per_launch = algo_bytes[0] / max(all_launches, 1)
        avg_launch_ms = write_ms / max(write_launches, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if write_ms > 0 else 0.0
        out = {
            "metric": "Mk-mers/s processed in kmerset-multiple-compress; bytes/k-mer after SPSS",
            "value": value,
            "unit": "Mk-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if g.key_bytes == 4 else "u64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: %d canonical k=%d sets of %d k-mers (seeded phylogeny family), "
                            "all %d pairs: A&B, A\\B, B\\A + counts per pair" % (n_sets, k, size, len(pairs)),
                "k": k, "n_bucket_bits": nbits, "key_bytes": g.key_bytes,
                "sets_per_gpu": n_sets, "kmers_per_set": [s.n_keys for s in sets],
                "pairs_per_step": len(pairs), "units_per_step": total_units_per_step,
                "parallelism": "1 process per GPU, pairs sharded by rank, RCCL all-gather of "
                               "per-pair diff sizes" if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_tile_merge<KeyT, 1> (write pass)",
                "pairs_per_launch": 1 if (args.per_pair_sync or args.two_pass) else len(pairs),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches": int(all_launches),
                "launches_timed": int(write_launches),
                "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "count_pass_avg_launch_ms": (count_ms / count_launches) if count_launches else None,
            },
            "cpu_baseline": cpu_baseline,
            "spss": spss,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
