#!/usr/bin/env python3
"""bench.py -- Mk-mers/s of kmerset-multiple-compress's hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

What is timed is what the reference times around the KmerSetSet constructor
(src/kmerset-multiple-compress.cc:96-101, lib/core/kmer_set_set.h:109-427): the inputs are
KmerSetCompact containers already resident in HBM; one STEP = one whole ksh_kss_build -- decode
of the inputs, sampled weight table, every merge iteration (Intersection + Sub + Sub), the SPSS
re-encodes the convergence checks and the result need.  Mk-mers/s = N_proc / wall with
N_proc = sum |S_i| + sum over iterations (|S_j| + |S_k|).

Workload at N = 1: 64 synthetic canonical k=23 sets of 10^8 k-mers (the sets of BASELINE.json
configs[3], the case north_star quotes the 1-GPU target on; 25.6 GB of resident keys), seeded
phylogeny family generated on the device before anything is timed.

N > 1 (one process per GPU, launched by torch.distributed.run): the SAME 64 sets, strong
scaling: the owner-sharded build (ksh_kss_build_owned) -- every input set is decoded and kept
by one rank, the 2 % sampled slices are all-gathered, the control loop runs replicated on the
samples, a merge and the encodes of its results run on the rank that owns the node, RCCL on
device buffers.

Prints ONE JSON line on rank 0.
  roofline     the loop's dominant stage, the neighbour probe of the SPSS encode (the rc partition,
               k_adj_rc1 and k_adj_fwd_targets; one timed "launch" = the stage of one encode):
               algorithmic bytes = 5.3 B per k-mer (read key + write adjacency byte
               + packed bases) x the k-mers of a launch, over its HIP-event duration on the
               context's stream, against the 8 TB/s HBM peak; the probe-inclusive figure
               (36 B per k-mer: the key + 8 neighbour lookups) is reported beside it; `traffic`
               from profiles/pmc_adjacency.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).
               The timed builds run the independent encodes of a check on several streams at once
               (lanes, ksh_ctx_set_lanes): a launch's duration there depends on what shares the GPU
               with it, so the kernel's own figure comes from ONE more build of the same inputs on ONE
               stream, run inside this script right after the timed region; the timed region's sums
               (stream time and union of the spans) are under roofline.timed_region.
  cpu_baseline the oracle's port of the same loop, compiled on this host with the reference's release
               flags (-O3 -march=native -DNDEBUG), bucket-parallel threads as the reference has
               them: the first I iterations of a 16 x 10^7 family at n_workers = all host cores,
               the GPU timed on the SAME sample beside it (same I, same N_proc, merge sequence
               compared); a smaller sample at 1 worker (rank 0, N = 1 only).
  pair_merge   the pair-algebra kernel on configs[1] (4 x 10^7), the round-1 headline, as an
               extra block.
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
ENCODE_BYTES_PER_KMER = 5.3       # compulsory read s + 1 B adjacency + w/4 packed bases
ENCODE_PROBE_BYTES_PER_KMER = 36  # s + 8 * s, the 8 neighbour lookups missing LDS


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may use: the CPU affinity mask, cut down to the cgroup's CPU quota when there is
    one (a GPU box hands a job a share of its host, not the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=23)
    ap.add_argument("--bucket-bits", type=int, default=14)
    ap.add_argument("--sets", type=int, default=64)
    ap.add_argument("--size", type=float, default=1e8, help="k-mers per set")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--max-iterations", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the Size / XOR-Hash check of every Get(i) after the timed builds")
    ap.add_argument("--no-pair-merge", action="store_true", help="skip the configs[1] pair-algebra block")
    ap.add_argument("--cpu-sets", type=int, default=16)
    ap.add_argument("--cpu-size", type=float, default=1e7, help="k-mers per set of the all-cores CPU sample")
    ap.add_argument("--cpu-size-1", type=float, default=2e6, help="k-mers per set of the 1-worker CPU sample")
    ap.add_argument("--cpu-iterations", type=int, default=4)
    ap.add_argument("--cpu-workers", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--lanes", type=int, default=0,
                    help="independent encodes / decodes of a build on this many HIP streams at once "
                         "(ksh_ctx_set_lanes; 0 = the library's default, 1 = one stream)")
    ap.add_argument("--owned", action="store_true",
                    help="with --gpus 1: the owner-sharded build (ksh_kss_build_owned) over a ONE-rank RCCL communicator "
                         "instead of ksh_kss_build -- what the multi-GPU path costs on one GPU (its control loop on "
                         "samples, the deferred merges), for the N-GPU model's terms")
    ap.add_argument("--dump-trace", default="",
                    help="write the last build's merge sequence, node sizes and phase times to this file "
                         "(input of tools/owned_schedule.py)")
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
    # Rehearsal on a one-GPU box: KSH_BENCH_BACKEND=gloo puts every rank on cuda:0 and runs the
    # exchanges through host memory (RCCL refuses two ranks on one device).
    backend = os.environ.get("KSH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    owned = world > 1 or args.owned
    if owned:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29591")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    from kmersets import capi, synth, synth_torch

    k, nbits, n_sets, size = args.k, args.bucket_bits, args.sets, int(args.size)
    g = capi.geom(k, nbits)
    ctx = capi.Context(local_rank)
    ctx.set_lanes(args.lanes)
    dev = ctx.device
    ids = synth.sample_bucket_ids(nbits, seed=args.seed + 1)

    # ---- inputs: KmerSetCompact containers resident in HBM before anything is timed.  In a
    # multi-GPU run a rank only builds the containers of the sets it owns (contiguous blocks:
    # set i -> rank i * world // n_sets).
    owners = capi.block_owners(n_sets, world)
    t0 = time.perf_counter()
    kmers = synth_torch.phylogeny_sets(k, n_sets, size, args.seed, dev)
    sizes = [int(km.numel()) for km in kmers]
    compacts = []
    for i, km in enumerate(kmers):
        if world > 1 and owners[i] != rank:
            compacts.append(None)
        else:
            compacts.append(ctx.spss_encode(synth_torch.device_set(g, km), mode=0))
        kmers[i] = None
    del kmers
    torch.cuda.synchronize()
    t_inputs = time.perf_counter() - t0

    comm = capi.Comm(ctx, dist, coll_dev) if owned else None     # one RCCL communicator for all builds

    def build():
        if not owned:
            return capi.DeviceKmerSetSet(ctx, compacts, ids, max_iterations=args.max_iterations)
        return capi.OwnedKmerSetSet(ctx, compacts, ids, dist, coll_dev, max_iterations=args.max_iterations,
                                    owners=owners, comm=comm)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_wall = None
    for _ in range(max(0, args.warmup)):
        fence()
        w0 = time.perf_counter()
        warm = build()
        fence()
        if first_wall is None:
            first_wall = time.perf_counter() - w0
        warm.close()
    ctx.enable_timing(1)
    ctx.timing_reset()
    fence()
    t0 = time.perf_counter()
    kss = None
    for _ in range(args.steps):
        if kss is not None:
            kss.close()
        kss = build()
    fence()
    elapsed = time.perf_counter() - t0
    KINDS = (("k_adjacency", 3), ("ranking_walks", 4), ("emit_walks", 5))
    timers = {name: (ctx.timing_read(kind), ctx.timing_units(kind), ctx.timing_wall(kind)) for name, kind in KINDS}
    ctx.enable_timing(False)
    # The roofline leg proper: ONE more build of the same inputs on ONE stream (lanes = 1), every launch of the
    # stage bracketed by HIP events on that stream.  In the timed region above several encodes share the GPU
    # (lanes): a launch's duration there depends on whose kernels run beside it, so it says how the streams
    # overlap, not what the kernel does with the GPU; both are reported (`roofline.timed_region`).
    lanes_used = args.lanes if args.lanes > 0 else int(os.environ.get("KSH_LANES", "4"))
    timers_timed, excl_wall = timers, None
    st = kss.stats()                 # (of the last TIMED build: phases, encode counts, SPSS sizes)
    it, cp, imp = kss.trace()
    if not owned and lanes_used != 1:
        kss.close()
        ctx.set_lanes(1)
        ctx.enable_timing(1)
        ctx.timing_reset()
        fence()
        e0 = time.perf_counter()
        kss = build()
        fence()
        excl_wall = time.perf_counter() - e0
        timers = {name: (ctx.timing_read(kind), ctx.timing_units(kind), ctx.timing_wall(kind)) for name, kind in KINDS}
        ctx.enable_timing(False)
        ctx.set_lanes(args.lanes)
        phase_seconds_one_stream = kss.stats()["phase_seconds"]
    else:
        phase_seconds_one_stream = None

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_proc = st["n_processed"]
    value = n_proc * args.steps / elapsed / 1e6
    multi_gpu = None
    if owned:
        cs = kss.comm_stats()
        v = torch.tensor([st["n_encodes"], st["n_encoded_kmers"], cs["p2p_sets"], cs["p2p_bytes_sent"],
                          cs["gather_bytes"]], dtype=torch.int64, device=coll_dev)
        parts = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        rows = [[int(x) for x in p.tolist()] for p in parts]
        ranks_seen = comm.ranks_seen()       # every rank's id through the transport's own all-gather
        multi_gpu = {"transport": comm.kind, "ranks_seen": ranks_seen,
                     "control_weights": "sharded by pair list + all-gather of int64 per iteration (%d all-gathers)"
                                        % cs["weight_gathers"] if cs["weight_gathers"] else
                                        "replicated on the all-gathered samples (no exchange)",
                     "encodes_per_rank": [r[0] for r in rows],
                     "encoded_kmers_per_rank": [r[1] for r in rows], "sets_sent_per_rank": [r[2] for r in rows],
                     "p2p_bytes_sent_per_rank": [r[3] for r in rows], "allgather_bytes_per_rank": [r[4] for r in rows],
                     "note": "last timed build"}

    if args.dump_trace and rank == 0 and not owned:
        n_in = len(compacts)
        node_sizes = [kss.node_size(i) for i in range(kss.size())]
        # sizes of (child, j', k') right after every merge: replay the trace backwards from the final sizes
        cur = list(node_sizes)
        triples = [None] * len(it)
        for t in range(len(it) - 1, -1, -1):
            j_, k_ = int(it[t][0]), int(it[t][1])
            child = n_in + t
            triples[t] = [cur[child], cur[j_], cur[k_]]
            # before the merge: |j| = |j'| + |child as it was born|; the child's birth size is its size now plus
            # whatever later merges moved out of it into ITS children, which the backward replay has restored
            cur[j_] += cur[child]
            cur[k_] += cur[child]
        json.dump({"n_inputs": n_in, "key_bytes": g.key_bytes, "input_sizes": sizes, "trace": it.tolist(),
                   "result_sizes": triples, "phase_seconds": st["phase_seconds"],
                   "n_encoded_kmers": st["n_encoded_kmers"], "n_encodes": st["n_encodes"]},
                  open(args.dump_trace, "w"))

    verified = None
    if not args.no_verify:
        # the reference's own --check (src/kmerset-multiple-compress.cc:104-126) on the device: Size
        # and XOR Hash of Get(i) against the decoded input, every i (every rank checks the sets it owns)
        bad, checked = [], 0
        for i, c in enumerate(compacts):
            if c is None:
                continue
            want = ctx.spss_decode(c)
            got = kss.get_size_and_hash(i)
            checked += 1
            if got != (want.n_keys, ctx.set_hash(want)):
                bad.append(i)
            del want
        if owned:
            b = torch.tensor([len(bad), checked], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(b)
            n_bad, checked = int(b[0].item()), int(b[1].item())
        else:
            n_bad = len(bad)
        verified = {"sets": checked, "mismatches": n_bad,
                    "method": "Size and XOR Hash of Get(i) == those of the decoded input i, on the device, "
                              "after the last timed build"}
        if n_bad:
            raise SystemExit("verification failed: %d of %d sets (rank %d: %s)" % (n_bad, checked, rank, bad))

    total = sum(sizes)
    spss = {
        "bytes_per_kmer": (st["packed_bytes"] + st["length_bytes"]) / total,
        "chars_per_kmer_before": st["initial_spss_weight"] / total,
        "chars_per_kmer_after": st["final_spss_weight"] / total,
        "nodes": st["nodes"], "iterations": int(it.shape[0]), "checkpoints": int(cp.shape[0]),
        "encodes": st["n_encodes"], "encoded_kmers": st["n_encoded_kmers"],
        "of_them_weight_only": st.get("n_weighed", 0), "weight_only_kmers": st.get("n_weighed_kmers", 0),
        "note": "from the last timed build; bytes = sum over nodes of ceil(2 * Weight / 8) + StreamVByte-0124 "
                "size of the lengths (SURVEY.md 8d, metric 2)",
    }
    if owned:
        lb = torch.tensor([st["length_bytes"]], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(lb)       # a node's lengths live on the rank that holds its SPSS
        spss["bytes_per_kmer"] = (st["packed_bytes"] + int(lb.item())) / total

    # ---- pair-merge block (configs[1]; the round-1 headline, kept as an extra key)
    pair_merge = None
    if rank == 0 and not owned and not args.no_pair_merge:
        kss.close()
        kss = None
        pm_sets = [synth_torch.device_set(g, km) for km in synth_torch.phylogeny_sets(k, 4, int(1e7), 2, dev)]
        pairs = [(i, j) for i in range(4) for j in range(i + 1, 4)]
        jobs = [(pm_sets[i], pm_sets[j]) for i, j in pairs]
        for _ in range(3):
            ctx.pair_algebra_batch(jobs)
        ctx.enable_timing(1)
        ctx.timing_reset()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        pm_steps = 10
        for _ in range(pm_steps):
            res = ctx.pair_algebra_batch(jobs)
        torch.cuda.synchronize()
        pm_wall = time.perf_counter() - p0
        wms, wl = ctx.timing_read(0)
        cms, cl = ctx.timing_read(1)
        ctx.enable_timing(False)
        units = sum(a.n_keys + b.n_keys for a, b in jobs)
        algo = sum((a.n_keys + b.n_keys + a.n_keys + int(t[2])) * g.key_bytes for (a, b), t in zip(jobs, res.totals))
        pair_merge = {
            "workload": "configs[1]: 4 canonical k=23 sets of 10^7 k-mers, all 6 pairs per step: A&B, A\\B, B\\A + counts",
            "mkmers_per_s": units * pm_steps / pm_wall / 1e6, "ms_per_step": pm_wall / pm_steps * 1e3,
            "write_pass_avg_launch_ms": wms / max(wl, 1), "count_pass_avg_launch_ms": cms / max(cl, 1),
            "algorithmic_bytes_per_launch": algo,
            "write_pass_frac_of_hbm_peak": algo / (wms / max(wl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if wms > 0 else None,
            "step_frac_of_hbm_peak": algo / (pm_wall / pm_steps) / 1e9 / HBM_PEAK_GBS,
            "note": "kernel-level fraction = algorithmic bytes / write-pass launch; step-level = the same bytes "
                    "over the whole step (count pass + plan + write pass + read-back)",
        }
        del pm_sets, jobs, res

    # ---- CPU baseline: the oracle's port of the loop, threads as the reference has them
    cpu_baseline = None
    if rank == 0 and not owned and not args.no_cpu_baseline:
        # the copy that is timed is compiled here, on the host it runs on, with the reference's release
        # flags (CMakeLists.txt:5); the portable build the tests load is the fall-back
        import subprocess
        import tempfile

        import shutil

        native_dir = tempfile.mkdtemp(prefix="ksh_oracle_")
        native = os.path.join(native_dir, "libkmersets_oracle_native.so")
        oracle_build, oracle_native = "-O3 -march=native -DNDEBUG (compiled on this host)", True
        try:
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", "NATIVE_OUT=" + native],
                           check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.environ["KSH_ORACLE_LIB"] = native
        except Exception as e:  # noqa: BLE001 -- no compiler on this host: the prebuilt portable copy, and say so
            oracle_build, oracle_native = "-O3 (prebuilt portable copy: the native build failed)", False
            print("bench.py: the native build of the oracle failed (%s): the CPU baseline is timed on the portable "
                  "-O3 copy" % e, file=sys.stderr)
        import oracle_lib as ol

        ol.lib()  # (loaded now: the temporary directory can go once the library is mapped)
        os.environ.pop("KSH_ORACLE_LIB", None)
        shutil.rmtree(native_dir, ignore_errors=True)

        cs, ci = args.cpu_sets, args.cpu_iterations
        # the reference posts n_workers^2 chunks per parallel step, each with its own 2^N key buffers
        # (spss.h:1896-1901): beyond a few dozen workers the chunk set-up dominates, so the "all cores"
        # run is capped at 32 workers (measured on the 256-thread host: 256 workers are 27x SLOWER than 1)
        cores = args.cpu_workers or min(host_cores(), 32)

        def sample(csize, workers):
            """The same family at `csize` k-mers per set: the oracle's loop at `workers`, then the GPU's
            (one untimed build, one timed) on the same containers; both stop after `ci` iterations."""
            km = synth_torch.phylogeny_sets(k, cs, csize, args.seed, dev)
            gc = [ctx.spss_encode(synth_torch.device_set(g, x), mode=0) for x in km]
            del km
            oc = [ol.Compact.from_strings(c.to_strings(), k, nbits, g.key_bytes) for c in gc]
            c0 = time.perf_counter()
            okss = ol.KmerSetSet(oc, ids, max_iterations=ci, n_workers=workers)
            cw = time.perf_counter() - c0
            o_proc, o_it = okss.stat(3), [tuple(r[:5]) for r in okss.iterations()]
            del okss, oc
            capi.DeviceKmerSetSet(ctx, gc, ids, max_iterations=ci).close()
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            gk = capi.DeviceKmerSetSet(ctx, gc, ids, max_iterations=ci)
            torch.cuda.synchronize()
            gw = time.perf_counter() - g0
            g_it = [tuple(int(x) for x in r) for r in gk.trace()[0]]
            same = g_it == o_it and gk.stats()["n_processed"] == o_proc
            gk.close()
            if not same:
                raise SystemExit("CPU baseline sample: the oracle's merge sequence differs from the GPU's")
            return {"n_processed": o_proc, "cpu_s": cw, "gpu_s": gw, "cpu_mkmers_per_s": o_proc / cw / 1e6,
                    "gpu_mkmers_per_s": o_proc / gw / 1e6, "size": csize, "workers": workers}

        big = sample(int(args.cpu_size), cores)
        one = sample(int(args.cpu_size_1), 1)
        cpu_baseline = {
            "value": big["cpu_mkmers_per_s"], "unit": "Mk-mers/s", "cores": cores, "kind": "port",
            "cpu": cpu_model(), "host_cores_available": host_cores(), "oracle_build": oracle_build,
            "oracle_native": oracle_native,
            "sample": "oracle KmerSetSet (C++ port of lib/core/kmer_set_set.h:109-427 with the reference's "
                      "bucket-parallel / pooled structure), %d sets of %d k-mers of the same family, first %d "
                      "iterations, N_proc = %d; %.1f s at %d workers"
                      % (cs, big["size"], ci, big["n_processed"], big["cpu_s"], cores),
            "gpu_same_sample": {"value": big["gpu_mkmers_per_s"], "seconds": big["gpu_s"],
                                "speedup_vs_cpu": big["cpu_s"] / big["gpu_s"],
                                "note": "the same containers, the same %d iterations, the same N_proc, one GPU build "
                                        "(after one untimed build)" % ci},
            "value_1_core": one["cpu_mkmers_per_s"],
            "sample_1_core": "%d sets of %d k-mers, first %d iterations, N_proc = %d; %.1f s at 1 worker; the GPU "
                             "on the same sample: %.1f Mk-mers/s" % (cs, one["size"], ci, one["n_processed"],
                                                                      one["cpu_s"], one["gpu_mkmers_per_s"]),
            "checked": "merge sequence (j, k, weight, sizes) and N_proc of both samples: GPU == oracle",
        }

    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_adjacency.json")
    if rank == 0 and os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get("k") == k and pmc.get("bytes_per_kmer"):
            traffic_per_kmer = pmc["bytes_per_kmer"]
            traffic_src = "profiles/pmc_adjacency.json"
        else:
            traffic_per_kmer = None
    else:
        traffic_per_kmer = None

    if rank == 0:
        # `timers`: the one-stream build (the stage's launches alone on the GPU: sum of spans == their union);
        # `timers_timed`: the timed region, where launches of different lanes overlap -- stream time = the sum of
        # the spans (what a kernel trace's per-kernel durations add up to), union = the time during which the
        # stage was running on at least one stream.
        (adj_ms, adj_launches), adj_units, adj_wall_ms = timers["k_adjacency"]
        kmers_per_launch = adj_units / max(adj_launches, 1)
        avg_launch_ms = adj_ms / max(adj_launches, 1)
        bytes_per_launch = ENCODE_BYTES_PER_KMER * kmers_per_launch
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if adj_ms > 0 else 0.0
        (t_ms, t_launches), t_units, t_wall_ms = timers_timed["k_adjacency"]
        timed_region = {
            "lanes": lanes_used, "launches": int(t_launches), "kmers": int(t_units),
            "stream_time_ms": t_ms, "union_ms": t_wall_ms,
            "ns_per_kmer_stream_time": t_ms * 1e6 / max(t_units, 1), "ns_per_kmer_union": t_wall_ms * 1e6 / max(t_units, 1),
            "frac_stream_time": ENCODE_BYTES_PER_KMER * t_units / max(t_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
            "frac_union": ENCODE_BYTES_PER_KMER * t_units / max(t_wall_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
            "share_of_timed_region_union": t_wall_ms * 1e-3 / elapsed,
            "note": "HIP events around every launch of the stage in the %d timed builds; launches of different lanes "
                    "overlap and share the GPU by time, so a launch takes longer there than alone" % args.steps,
        }
        if traffic_per_kmer is not None:
            traffic = traffic_per_kmer * kmers_per_launch
        other = {}
        for name in ("ranking_walks", "emit_walks"):
            (ms, n), units, wall = timers[name]
            (tms, tn), tunits, twall = timers_timed[name]
            other[name] = {"ms_total": ms, "launches": n, "ns_per_kmer": ms * 1e6 / max(units, 1),
                           "timed_region_ns_per_kmer_stream_time": tms * 1e6 / max(tunits, 1),
                           "timed_region_ns_per_kmer_union": twall * 1e6 / max(tunits, 1)}
        out = {
            "metric": "Mk-mers/s processed in kmerset-multiple-compress; bytes/k-mer after SPSS",
            "value": value,
            "unit": "Mk-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": {2: "u16", 4: "u32", 8: "u64"}[g.key_bytes],
            "data": "synthetic",
            "config": {
                "workload": "%d canonical k=%d sets of %d k-mers (seeded phylogeny family), the whole KmerSetSet "
                            "constructor per step: decode, weights, every merge, SPSS encodes (configs[3]'s sets; "
                            "north_star's 1-GPU case)" % (n_sets, k, size),
                "k": k, "n_bucket_bits": nbits, "key_bytes": g.key_bytes,
                "sets": n_sets, "sum_input_kmers": total, "n_processed_per_step": n_proc,
                "iterations": int(it.shape[0]), "nodes": st["nodes"],
                "input_build_s": t_inputs, "first_build_wall_s": first_wall,
                "sum_input_mkmers_per_s": total * args.steps / elapsed / 1e6,
                "parallelism": "1 GPU" if not owned else
                               "1 process per GPU, owner-sharded sets (contiguous blocks of %d), replicated control "
                               "loop on the all-gathered 2 %% samples, merges and encodes on the owner of j, %s "
                               "all-gather / send-recv on device buffers" % (n_sets // world, comm.kind.upper()),
            },
            "phase_seconds_last_build": st["phase_seconds"],
            "phase_seconds_one_stream_build": phase_seconds_one_stream,
            "roofline": {
                "bound": "hbm",
                "kernel": "neighbour-probe stage of the SPSS encode (k_rc_hist / k_rc_columns / k_rc_scatter_l1+l2 / k_rc_bounds / "
                          "k_adj_rc1 (k_adj_rc for groups whose records do not fit it) / k_tgt_bounds / k_tgt_split / k_tgt_subcuts / k_adj_fwd_targets; the in-place k_adjacency for sets outside their range): "
                          "one timed launch = the whole stage of one encode",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches": int(adj_launches),
                "avg_launch_ms": avg_launch_ms,
                "timing": "HIP events on the context's stream around every launch of the stage, in one build of the "
                          "same inputs on ONE stream (lanes = 1) run right after the timed builds%s"
                          % ("" if excl_wall is None else ": %.1f ms for that build" % (excl_wall * 1e3)),
                "kmers_per_launch": kmers_per_launch,
                "ns_per_kmer": adj_ms * 1e6 / max(adj_units, 1),
                "timed_region": timed_region,
                "algorithmic_bytes_per_kmer": ENCODE_BYTES_PER_KMER,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "probe_inclusive_bytes_per_kmer": ENCODE_PROBE_BYTES_PER_KMER,
                "probe_inclusive_frac": achieved / HBM_PEAK_GBS * ENCODE_PROBE_BYTES_PER_KMER / ENCODE_BYTES_PER_KMER,
                "share_of_one_stream_build": adj_ms * 1e-3 / (excl_wall if excl_wall else elapsed / args.steps),
                "other_kernels": other,
            },
            "cpu_baseline": cpu_baseline,
            "spss": spss,
            "verified": verified,
            "multi_gpu": multi_gpu,
            "pair_merge": pair_merge,
        }
        print(json.dumps(out))
    if kss is not None:
        kss.close()
    if comm is not None:
        comm.close()
    if owned:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
