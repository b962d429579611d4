#!/usr/bin/env python3
"""Phase-by-phase time model of the owner-sharded build (ksh_kss_build_owned) on N GPUs, replayed over
the merge sequence of a real single-GPU run (bench.py --dump-trace): who owns which node after every
merge (contiguous blocks, merge on the owner of j, k's remainder and the child stay there), which sets
travel, and what every rank encodes at every convergence check.  The per-phase rates are the
single-GPU measurements of the same run (its `phase_seconds`), so the output is the expected N-GPU
wall time with every check resolved on the spot (KSH_OWNED_LOOKAHEAD=0) and with the checks deferred
by one, which is what the code does; it is arithmetic, not a measurement.

Both control modes are reported: the replicated control loop (the default: every rank weighs its replica of
the samples, nothing is exchanged, the merges of other ranks are not waited for) and KSH_OWNED_WEIGHTS=sharded
(the weight tables dealt out by pair list, ONE all-gather of int64 per iteration: the weighing shrinks by N,
every iteration pays the all-gather's latency and waits for the rank that runs the iteration's full merge).

  owned_schedule.py trace.json [--gpus 8] [--link-gbs 64] [--allgather-us 40]
"""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--link-gbs", type=float, default=64.0, help="one xGMI link, one direction, GB/s")
    ap.add_argument("--allgather-us", type=float, default=40.0,
                    help="one small all-gather of int64 over RCCL incl. its host staging, microseconds (assumed)")
    ap.add_argument("--sample-merge-us", type=float, default=150.0,
                    help="the merge of an iteration's two 2 %% samples, which every rank runs for itself (plan with its "
                         "read-back + write; assumed -- the single-GPU build has no counterpart to measure)")
    args = ap.parse_args()
    d = json.load(open(args.trace))
    n0, key_bytes, world = d["n_inputs"], d["key_bytes"], args.gpus
    rows = d["trace"]                      # j, k, weight, original_size, size_diff
    sizes_now = list(d["input_sizes"])     # node sizes as the loop goes
    ph = d["phase_seconds"]                # decode_inputs, weights, merges, encodes of the 1-GPU build
    enc_kmers = d["n_encoded_kmers"]
    enc_rate = ph["encodes"] / enc_kmers   # s per encoded k-mer (whole 1-GPU loop average)
    merge_rate = ph["merges"] / max(sum(r[3] for r in rows), 1)
    interval = n0 // 8 + 1
    owner = [i * world // n0 for i in range(n0)]
    stale = set()
    t_barrier = t_look = 0.0
    pending = []                           # per-check lists of (rank, k-mers) for the lookahead variant
    t_p2p = t_merge = 0.0
    sets_moved = bytes_moved = 0
    per_check = []
    # the full-size merges deferred to the end of their interval (what the library does since round 4): every rank
    # works its part of the interval's list off in iteration order; a receiver's merge waits for the sender to get
    # there.  Per-rank clocks inside an interval, the interval costs the latest one.
    t_merge_deferred = 0.0
    clock = [0.0] * world

    def check():
        nonlocal t_barrier, t_merge_deferred, clock
        t_merge_deferred += max(clock)
        clock = [0.0] * world
        load = [0.0] * world
        for node in stale:
            load[owner[node]] += sizes_now[node] * enc_rate
        per_check.append(load)
        t_barrier += max(load)
        pending.append(load)
        stale.clear()

    # node sizes after a merge need |n|, |j'|, |k'|: the trace has their sum only; the split comes from the
    # run's final node sizes for nodes that are never merged again and is approximated (child = the rest)
    # otherwise -- the dump carries the exact triple when bench.py wrote it
    triples = d.get("result_sizes")
    for it, (j, k, w, original, diff) in enumerate(rows):
        if it > 0 and it % interval == 0:
            check()
        ex, src = owner[j], owner[k]
        if src != ex:
            nbytes = sizes_now[k] * key_bytes
            t_p2p += nbytes / (args.link_gbs * 1e9)
            sets_moved += 1
            bytes_moved += nbytes
            arrive = clock[src] + nbytes / (args.link_gbs * 1e9)
            clock[src] = arrive
            clock[ex] = max(clock[ex], arrive)
        t_merge += original * merge_rate
        clock[ex] += original * merge_rate
        nn, nj, nk = triples[it]
        sizes_now[j], sizes_now[k] = nj, nk
        sizes_now.append(nn)
        owner[k] = ex
        owner.append(ex)
        stale.update((j, k, len(owner) - 1))
    check()
    # lookahead by one check: the encodes of check c and c + 1 are one pool per rank (a rank that is done
    # with check c goes on with c + 1's merges and encodes; the sum of check c arrives asynchronously)
    for c in range(0, len(pending), 2):
        pool = [sum(x) for x in zip(*pending[c:c + 2])]
        t_look += max(pool)
    # ---- the schedule the library runs by default: a check's exchange is started after the rank's encodes
    # and waited for at the NEXT check; before encoding, a check's nodes are dealt out again (the most loaded
    # rank hands its largest node to the least loaded one while that shortens the longer of the two, a
    # quarter of its cost for the trip, a rank either gives or takes; loads carry each rank's lag from the
    # checks before) -- ksh_kss.hip build_owned, encode_stale.  Per-rank timelines:
    def implemented():
        sizes2 = list(d["input_sizes"])
        own = [i * world // n0 for i in range(n0)]
        st2 = set()
        t = [0.0] * world
        lag = [0.0] * world
        done_prev = 0.0
        moved = 0

        def check2():
            nonlocal done_prev, moved
            tasks = sorted(st2)
            cost = {n: sizes2[n] * enc_rate for n in tasks}
            enc = {n: own[n] for n in tasks}
            load = list(lag)
            for n in tasks:
                load[own[n]] += cost[n]
            role = [0] * world
            for _ in tasks:
                o = max(range(world), key=lambda r: (load[r], -r))
                m = min(range(world), key=lambda r: (load[r], r))
                if o == m or role[o] < 0 or role[m] > 0:
                    break
                cand = [n for n in tasks if enc[n] == o and own[n] == o and load[m] + 1.25 * cost[n] < load[o]]
                if not cand:
                    break
                n = max(cand, key=lambda x: cost[x])
                enc[n] = m
                load[o] -= cost[n]
                load[m] += 1.25 * cost[n]
                role[o], role[m] = 1, -1
            lightest = min(load)
            for r in range(world):
                lag[r] = load[r] - lightest
            start = list(t)
            for r in range(world):
                for n in tasks:
                    if enc[n] == r and own[n] == r:
                        t[r] += cost[n]
            for n in tasks:
                if enc[n] != own[n]:
                    r = enc[n]
                    arrive = start[own[n]] + sizes2[n] * key_bytes / (args.link_gbs * 1e9)
                    t[r] = max(t[r], arrive) + cost[n]
                    own[n] = r
                    moved += 1
            done_now = max(t)
            for r in range(world):
                t[r] = max(t[r], done_prev)       # the exchange of the check before, waited for here
            done_prev = done_now
            st2.clear()

        for it2, (j2, k2, w2, orig2, diff2) in enumerate(rows):
            if it2 > 0 and it2 % interval == 0:
                check2()
            nn2, nj2, nk2 = triples[it2]
            sizes2[j2], sizes2[k2] = nj2, nk2
            sizes2.append(nn2)
            own[k2] = own[j2]
            own.append(own[j2])
            st2.update((j2, k2, len(own) - 1))
        check2()
        return max(max(t), done_prev), moved

    t_impl, moved_impl = implemented()
    decode = ph["decode_inputs"] / world
    control = ph["weights"]                 # initial table + re-weighting + sample merges: replicated
    t1 = sum(ph.values())
    out = {
        "gpus": world, "iterations": len(rows), "checks": len(per_check),
        "single_gpu_phase_seconds": ph, "single_gpu_total_s": t1,
        "decode_s": decode, "control_replicated_s": control, "merges_s": t_merge, "p2p_s": t_p2p,
        "sets_moved": sets_moved, "gb_moved": bytes_moved / 1e9,
        "encodes_barrier_per_check_s": t_barrier, "encodes_one_check_lookahead_s": t_look,
        "encodes_perfect_balance_s": ph["encodes"] / world,
        "expected_total_barrier_s": decode + control + t_merge + t_p2p + t_barrier,
        "expected_total_lookahead_s": decode + control + t_merge + t_p2p + t_look,
        "encodes_implemented_schedule_s": t_impl, "sets_handed_over_for_encoding": moved_impl,
        "expected_total_implemented_s": decode + control + t_merge + t_p2p + t_impl,
        "max_rank_share_per_check": [max(l) / max(sum(l), 1e-12) for l in per_check],
    }
    out["speedup_barrier"] = t1 / out["expected_total_barrier_s"]
    out["speedup_lookahead"] = t1 / out["expected_total_lookahead_s"]
    out["speedup_implemented"] = t1 / out["expected_total_implemented_s"]
    # the merges run on one rank each while the others go on: a rank's own share of them is about 1 / N
    out["expected_total_implemented_merges_spread_s"] = decode + control + t_merge / world + t_p2p + t_impl
    out["speedup_implemented_merges_spread"] = t1 / out["expected_total_implemented_merges_spread_s"]
    # ---- KSH_OWNED_WEIGHTS=sharded: the weighing by pair list + one all-gather per iteration.  The exchange
    # sits behind the iteration's full merge on the rank that runs it (same stream), so every rank waits for
    # every merge: they are charged in full.
    control_sharded = control / world + (len(rows) + 1) * args.allgather_us * 1e-6
    out["control_sharded_s"] = control_sharded
    out["allgather_us_assumed"] = args.allgather_us
    out["expected_total_sharded_control_s"] = decode + control_sharded + t_merge + t_p2p + t_impl
    out["speedup_sharded_control"] = t1 / out["expected_total_sharded_control_s"]
    # ---- the same with the full-size merges deferred to the end of their interval (KSH_OWNED_MERGES, the default
    # since round 4): the all-gather of an iteration no longer sits behind anybody's merge; an interval's merges run
    # side by side on their executors (transfers included: merges_deferred_s).  Every rank still runs the merge of
    # the iteration's two SAMPLES for itself (sample_merge_us each, assumed), in both control modes.
    sample_merges = len(rows) * args.sample_merge_us * 1e-6
    out["sample_merges_s"] = sample_merges
    out["merges_deferred_s"] = t_merge_deferred
    out["expected_total_replicated_deferred_s"] = decode + control + sample_merges + t_merge_deferred + t_impl
    out["speedup_replicated_deferred"] = t1 / out["expected_total_replicated_deferred_s"]
    out["expected_total_sharded_deferred_s"] = decode + control_sharded + sample_merges + t_merge_deferred + t_impl
    out["speedup_sharded_deferred"] = t1 / out["expected_total_sharded_deferred_s"]
    out["note"] = "arithmetic over the measured single-GPU phases and merge sequence: modelled, not measured"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
