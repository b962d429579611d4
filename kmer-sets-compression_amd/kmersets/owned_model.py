"""Host-level model of the owner-sharded build (ksh_kss_build_owned, csrc/ksh_kss.hip): the same
schedule over plain numpy sets, with the exchanges through any torch.distributed backend.  It exists
so that the protocol -- who owns what, what is exchanged when, and the claim that the control loop
can run on the 2 % samples alone -- is checked with world_size 2 on CPU (tests/test_dist_cpu.py),
where the HIP kernels cannot run; the SPSS weight of a node comes from a caller-supplied function
(the tests pass the oracle's encoder).  The model resolves every check on the spot; the library defers a
check's exchange by one interval and rolls back on "stop", which changes when ranks wait, not what they
compute (tests/test_gpu_kmer_set_set.py runs both routes).

A set is a sorted uint64 array of 2K-bit k-mers; a sample is the sub-array whose bucket
(k-mer >> key_bits) is one of bucket_ids.
"""
import numpy as np


def block_owners(n_sets, world):
    return [i * world // n_sets for i in range(n_sets)]


def sample_of(kmers, key_bits, bucket_ids):
    return kmers[np.isin(kmers >> np.uint64(key_bits), np.asarray(bucket_ids, dtype=np.uint64))]


def merge(a, b):
    inter = np.intersect1d(a, b, assume_unique=True)
    return inter, np.setdiff1d(a, inter, assume_unique=True), np.setdiff1d(b, inter, assume_unique=True)


def build_owned(my_sets, owners, key_bits, bucket_ids, spss_weight, dist, max_iterations=-1):
    """my_sets[i]: input i (sorted uint64) on its owner, None elsewhere.  Returns a dict with the trace
    rows {j, k, weight, original_size, size_diff}, the checkpoints, children, owner of every node and
    this rank's node sets."""
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    n0 = len(owners)
    owner = list(owners)
    sets = [s if owner[i] == rank else None for i, s in enumerate(my_sets)]

    def gather_obj(x):
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    # sizes and samples of the inputs: from their owners, once
    mine = {i: (int(sets[i].size), sample_of(sets[i], key_bits, bucket_ids)) for i in range(n0) if owner[i] == rank}
    allv = gather_obj(mine)
    size = [allv[owner[i]][i][0] for i in range(n0)]
    samples = [allv[owner[i]][i][1] for i in range(n0)]
    weights = {(i, j): int(np.intersect1d(samples[i], samples[j], assume_unique=True).size)
               for i in range(n0) for j in range(i + 1, n0)}
    spss = [None] * n0          # SPSS weight per node, None = stale
    initial = gather_obj({i: spss_weight(sets[i]) for i in range(n0) if owner[i] == rank})
    for i in range(n0):
        spss[i] = initial[owner[i]][i]

    def total_weight():
        stale = [i for i in range(len(spss)) if spss[i] is None]
        got = gather_obj({i: (spss_weight(sets[i]), int(sets[i].size)) for i in stale if owner[i] == rank})
        for i in stale:
            spss[i], size[i] = got[owner[i]][i]
        return sum(spss)

    total = total_weight()
    interval = n0 // 8 + 1
    threshold = np.float32(0.1 * interval / n0)
    children, rows, checkpoints, executor = {}, [], [], []
    p2p = 0
    it = 0
    while True:
        if 0 <= max_iterations <= it:
            break
        if it > 0 and it % interval == 0:
            updated = total_weight()
            improvement = np.float32(np.float32(total - updated) / np.float32(total))
            stop = bool(improvement <= threshold)
            checkpoints.append((it, total, updated, stop))
            if stop:
                break
            total = updated
        best, j, k = 0, -1, -1
        for (a, b) in sorted(weights):
            if weights[(a, b)] > best:
                best, j, k = weights[(a, b)], a, b
        if best == 0:
            break
        n = len(owner)
        # control plane: the merge on the samples, every rank
        sn, sj, sk = merge(samples[j], samples[k])
        samples[j], samples[k] = sj, sk
        samples.append(sn)
        # data plane: the owner of j, k pulled in when it lives elsewhere
        ex, src = owner[j], owner[k]
        executor.append(ex)
        row = None
        sets.append(None)
        if rank == ex:
            if src != ex:
                cnt = torch.zeros(1, dtype=torch.int64)
                dist.recv(cnt, src)
                buf = torch.zeros(int(cnt.item()), dtype=torch.int64)
                if buf.numel():
                    dist.recv(buf, src)
                set_k = buf.numpy().view(np.uint64)
                p2p += 1
            else:
                set_k = sets[k]
            original = int(sets[j].size + set_k.size)
            fn, fj, fk = merge(sets[j], set_k)
            sets[j], sets[k], sets[n] = fj, fk, fn
            row = (j, k, best, original, int(fn.size + fj.size + fk.size) - original)
        elif rank == src:
            dist.send(torch.tensor([sets[k].size], dtype=torch.int64), ex)
            if sets[k].size:
                dist.send(torch.from_numpy(sets[k].view(np.int64).copy()), ex)
            sets[k] = None
        rows.append(row)
        for node in (j, k):
            spss[node] = None
        spss.append(None)
        size.append(0)
        owner[k] = ex
        owner.append(ex)
        children.setdefault(j, []).append(n)
        children.setdefault(k, []).append(n)
        for l in range(n):
            if l != j:
                weights[(min(j, l), max(j, l))] = int(np.intersect1d(samples[j], samples[l], assume_unique=True).size)
        for l in range(n):
            if l != k:
                weights[(min(k, l), max(k, l))] = int(np.intersect1d(samples[k], samples[l], assume_unique=True).size)
        for l in range(n):
            weights[(l, n)] = int(np.intersect1d(samples[l], samples[n], assume_unique=True).size)
        it += 1
    final = total_weight()
    all_rows = gather_obj(rows)
    trace = [all_rows[executor[t]][t] for t in range(len(executor))]
    return {"trace": trace, "checkpoints": checkpoints, "children": children, "owner": owner, "sets": sets,
            "final_spss_weight": final, "sizes": size, "sets_received": p2p}
