"""GPU parity tests for the set-of-sets loop (lib/core/kmer_set_set.h:109-454) through
the C ABI: merge trace, convergence checkpoints, DAG, every node's SPSS strings and
Get(i) must equal the oracle's on the same seeded family."""
import numpy as np
import pytest

import oracle_lib as ol
from kmersets import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    c = capi.Context(0)
    yield c
    c.close()


def build_both(ctx, k, n, kb, n_sets, size, seed, max_iterations=-1):
    sets = synth.phylogeny_sets(k, n_sets, size, seed=seed)
    osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
    ocompacts = [s.compact() for s in osets]
    ids = synth.sample_bucket_ids(n, seed=seed + 1)
    okss = ol.KmerSetSet(ocompacts, ids, max_iterations=max_iterations)
    g = capi.geom(k, n)
    dcompacts = [capi.DeviceSpss.from_strings(g, c.strings(), ctx.device) for c in ocompacts]
    dkss = capi.DeviceKmerSetSet(ctx, dcompacts, ids, max_iterations=max_iterations)
    return sets, osets, okss, dkss


def compare(sets, osets, okss, dkss):
    n0 = len(sets)
    assert np.array_equal(dkss.initial_weights(), okss.initial_weights(n0))
    it, cp, imp = dkss.trace()
    assert np.array_equal(it, okss.iterations())
    ocp, oimp = okss.checkpoints()
    assert np.array_equal(cp, ocp)
    assert np.array_equal(imp, oimp)            # same float arithmetic, bit for bit
    assert dkss.size() == okss.size()
    assert dkss.meta() == okss.meta()
    st = dkss.stats()
    assert st["initial_total_size"] == okss.stat(0) and st["final_total_size"] == okss.stat(1)
    assert st["initial_spss_weight"] == okss.stat(2) and st["n_processed"] == okss.stat(3)
    for i in range(okss.size()):
        node = okss.node(i)
        assert dkss.node_strings(i) == node.strings(), "node %d" % i
        assert dkss.node_size(i) == node.size()
    for i in range(n0):
        got = dkss.get_kmers(i)
        assert np.array_equal(got, sets[i])      # test/kmer_set_set.cc:30-34
        assert np.array_equal(got, okss.get(i).kmers())
    return len(it)


@pytest.mark.parametrize("case", [(9, 10, 1, 6, 3000, 11), (15, 14, 2, 8, 20000, 3),
                                  (23, 14, 4, 8, 30000, 5), (31, 14, 8, 4, 20000, 7)])
def test_loop_vs_oracle(ctx, case):
    k, n, kb, n_sets, size, seed = case
    sets, osets, okss, dkss = build_both(ctx, k, n, kb, n_sets, size, seed)
    merges = compare(sets, osets, okss, dkss)
    if k >= 15:
        assert merges > 0, "a correlated family must merge"
    dkss.close()


def test_loop_vs_oracle_mid_size(ctx):
    """6 sets of 2 x 10^5 k-mers (k = 23): arrays of 10^5 .. 10^6 elements inside the encode and
    the plan, i.e. the mid-size scan paths (single-launch chained scan, packed two-counter sums)
    that the 2 x 10^4 cases above do not reach; everything against the oracle as there."""
    sets, osets, okss, dkss = build_both(ctx, 23, 14, 4, 6, 200000, 17)
    assert compare(sets, osets, okss, dkss) > 0
    dkss.close()


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_loop_lanes(gpu, lanes):
    """The stale nodes of a check and the inputs' decodes on 1, 2 and 4 streams at once (ksh_ctx_set_lanes): the
    same trace, checkpoints, DAG, node strings and Get(i) as the oracle's whichever lane encoded a node
    (kmer_set_set.h:138-153,287,345-360: the reference runs these on its pool too), twice on the same context
    (the helper lanes and their scratch are reused), and the timers add up: the union of the probe stage's
    spans is no longer than their sum over the lanes."""
    c = capi.Context(0)
    c.set_lanes(lanes)
    try:
        for rep in range(2):
            c.enable_timing(1)
            c.timing_reset()
            sets, osets, okss, dkss = build_both(c, 23, 14, 4, 12, 60000, 21 + rep)
            assert compare(sets, osets, okss, dkss) > 0
            (ms, launches), wall = c.timing_read(3), c.timing_wall(3)
            assert launches == dkss.stats()["n_encodes"] and 0 < wall <= ms * 1.001 + 0.05
            if lanes == 1:
                assert abs(wall - ms) <= 0.02 * ms + 0.05
            c.enable_timing(False)
            dkss.close()
    finally:
        c.close()


def test_loop_lanes_oversize_buckets(gpu):
    """(19, 10) sets of 1.7 x 10^7 k-mers: 16 600 keys per bucket, more than the bucket sort's LDS window holds, so every
    decode takes a scratch copy the size of the set's keys from the pool and gives it back with the sort still
    queued -- fine on one stream, and the block that faulted the GPU when a job of another lane received it as its
    RESULT (tools/scale_sweep.py --seed 11, case 53).  Four inputs on three lanes, three builds:
    every Get(i) == input i by Size and XOR Hash, the nodes' sizes add up."""
    import torch
    from kmersets import synth_torch

    k, n = 19, 10
    g = capi.geom(k, n)
    c = capi.Context(0)
    c.set_lanes(3)
    try:
        fam = synth_torch.phylogeny_sets(k, 4, 17_000_001, 472687, c.device, rate=0.02)
        sets = [synth_torch.device_set(g, f) for f in fam]
        del fam
        compacts = [c.spss_encode(d, mode=0) for d in sets]
        want = [(d.n_keys, c.set_hash(d)) for d in sets]
        ids = synth.sample_bucket_ids(n, seed=5)
        for _ in range(3):
            kss = capi.DeviceKmerSetSet(c, compacts, ids)
            assert [kss.get_size_and_hash(i) for i in range(4)] == want
            assert sum(kss.node_size(i) for i in range(kss.size())) == kss.stats()["final_total_size"]
            kss.close()
        torch.cuda.synchronize()
    finally:
        c.close()


def test_loop_repeat_rich_family(ctx):
    """The whole loop on a repeat-rich family (bench_loop.py --repeats): every node's strings, the trace and the
    checkpoints against the oracle where the encodes are the expensive, branching kind."""
    k, n, kb = 23, 14, 4
    sets = synth.phylogeny_sets(k, 6, 120_000, seed=37, rate=0.003, repeats=(500, 6))
    osets = [ol.Set.from_kmers(k, n, kb, s) for s in sets]
    ocompacts = [s.compact() for s in osets]
    ids = synth.sample_bucket_ids(n, seed=38)
    okss = ol.KmerSetSet(ocompacts, ids)
    dcompacts = [capi.DeviceSpss.from_strings(capi.geom(k, n), c.strings(), ctx.device) for c in ocompacts]
    dkss = capi.DeviceKmerSetSet(ctx, dcompacts, ids)
    assert compare(sets, osets, okss, dkss) > 0
    dkss.close()


def test_loop_truncated_and_no_merge(ctx):
    k, n, kb = 23, 14, 4
    sets, osets, okss, dkss = build_both(ctx, k, n, kb, 6, 20000, 9, max_iterations=2)
    assert compare(sets, osets, okss, dkss) == 2
    dkss.close()
    # independent random sets share nothing: the loop stops at weight == 0 (kmer_set_set.h:319-322)
    g = capi.geom(k, n)
    ids = synth.sample_bucket_ids(n, seed=3)
    a, b = synth.uniform_pair(k, 20000, 0.0, seed=5)
    comp = [capi.DeviceSpss.from_strings(g, ol.Set.from_kmers(k, n, kb, s).spss(), ctx.device) for s in (a, b)]
    d = capi.DeviceKmerSetSet(ctx, comp, ids)
    assert d.size() == 2 and d.trace()[0].shape[0] == 0
    d.close()


def test_set_union(ctx):
    k, n = 23, 14
    sets = synth.phylogeny_sets(k, 2, 50000, seed=4)
    g = capi.geom(k, n)
    a = capi.DeviceSet.from_kmers(g, sets[0], ctx.device)
    b = capi.DeviceSet.from_kmers(g, sets[1], ctx.device)
    u = ctx.set_union(a, b)
    want = np.union1d(sets[0], sets[1])
    assert u.n_keys == want.size and np.array_equal(u.kmers(), want)
    off, keys = u.to_numpy()
    w_off, w_keys = synth.to_bucketed(want, k, n, 4)
    assert np.array_equal(off, w_off) and np.array_equal(keys, w_keys)
    e = capi.DeviceSet.from_kmers(g, np.zeros(0, dtype=np.uint64), ctx.device)
    assert np.array_equal(ctx.set_union(e, a).kmers(), sets[0])
    assert np.array_equal(ctx.set_union(a, a).kmers(), sets[0])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build(gpu, world):
    """ksh_kss_build_sharded with `world` processes (all on this GPU, gloo for the exchange): every
    rank reproduces the oracle's trace, checkpoints, DAG and sets; each node's SPSS is held by,
    and equal to the oracle's on, exactly one rank (tests/dist_kss_worker.py does the checks)."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29541 + world),
           os.path.join(here, "dist_kss_worker.py"), "15", "14", "2", "8", "20000", "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["iterations"] > 0
    assert sum(res["encodes_per_rank"]) > 0 and min(res["encodes_per_rank"]) > 0   # the encodes were shared


@pytest.mark.parametrize("world,shape,layout", [(2, ("15", "14", "2", "8", "20000", "3"), "block"),
                                                (3, ("23", "14", "4", "8", "30000", "5"), "striped"),
                                                (2, ("31", "14", "8", "6", "20000", "7"), "striped")])
def test_owned_build(gpu, world, shape, layout):
    """ksh_kss_build_owned with `world` processes (all on this GPU; gloo through host memory as the
    transport): every input decoded by its owner only, the control loop replicated on the all-gathered
    samples, merges on the owner of j with k's keys sent across.  Trace, checkpoints, DAG, sizes ==
    the oracle's on every rank; every node's set and SPSS == the oracle's on the one rank that owns it
    and unreadable elsewhere (tests/dist_owned_worker.py does the checks).  "striped" puts siblings on
    different ranks, so the merges do pull sets across."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29551 + world + int(shape[0])),
           os.path.join(here, "dist_owned_worker.py"), *shape, layout]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["ok"] and res["iterations"] > 0
    assert sum(res["nodes_per_rank"]) == res["nodes"]
    assert res["bytes_sent"] == res["bytes_received"]
    if layout == "striped":
        assert sum(res["sets_sent_per_rank"]) > 0          # sets did travel
    assert sum(res["encodes_per_rank"]) > 0
    # the checks were deferred by one (the default); a loop that stops at a check has undone the
    # interval it ran in the meantime
    assert res["checks_deferred"] == res["checks"] > 0
    if shape[0] == "23":
        assert res["rollbacks"] == 1      # this family stops at a check: the interval after it was undone


@pytest.mark.parametrize("layout", ["striped", "block"])
def test_owned_build_rccl_two_gpus(gpu, layout):
    """The first thing to run on a box with more than one GPU: the owner-sharded build with ONE
    GPU PER RANK over the library's RCCL transport -- ncclAllGather on the context's stream, ungrouped ncclSend /
    ncclRecv pairs for the sets that travel, the second communicator for the deferred check exchanges --
    against the oracle, exactly as test_owned_build does over gloo.  Skips on a one-GPU box (where RCCL
    refuses two ranks on one device): nobody has to remember to run it the day a multi-GPU node appears."""
    import json
    import os
    import subprocess
    import sys

    import torch

    n_gpus = torch.cuda.device_count()
    if n_gpus < 2:
        pytest.skip("needs two GPUs (this box has %d): RCCL with more than one rank stays unverified" % n_gpus)
    world = min(n_gpus, 4)
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", KSH_OWNED_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29611 + (layout == "block")),
           os.path.join(here, "dist_owned_worker.py"), "23", "14", "4", "8", "30000", "5", layout]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["ok"] and res["transport"] == "rccl" and res["ranks_seen"] == world and res["iterations"] > 0
    assert sum(res["nodes_per_rank"]) == res["nodes"] and res["bytes_sent"] == res["bytes_received"]
    if layout == "striped":
        assert sum(res["sets_sent_per_rank"]) > 0
    assert res["checks_deferred"] == res["checks"] > 0


def test_owned_build_eight_ranks(gpu):
    """The owner-sharded build at the rank count it is designed for: 8 ranks, 64 sets of 2 x 10^5 k-mers
    (k = 23), block owners (owners[i] = i * 8 // 64) -- the deal of a check's encodes over eight ranks, the
    deferred checks and the roll-back, all against the oracle.  A one-GPU box admits six processes on its
    card, so the eight ranks are eight host threads of one process, each with its own context and
    stream, over an in-process transport (tests/owned_threads_worker.py)."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, os.path.join(here, "owned_threads_worker.py"), "23", "14", "4", "64", "200000", "21", "8",
           "block"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["ok"] and res["world"] == 8 and res["iterations"] >= 32
    assert sum(res["nodes_per_rank"]) == res["nodes"] and min(res["nodes_per_rank"]) > 0
    assert res["bytes_sent"] == res["bytes_received"] and sum(res["sets_sent_per_rank"]) > 0
    assert min(res["encodes_per_rank"]) > 0
    assert res["checks_deferred"] == res["checks"] > 0


@pytest.mark.parametrize("inject", ["1:4:150000", "2:1:150000", "0:7:150000"])
def test_owned_build_failure_reaches_every_rank(gpu, inject):
    """A rank whose large allocations start failing in the middle of the owner-sharded build (the injected
    stand-in for a GPU out of memory: in a merge, while a set arrives, during the decode of its inputs)
    must not leave the others waiting in an exchange: it goes on following the protocol with empty
    stand-ins, its status travels with the next all-gather, and EVERY rank's ksh_kss_build_owned returns
    an error -- the failed rank its own, the others "rank r failed".  3 rank threads, striped owners (the
    merges pull sets across ranks), tests/owned_threads_worker.py."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, KSH_FAIL_INJECT=inject)
    cmd = [sys.executable, os.path.join(here, "owned_threads_worker.py"), "23", "14", "4", "12", "60000", "5", "3",
           "striped", "expect_fail"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["raised"] == [True, True, True], res
    failed = int(inject.split(":")[0])
    assert "injected" in res["messages"][failed]
    for r_, m in enumerate(res["messages"]):
        if r_ != failed:
            assert "rank %d failed" % failed in m, res


@pytest.mark.parametrize("mode", ["sharded", "replicated", "sharded+inline-merges"])
def test_owned_build_control_modes(gpu, mode):
    """The two control modes and the two merge schedules of the owner-sharded build, each against the oracle (4 rank
    threads).  "sharded" (the default): the weight tables dealt out by pair list, one all-gather of per-pair
    int64 weights per iteration (lib/core/kmer_set_set.h:205-218,385-425; the collective north_star names), the
    full-size merges deferred to the end of their interval.  "replicated": every rank weighs its replica of the
    samples, no exchange.  "inline-merges": every merge in its iteration, as before round 4."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, KSH_OWNED_WEIGHTS=mode.split("+")[0])
    if "inline" in mode:
        env["KSH_OWNED_MERGES"] = "inline"
    cmd = [sys.executable, os.path.join(here, "owned_threads_worker.py"), "23", "14", "4", "16", "40000", "9", "4",
           "block"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["ok"] and res["iterations"] > 0
    if mode == "replicated":
        assert res["weight_gathers"] == 0
    else:
        assert res["weight_gathers"] >= res["iterations"] + 1  # the initial table + one per iteration (+ an undone interval's)


def test_owned_build_without_lookahead(gpu, monkeypatch):
    """KSH_OWNED_LOOKAHEAD=0: every check resolved on the spot -- the same result by the other route."""
    monkeypatch.setenv("KSH_OWNED_LOOKAHEAD", "0")
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29597",
           os.path.join(here, "dist_owned_worker.py"), "23", "14", "4", "8", "30000", "5", "striped"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert res["ok"] and res["checks_deferred"] == 0 and res["rollbacks"] == 0


def test_owned_build_rccl_one_rank(gpu):
    """The RCCL transport itself, as far as one GPU allows: a one-rank communicator (librccl opened at
    run time, ncclCommInitRank, the all-gathers on the context's stream) behind ksh_kss_build_owned ==
    ksh_kss_build on the same inputs."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from kmersets import capi, synth\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')\n"
        "os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "ctx = capi.Context(0); g = capi.geom(23, 14)\n"
        "sets = synth.phylogeny_sets(23, 6, 40000, seed=11); ids = synth.sample_bucket_ids(14, seed=12)\n"
        "comp = [ctx.spss_encode(capi.DeviceSet.from_kmers(g, s, ctx.device)) for s in sets]\n"
        "a = capi.DeviceKmerSetSet(ctx, comp, ids)\n"
        "b = capi.OwnedKmerSetSet(ctx, comp, ids, dist, torch.device('cuda', 0))\n"
        "assert b.comm.kind == 'rccl'\n"
        "assert all(np.array_equal(x, y) for x, y in zip(a.trace(), b.trace()))\n"
        "assert a.meta() == b.meta() and a.size() == b.size()\n"
        "sa, sb = a.stats(), b.stats()\n"
        "assert all(sa[k] == sb[k] for k in ('n_processed', 'final_spss_weight', 'packed_bytes', 'length_bytes', 'nodes'))\n"
        "assert all(a.node_strings(i) == b.node_strings(i) for i in range(a.size()))\n"
        "assert all(b.get_size_and_hash(i) == a.get_size_and_hash(i) for i in range(6))\n"
        "a.close(); b.close(); dist.destroy_process_group(); print('rccl one-rank ok')\n"
    ) % (os.path.join(here, "..", "kmer-sets-compression_amd"), here)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_loop_control_ahead(gpu):
    """KSH_KSS_LOOP=ahead (the control loop on the samples ahead of the sets, stale nodes that are merged
    again only weighed): the same trace, checkpoints, DAG and node strings as the oracle -- in a process of
    its own, the switch is read once."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from kmersets import capi, synth\n"
        "ctx = capi.Context(0)\n"
        "for (k, n, kb, n_sets, size, seed) in ((15, 14, 2, 10, 6000, 3), (23, 14, 4, 12, 20000, 5), (31, 14, 8, 6, 20000, 7)):\n"
        "    g = capi.geom(k, n)\n"
        "    sets = synth.phylogeny_sets(k, n_sets, size, seed=seed); ids = synth.sample_bucket_ids(n, seed=seed + 1)\n"
        "    oc = [ol.Set.from_kmers(k, n, kb, s).compact() for s in sets]\n"
        "    o = ol.KmerSetSet(oc, ids)\n"
        "    d = capi.DeviceKmerSetSet(ctx, [capi.DeviceSpss.from_strings(g, c.strings(), ctx.device) for c in oc], ids)\n"
        "    it, cp, imp = d.trace(); ocp, oimp = o.checkpoints()\n"
        "    assert np.array_equal(it, o.iterations()) and np.array_equal(cp, ocp) and np.array_equal(imp, oimp)\n"
        "    assert d.meta() == o.meta() and d.size() == o.size()\n"
        "    assert all(d.node_strings(i) == o.node(i).strings() for i in range(o.size()))\n"
        "    st = d.stats(); assert st['n_processed'] == o.stat(3)\n"
        "    print('weighed', st['n_weighed'], 'of', st['n_encodes'])\n"
        "    d.close()\n"
        "print('control ahead ok')\n"
    ) % (os.path.join(here, "..", "kmer-sets-compression_amd"), here)
    env = dict(os.environ, KSH_KSS_LOOP="ahead")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "control ahead ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_loop_reweigh_full(gpu):
    """KSH_REWEIGH=full: all three weight families of an iteration weighed again, as the reference does
    (kmer_set_set.h:385-425), instead of the default (the new node's family weighed, the two remainders'
    by subtraction: |(j \\ n) & l| = |j & l| - |n & l|) -- the same trace, checkpoints and nodes as the oracle
    either way; the default route is what every other test of this file runs."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from kmersets import capi, synth\n"
        "ctx = capi.Context(0)\n"
        "for (k, n, kb, n_sets, size, seed) in ((15, 14, 2, 10, 6000, 3), (23, 14, 4, 12, 20000, 5)):\n"
        "    g = capi.geom(k, n)\n"
        "    sets = synth.phylogeny_sets(k, n_sets, size, seed=seed); ids = synth.sample_bucket_ids(n, seed=seed + 1)\n"
        "    oc = [ol.Set.from_kmers(k, n, kb, s).compact() for s in sets]\n"
        "    o = ol.KmerSetSet(oc, ids)\n"
        "    d = capi.DeviceKmerSetSet(ctx, [capi.DeviceSpss.from_strings(g, c.strings(), ctx.device) for c in oc], ids)\n"
        "    it, cp, imp = d.trace(); ocp, oimp = o.checkpoints()\n"
        "    assert np.array_equal(it, o.iterations()) and np.array_equal(cp, ocp) and np.array_equal(imp, oimp)\n"
        "    assert d.meta() == o.meta() and d.size() == o.size()\n"
        "    assert all(d.node_strings(i) == o.node(i).strings() for i in range(o.size()))\n"
        "    d.close()\n"
        "print('reweigh full ok')\n"
    ) % (os.path.join(here, "..", "kmer-sets-compression_amd"), here)
    env = dict(os.environ, KSH_REWEIGH="full")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "reweigh full ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
