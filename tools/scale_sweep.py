"""Property checks of the HIP path at sizes the oracle does not reach in a test's time, over geometries, key
widths and set kinds: the parity suites compare with the oracle on small sets, the full-size tests run two
geometries -- this sweeps what lies between (the 2-byte-key fault of round 3 sat there: DESIGN.md 5.3).

Per case: a family of sets made on the device (genomes diverging at a random rate), then
  * SPSS encode -> decode == the set (size, XOR hash, and the keys themselves), canonical and as-is, fast and
    unitigs only;
  * pair algebra: |A & B| + |A \\ B| == |A|, the parts are disjoint by hash arithmetic, (A \\ B) | (A & B) == A;
  * the KmerSetSet loop over the family: every Get(i) == input i (size, hash), total sizes add up.
Not part of the pytest suites (minutes); prints a line per case.

    python tools/scale_sweep.py [--seed 1] [--max-size 6e7] [--seconds 600]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kmer-sets-compression_amd"))
from kmersets import capi, synth, synth_torch  # noqa: E402

GEOMS = [(15, 14), (17, 14), (19, 10), (21, 14), (23, 14), (23, 10), (27, 14), (31, 14), (31, 12)]  # (decode: N <= 14)


def keys_equal(a, b):
    ao, ak = a.to_numpy()
    bo, bk = b.to_numpy()
    return np.array_equal(ao, bo) and np.array_equal(ak, bk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-size", type=float, default=6e7)
    ap.add_argument("--seconds", type=float, default=600)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = capi.Context(0)
    t_end = time.time() + args.seconds
    n_cases = 0
    while time.time() < t_end:
        k, n = GEOMS[int(rng.integers(0, len(GEOMS)))]
        g = capi.geom(k, n)
        space = 4 ** k
        size = int(10 ** rng.uniform(5.5, np.log10(args.max_size)))
        size = int(min(size, space // 40)) | 1
        n_sets = int(rng.integers(2, 5))
        seed = int(rng.integers(0, 1 << 20))
        canonical = bool(rng.random() < 0.8)
        rate = float(10 ** rng.uniform(-3.5, -1.5))
        t0 = time.time()
        fam = synth_torch.phylogeny_sets(k, n_sets, size, seed, ctx.device, rate=rate)
        sets = [synth_torch.device_set(g, f) for f in fam]
        del fam
        # ---- encode -> decode
        compacts = []
        for i, d in enumerate(sets):
            mode = 0 if i else int(rng.integers(0, 2))  # the first one also as unitigs now and then
            sp = ctx.spss_encode(d, mode=mode, canonical=canonical)
            back = ctx.spss_decode(sp, canonical=canonical)
            assert back.n_keys == d.n_keys and ctx.set_hash(back) == ctx.set_hash(d), ("round trip", k, n, size, i, mode)
            if d.n_keys < 20_000_000:
                assert keys_equal(back, d), ("round trip keys", k, n, size, i)
            if mode == 0:
                compacts.append(sp)
            del back
        # ---- pair algebra
        a, b = sets[0], sets[1]
        inter, a_only, b_only = ctx.pair_algebra(a, b)
        assert inter.n_keys + a_only.n_keys == a.n_keys and inter.n_keys + b_only.n_keys == b.n_keys
        assert ctx.set_hash(inter) ^ ctx.set_hash(a_only) == ctx.set_hash(a)
        assert ctx.set_hash(inter) ^ ctx.set_hash(b_only) == ctx.set_hash(b)
        del inter, a_only, b_only
        # ---- the loop (canonical families; needs every input as an SPSS of mode 0)
        if canonical and len(compacts) == len(sets):
            ids = synth.sample_bucket_ids(n, seed=seed + 1)
            kss = capi.DeviceKmerSetSet(ctx, compacts, ids)
            for i, d in enumerate(sets):
                assert kss.get_size_and_hash(i) == (d.n_keys, ctx.set_hash(d)), ("loop", k, n, size, i)
            st = kss.stats()
            assert sum(kss.node_size(i) for i in range(kss.size())) == st["final_total_size"]
            kss.close()
        n_cases += 1
        print("ok %3d: k=%d N=%d key_bytes=%d sets=%d size=%d rate=%.4f canonical=%d  %.1f s" % (
            n_cases, k, n, g.key_bytes, n_sets, size, rate, canonical, time.time() - t0), flush=True)
        del sets, compacts
        torch.cuda.empty_cache()
    print("sweep ok: seed %d, %d cases" % (args.seed, n_cases))


if __name__ == "__main__":
    main()
