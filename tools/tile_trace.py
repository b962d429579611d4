"""Per-phase latency breakdown of k_tile_merge (debugging aid).

Needs libkmersets_hip.so built with -DKSH_TRACE (make CXXFLAGS+=-DKSH_TRACE); prints, for the
count pass and the write pass of one config-2 pair, the mean shader cycles between the
marks placed in the kernel.  Not part of the product path or of the test suite.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402


def main():
    k, n_bits = 23, 14
    size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda:0")
    g = capi.geom(k, n_bits)
    ctx = capi.Context(0)
    sets = [synth_torch.device_set(g, km) for km in synth_torch.phylogeny_sets(k, 2, size, 7, dev)]
    lib = capi.lib()
    lib.ksh_debug_set_tile_trace.argtypes = [C.c_void_p]
    lib.ksh_debug_set_tile_trace.restype = C.c_int
    for _ in range(3):
        ctx.pair_algebra_batch([(sets[0], sets[1])])
    lib.ksh_debug_set_tile_stop.argtypes = [C.c_int]
    ctx.enable_timing(True)
    for stop in (0, 1):
        assert lib.ksh_debug_set_tile_stop(stop) == 0
        ctx.timing_reset()
        for _ in range(5):
            ctx.pair_algebra_batch([(sets[0], sets[1])])
        print("stop_after", stop, "write pass", ctx.timing_read(0), "count pass", ctx.timing_read(1))
    assert lib.ksh_debug_set_tile_stop(0) == 0
    n_slots = 1 << 17
    buf = torch.zeros(2 * n_slots * 8, dtype=torch.int64, device=dev)
    # gridDim.x is what the kernel indexes with; over-allocate and find the rows that were written
    assert lib.ksh_debug_set_tile_trace(C.c_void_p(buf.data_ptr())) == 0
    ctx.pair_algebra_batch([(sets[0], sets[1])])
    ctx.sync()
    assert lib.ksh_debug_set_tile_trace(C.c_void_p(0)) == 0
    t = buf.cpu().view(-1, 8)
    rows = t[(t[:, 1] != 0)]
    names = ["desc/total loads, first key loads issued", "keys -> LDS (waits for the loads) + barrier",
             "split search / read", "merge steps", "scan", "compaction", "copy out"]
    for label, sel, last in (("count", (rows[:, 5] != 0) & (rows[:, 6] == 0), 5), ("write", rows[:, 7] != 0, 7)):
        for which, sub in (("first tile of its wave", rows[:, 0] != 0), ("later tile of its wave", rows[:, 0] == 0)):
            r = rows[sel & sub]
            if r.shape[0] == 0:
                continue
            first = 0 if which.startswith("first") else 1
            print(label, "-", which, "- tiles", r.shape[0], "- cycles from first mark to last",
                  float((r[:, last] - r[:, first]).double().mean()))
            for ph in range(first, last):
                dlt = (r[:, ph + 1] - r[:, ph]).double()
                print("   %-45s mean %8.0f  p50 %8.0f  p90 %8.0f" % (names[ph], dlt.mean(), dlt.median(), dlt.quantile(0.9)))


if __name__ == "__main__":
    main()
