"""Per-phase cycle breakdown of k_adj_fwd_targets' workgroups (s_memtime marks of the first thread of every
workgroup; debugging aid) on a genome set or on the difference of two: argv[1] = genome | difference, argv[2] = size.
Needs the trace build: make -C kmer-sets-compression_amd/csrc BUILD=build_trace OUT=libkmersets_hip_trace.so EXTRA=-DKSH_TRACE
and KSH_LIB pointing at it.  Not part of the product path or of the test suite."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "genome"
size = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
ctx = capi.Context(0)
g = capi.geom(23, 14)
fam = synth_torch.phylogeny_sets(23, 2, size, 4, ctx.device)
a, b = (synth_torch.device_set(g, x) for x in fam)
del fam
s = a
if which == "difference":
    s = ctx.pair_algebra(a, b)[1]
lib = capi.lib()
lib.ksh_debug_set_probe_trace.argtypes = [C.c_void_p, C.c_longlong]
lib.ksh_debug_set_probe_trace.restype = C.c_int
for _ in range(2):
    ctx.spss_encode(s, mode=0)
rows = max((s.n_keys + 511) // 512, 1 << 14) + 8
buf = torch.zeros(4 * rows * 16, dtype=torch.int64, device=ctx.device)
assert lib.ksh_debug_set_probe_trace(C.c_void_p(buf.data_ptr()), rows) == 0
ctx.spss_encode(s, mode=0)
torch.cuda.synchronize()
assert lib.ksh_debug_set_probe_trace(C.c_void_p(0), 0) == 0
t = buf.cpu().view(4, rows, 16)[0]
f = t[t[:, 6] != 0]
names = ["cuts arrive (1st round trip)", "tables + window loads -> LDS (2nd round trip)", "barrier", "stream (first wave)",
         "barrier (the other waves' streams)", "marks + store"]
print("%s: n %d, %d workgroups, cycles first mark -> last: mean %.0f p50 %.0f p90 %.0f max %.0f" % (
    which, s.n_keys, f.shape[0], *(float(x) for x in ((f[:, 6] - f[:, 0]).double().mean(), (f[:, 6] - f[:, 0]).double().median(),
                                                       (f[:, 6] - f[:, 0]).double().quantile(0.9), (f[:, 6] - f[:, 0]).double().max()))))
for ph in range(6):
    d = (f[:, ph + 1] - f[:, ph]).double()
    print("   %-50s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (names[ph], d.mean(), d.median(), d.quantile(0.9), d.max()))
q = f[:, 7].double()
print("streamed k-mers per workgroup: mean %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    q.mean(), q.quantile(0.1), q.median(), q.quantile(0.9), q.quantile(0.99), q.max()))
span = (f[:, 6].max() - f[:, 0].min()).item()
print("kernel span %.0f cycles (100 MHz clock: %.1f us)" % (span, span / 100.0))
# when the workgroups started and ended, by tenth of the grid
n = f.shape[0]
t0 = f[:, 0].min()
for d in range(10):
    part = f[d * n // 10:(d + 1) * n // 10]
    print("   workgroups %3d%%..: start %8.0f .. %8.0f, end max %8.0f, mean stream %6.0f, mean cycles %6.0f" % (
        10 * d, (part[:, 0].min() - t0).item(), (part[:, 0].max() - t0).item(), (part[:, 6].max() - t0).item(),
        part[:, 7].double().mean().item(), (part[:, 6] - part[:, 0]).double().mean().item()))
