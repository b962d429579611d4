"""Per-phase cycle breakdown of the two probe kernels' workgroups (k_adj_fwd_staged, k_adj_rc) on one
1e8-k-mer k=23 genome set: s_memtime marks of the first thread of every workgroup (debugging aid).

A pass of k_adj_rc that takes several batches (a group denser than its window: the A... buckets and the
densest C... ones of a canonical set, 35 % of the workgroups at 10^8) overwrites its marks batch by batch: the
phases printed are those of its LAST batch, and "plan" holds the batches before it.

Needs the trace build: make -C kmer-sets-compression_amd/csrc BUILD=build_trace OUT=libkmersets_hip_trace.so EXTRA=-DKSH_TRACE
and KSH_LIB pointing at it.  Not part of the product path or of the test suite.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmer-sets-compression_amd"))
from kmersets import capi, synth_torch  # noqa: E402


def main():
    size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    ctx = capi.Context(0)
    g = capi.geom(23, 14)
    a = synth_torch.device_set(g, synth_torch.phylogeny_sets(23, 1, size, 4, ctx.device)[0])
    lib = capi.lib()
    lib.ksh_debug_set_probe_trace.argtypes = [C.c_void_p, C.c_longlong]
    lib.ksh_debug_set_probe_trace.restype = C.c_int
    for _ in range(2):
        ctx.spss_encode(a, mode=0)
    rows = max((a.n_keys + 511) // 512, 1 << 14) + 8
    buf = torch.zeros(4 * rows * 16, dtype=torch.int64, device=ctx.device)
    assert lib.ksh_debug_set_probe_trace(C.c_void_p(buf.data_ptr()), rows) == 0
    ctx.spss_encode(a, mode=0)
    torch.cuda.synchronize()
    assert lib.ksh_debug_set_probe_trace(C.c_void_p(0), 0) == 0
    t = buf.cpu().view(4, rows, 16)
    fwd_names = ["bound record arrives (1st round trip)", "window loads arrive, LDS stores (2nd round trip)",
                 "barrier", "Next search", "Prev searches", "verdict + store"]
    rc_names = ["plan (record range + bounds arrive)", "stage: loads -> LDS", "barrier", "slice index + barrier",
                "record look-ups (this wave)", "barrier", "store marks"]
    if os.environ.get("KSH_FWD") == "staged":  # (the default forward kernel has its own tool: tools/tgt_trace.py)
        f = t[0][t[0][:, 6] != 0]
        print("k_adj_fwd_staged: %d workgroups, cycles first mark -> last: mean %.0f" % (f.shape[0], float((f[:, 6] - f[:, 0]).double().mean())))
        for ph in range(6):
            d = (f[:, ph + 1] - f[:, ph]).double()
            print("   %-50s mean %8.0f  p50 %8.0f  p90 %8.0f" % (fwd_names[ph], d.mean(), d.median(), d.quantile(0.9)))
    r = t[1][(t[1][:, 7] != 0) & (t[1][:, 14] != 0)]
    print("k_adj_rc: %d workgroups with both passes, cycles first mark -> last: mean %.0f" % (r.shape[0], float((r[:, 14] - r[:, 0]).double().mean())))
    for p in range(2):
        print("  pass %d" % p)
        prev = r[:, 0] if p == 0 else r[:, 7]
        for ph in range(7):
            cur = r[:, 1 + 7 * p + ph]
            d = (cur - prev).double()
            prev = cur
            print("   %-50s mean %8.0f  p50 %8.0f  p90 %8.0f" % (rc_names[ph], d.mean(), d.median(), d.quantile(0.9)))
    waves(t)


def waves(t):
    """k_adj_rc: when the waves of a workgroup started, and when they reached the planner's barrier (pass 0)."""
    ok = (t[1][:, 7] != 0) & (t[1][:, 14] != 0) & (t[2][:, 15] != 0)
    r, st, ar = t[1][ok], t[2][ok], t[3][ok]
    plan = (r[:, 1] - r[:, 0]).double()
    spread = (st.max(dim=1).values - st.min(dim=1).values).double()
    first_to_barrier = (ar.max(dim=1).values - st.min(dim=1).values).double()
    w0 = (ar[:, 0] - st[:, 0]).double()
    print("k_adj_rc waves (16-wave workgroups only): start spread mean %.0f p50 %.0f p90 %.0f; planner wave start -> barrier "
          "mean %.0f p90 %.0f; first start -> last arrival mean %.0f p90 %.0f" % (
              spread.mean(), spread.median(), spread.quantile(0.9), w0.mean(), w0.quantile(0.9),
              first_to_barrier.mean(), first_to_barrier.quantile(0.9)))
    slow = plan > 20000
    print("   workgroups with plan > 20k cycles: %d of %d; their start spread mean %.0f, planner wave mean %.0f; the others' %.0f, %.0f"
          % (int(slow.sum()), plan.numel(), spread[slow].mean(), w0[slow].mean(), spread[~slow].mean(), w0[~slow].mean()))
    last_wave = (ar - ar.min(dim=1, keepdim=True).values).double()
    print("   arrival at the barrier after the first arrival, by wave number (mean over slow workgroups):",
          [int(x) for x in last_wave[slow].mean(dim=0).tolist()])
    start_by_wave = (st - st.min(dim=1, keepdim=True).values).double()
    print("   start after the first start, by wave number (slow workgroups):", [int(x) for x in start_by_wave[slow].mean(dim=0).tolist()])
    idx = torch.nonzero(ok).flatten()
    print("   slow workgroups by index decile:", torch.histc(idx[slow].double(), bins=10, min=0, max=float(idx.max())).int().tolist())


if __name__ == "__main__":
    main()
