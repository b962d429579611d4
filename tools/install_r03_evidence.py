#!/usr/bin/env python3
"""Copies what tools/collect_r03.sh TAG left under gpurun_out/r03/ into profiles/ under the names DESIGN.md
cites (profiles/r03_*), and rewrites profiles/pmc_adjacency.json (what bench.py reads for roofline.traffic)
from the same run.  gpurun_out/ is scratch; profiles/ is what is tracked.

    python tools/install_r03_evidence.py f5
"""
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "r03")
    dst = os.path.join(ROOT, "profiles")
    names = {
        "bench.json": "r03_bench_64x1e8.json",
        "kernel_stats.csv": "r03_kernel_stats_64x1e8.csv",
        "gaps.txt": "r03_gpu_idle_gaps.txt",
        "trace.json": "r03_trace_64x1e8.json",
        "pmc_ranking_walks.json": "r03_pmc_ranking_walks_16x1e8.json",
        "pmc_emit_from_logs.json": "r03_pmc_emit_from_logs_16x1e8.json",
        "pmc_links_and_ends.json": "r03_pmc_links_and_ends_16x1e8.json",
        "pmc_decode.json": "r03_pmc_decode_16x1e8.json",
    }
    for n in (2, 4, 8):
        names["owned_schedule_model_%dgpu.json" % n] = "r03_owned_schedule_model_%dgpu.json" % n
    for a, b in names.items():
        shutil.copyfile(os.path.join(src, "%s_%s" % (tag, a)), os.path.join(dst, b))
    d = json.load(open(os.path.join(src, "%s_pmc_adjacency_stage.json" % tag)))
    d["round"] = 3
    d["k"] = 23
    d["workload"] = "bench.py --sets 16 (16 x 1e8, k = 23), one build, every dispatch of the stage"
    d["note"] = ("round 2's file of this name left out k_rc_scatter_l1 / k_rc_scatter_l2 (the kernel list matched names "
                 "exactly and held only k_rc_scatter): its 54.4 B per k-mer was the stage without its scatter, 29 B per "
                 "k-mer by this round's counters")
    for name in ("pmc_adjacency.json", "r03_pmc_adjacency_stage_16x1e8.json"):
        json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    b = json.loads(open(os.path.join(dst, "r03_bench_64x1e8.json")).read().strip().splitlines()[-1])
    r = b["roofline"]
    print("value %.1f Mk-mers/s, %.1f ms per build; probe %.4f ns/k-mer, frac %.4f; ranking %.4f, emit %.4f; pmc %.1f B/k-mer"
          % (b["value"], b["ms_per_step"], r["ns_per_kmer"], r["frac"], r["other_kernels"]["ranking_walks"]["ns_per_kmer"],
             r["other_kernels"]["emit_walks"]["ns_per_kmer"], d["bytes_per_kmer"]))


if __name__ == "__main__":
    main()
