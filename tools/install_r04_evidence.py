#!/usr/bin/env python3
"""Copies what tools/collect_r04.sh TAG left under gpurun_out/r04/ into profiles/ under the names DESIGN.md
cites (profiles/r04_*), and rewrites profiles/pmc_adjacency.json (what bench.py reads for roofline.traffic)
from the same run.  gpurun_out/ is scratch; profiles/ is what is tracked.  Refuses a bench line whose CPU
baseline was timed on the portable fall-back build of the oracle (cpu_baseline.oracle_native false).

    python tools/install_r04_evidence.py TAG
"""
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "r04")
    dst = os.path.join(ROOT, "profiles")
    b = json.loads(open(os.path.join(src, "%s_bench.json" % tag)).read().strip().splitlines()[-1])
    cb = b.get("cpu_baseline")
    if cb is not None and not cb.get("oracle_native", False):
        raise SystemExit("refused: the CPU baseline of this line was timed on the portable fall-back build of the oracle")
    names = {
        "bench.json": "r04_bench_64x1e8.json",
        "kernel_stats_l1.csv": "r04_kernel_stats_64x1e8_one_stream.csv",
        "kernel_stats_l4.csv": "r04_kernel_stats_64x1e8_four_lanes.csv",
        "gaps_l1.txt": "r04_gpu_idle_gaps_one_stream.txt",
        "gaps_l4.txt": "r04_gpu_idle_gaps_four_lanes.txt",
        "overlap_l1.txt": "r04_lane_overlap_one_stream.txt",
        "overlap_l4.txt": "r04_lane_overlap_four_lanes.txt",
        "trace.json": "r04_trace_64x1e8.json",
        "pmc_ranking_walks.json": "r04_pmc_ranking_walks_16x1e8.json",
        "pmc_emit_from_logs.json": "r04_pmc_emit_from_logs_16x1e8.json",
        "pmc_decode.json": "r04_pmc_decode_16x1e8.json",
    }
    for n in (2, 4, 8):
        names["owned_schedule_model_%dgpu.json" % n] = "r04_owned_schedule_model_%dgpu.json" % n
    for a, c in names.items():
        shutil.copyfile(os.path.join(src, "%s_%s" % (tag, a)), os.path.join(dst, c))
    d = json.load(open(os.path.join(src, "%s_pmc_adjacency_stage.json" % tag)))
    d["round"] = 4
    d["k"] = 23
    d["workload"] = "bench.py --sets 16 --lanes 1 (16 x 1e8, k = 23), one build on one stream, every dispatch of the stage"
    for name in ("pmc_adjacency.json", "r04_pmc_adjacency_stage_16x1e8.json"):
        json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    r = b["roofline"]
    print("value %.1f Mk-mers/s, %.1f ms per build; probe %.4f ns/k-mer (one stream), frac %.4f; ranking %.4f, emit %.4f; "
          "pmc %.1f B/k-mer; timed region: %s"
          % (b["value"], b["ms_per_step"], r["ns_per_kmer"], r["frac"], r["other_kernels"]["ranking_walks"]["ns_per_kmer"],
             r["other_kernels"]["emit_walks"]["ns_per_kmer"], d["bytes_per_kmer"], r["timed_region"]))


if __name__ == "__main__":
    main()
