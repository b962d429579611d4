#!/bin/bash
# Per-kernel totals of one timed build of the contract bench under rocprofv3 --kernel-trace --stats.
# usage: tools/loop_stats.sh TAG [bench args...]
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-pair-merge "$@" > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof_stats.err
S=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp $S $O/${TAG}_kernel_stats.csv
python3 $R/tools/gpu_gaps.py "$(find $O/${TAG}_stats -name '*kernel_trace.csv' | head -1)" > $O/${TAG}_gaps.txt
rm -rf $O/${TAG}_stats
python3 - "$O/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ksh::" in n:
        rows.append((int(r["TotalDurationNs"]) / 1e6, n.split("ksh::")[1].split("(")[0][:44], int(r["Calls"])))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
for t, n, c in rows[:32]:
    print("%-44s calls %5d total %9.2f ms  %5.1f %%" % (n, c, t, 100 * t / tot))
print("all ksh kernels: %.1f ms" % tot)
PY
