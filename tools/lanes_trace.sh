#!/bin/bash
# Kernel traces of one timed build at 1 lane and at N lanes, reduced by tools/lane_overlap.py.
# usage: tools/lanes_trace.sh TAG [lanes...]
set -e -o pipefail
TAG=${1:-lanes}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "${@:-1 3}"; do
  rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_l${L} -o t -- python3 $R/bench.py --steps 1 --warmup 1 --lanes $L --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_l${L}.json 2> $O/${TAG}_l${L}.err
  python3 $R/tools/lane_overlap.py "$(find $O/${TAG}_l${L} -name '*kernel_trace.csv' | head -1)" -2 > $O/${TAG}_l${L}_overlap.txt
  rm -rf $O/${TAG}_l${L}
  head -3 $O/${TAG}_l${L}_overlap.txt
done
