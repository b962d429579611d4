#!/bin/bash
# Round evidence for the whole KmerSetSet loop (bench_loop.py): the configurations quoted in
# DESIGN.md 5 (each with --verify: after the timed build, Size and XOR Hash of Get(i) against the
# decoded input for every i, the reference's --check), then the 16 x 1e8 run once more under
# rocprofv3 --kernel-trace --stats.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/loop_evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; python3 $R/bench_loop.py --verify "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; cut -c1-220 $O/$name.json; echo; }
run r01_loop_16x1e7 --sets 16 --size 1e7 --cpu-iterations 3 --cpu-size 1e6
run r01_loop_config3_16x1e8 --sets 16 --size 1e8
run r01_loop_k31_16x1e7 --k 31 --sets 16 --size 1e7
run r01_loop_k31_8x5e8 --k 31 --sets 8 --size 5e8
run r01_loop_64x1e8_one_gpu --sets 64 --size 1e8
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o loop16 -- python3 $R/bench_loop.py --sets 16 --size 1e8 > $O/loop16_under_rocprof.json 2> $O/loop16_rocprof.err
