#!/bin/bash
# Round evidence for the contract bench: the bench line, the same command under
# rocprofv3 --kernel-trace --stats, and two PMC passes (FETCH_SIZE, WRITE_SIZE) kept apart from
# any other tracing, as MI355X_MICROARCH.md prescribes.  The stats pass runs the bench without its
# untimed KmerSetSet extra (--no-spss), so that every launch of the merge kernels in the summary is
# a 6-pair launch of the timed kind.  Outputs under gpurun_out/evidence/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 $R/bench.py --no-spss > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-spss > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-spss > $O/pmc_write.json 2> $O/pmc_write.err
find $O -name "*.csv" | head -20
