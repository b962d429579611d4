#!/bin/bash
# Round-4 evidence of the final tree: tools/collect_r04.sh TAG, the per-kernel averages of one encode (genome and
# difference sets of 10^8-k-mer siblings), and the other loop shapes quoted in DESIGN.md 5.0 (bench_loop.py --verify).
# usage: tools/collect_r04_final.sh TAG
set -o pipefail
TAG=${1:-fin}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export KSH_ROUND=r04
O=$R/gpurun_out/r04
mkdir -p $O
bash $R/tools/collect_r04.sh $TAG > $O/${TAG}_collect.log 2>&1 || { tail -20 $O/${TAG}_collect.log; exit 1; }
tail -3 $O/${TAG}_collect.log
{
  echo "One SPSS encode of a 10^8-k-mer set, per-kernel averages over 3 encodes (tools/encode_prof_set.sh, rocprofv3 --kernel-trace --stats; final tree of round 4)"
  for w in genome difference; do echo; echo "== $w"; bash $R/tools/encode_prof_set.sh ${TAG}_$w $w 2>&1 | grep -v "^W2026"; grep "^n " $O/encs_${TAG}_$w.log; done
} > $O/${TAG}_encode_per_kernel.txt
grep -E "^==|^sum|^n " $O/${TAG}_encode_per_kernel.txt
cd $R
run() { name=$1; shift; python3 $R/bench_loop.py --verify "$@" 2> $O/${TAG}_$name.err | grep '^{' | tail -1 > $O/${TAG}_$name.json; cut -c1-200 $O/${TAG}_$name.json; echo; }
run loop_k23_4x1e8 --k 23 --sets 4 --size 1e8
run loop_k31_4x1e8 --k 31 --sets 4 --size 1e8
run loop_k15_8x3e7 --k 15 --sets 8 --size 3e7
run loop_genome_8x1e8 --k 23 --sets 8 --size 1e8
run loop_repeats_8x1e8 --k 23 --sets 8 --size 1e8 --repeats 5000,100
run loop_k31_8x5e8 --k 31 --sets 8 --size 5e8
