#!/bin/bash
# The PMC part of tools/collect_r04.sh alone (FETCH_SIZE and WRITE_SIZE passes on the 16 x 1e8 loop at one lane).
# usage: tools/collect_r04_pmc.sh TAG
set -e -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o fetch -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o write -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_write.json 2> $O/${TAG}_pmc_write.err
F=$(find $O/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/${TAG}_pmc_write -name "*counter_collection.csv" | head -1)
STAGE=$(grep '^STAGE=' $R/tools/collect_r04.sh | cut -d= -f2)
python3 $R/tools/pmc_kernel.py $F $W $STAGE $O/${TAG}_pmc_adjacency_stage.json --units-from k_link_cut
python3 $R/tools/pmc_kernel.py $F $W k_rank_walk,k_rank_heads,k_rank_unset,k_ruler_jump,k_l2_walk,k_l2_jump,k_l2_resolve $O/${TAG}_pmc_ranking_walks.json --units-from k_link_cut
python3 $R/tools/pmc_kernel.py $F $W k_emit_log_rulers,k_emit_log_heads $O/${TAG}_pmc_emit_from_logs.json --units-from k_link_cut
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
cut -c1-600 $O/${TAG}_pmc_adjacency_stage.json
