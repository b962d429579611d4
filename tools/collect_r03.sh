#!/bin/bash
# Round-3 evidence for the contract bench (the whole KmerSetSet constructor on 64 x 1e8, k = 23):
#   1. the bench line                                    -> gpurun_out/${KSH_ROUND:-r03}/$TAG_bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats (one warm-up + one timed build)
#   3. FETCH_SIZE and WRITE_SIZE passes (separate runs, --kernel-trace only) on the 16 x 1e8 loop,
#      reduced per kernel by tools/pmc_kernel.py
# usage: tools/collect_r02.sh TAG [bench args...]
set -e -o pipefail
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --dump-trace $O/${TAG}_trace.json "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cut -c1-600 $O/${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof_stats.err
python3 $R/tools/gpu_gaps.py "$(find $O/${TAG}_stats -name '*kernel_trace.csv' | head -1)" > $O/${TAG}_gaps.txt
echo stats done
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o fetch -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
echo fetch done
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o write -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_write.json 2> $O/${TAG}_pmc_write.err
echo write done
F=$(find $O/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/${TAG}_pmc_write -name "*counter_collection.csv" | head -1)
STAGE=k_rc_hist,k_rc_columns,k_rc_scatter,k_rc_scatter_l1,k_rc_scatter_l2,k_rc_bounds,k_adj_rc,k_fwd_bounds,k_adj_fwd_staged
if grep -q "k_adj_fwd_staged" $F; then
  python3 $R/tools/pmc_kernel.py $F $W $STAGE $O/${TAG}_pmc_adjacency_stage.json --units-from k_adj_fwd_staged
else
  python3 $R/tools/pmc_kernel.py $F $W k_adjacency $O/${TAG}_pmc_adjacency_stage.json
fi
# the walks have a thread per ruler or per end k-mer: their k-mers are those of the same encodes' forward probe
python3 $R/tools/pmc_kernel.py $F $W k_rank_walk,k_rank_heads,k_rank_unset,k_ruler_jump,k_l2_walk,k_l2_jump,k_l2_resolve $O/${TAG}_pmc_ranking_walks.json --units-from k_adj_fwd_staged
python3 $R/tools/pmc_kernel.py $F $W k_emit_log_rulers,k_emit_log_heads $O/${TAG}_pmc_emit_from_logs.json --units-from k_adj_fwd_staged
python3 $R/tools/pmc_kernel.py $F $W k_link_cut,k_end_counts,k_end_fill,k_choose_ends $O/${TAG}_pmc_links_and_ends.json --units-from k_adj_fwd_staged
# the decode of the inputs: its k-mers are the keys k_bucket_sort... has one workgroup per bucket; k_decode_l2 a thread per 8 keys: units from the encode of the same 16 inputs is not available here, so per-launch bytes only
python3 $R/tools/pmc_kernel.py $F $W k_decode,k_decode_l1,k_decode_l2,k_hist_columns,k_bucket_sort $O/${TAG}_pmc_decode.json --units-from k_decode_l2 --units-scale 8 || true
for n in 2 4 8; do python3 $R/tools/owned_schedule.py $O/${TAG}_trace.json --gpus $n > $O/${TAG}_owned_schedule_model_${n}gpu.json; done
# keep the merged-back payload small: the per-dispatch CSVs are tens of MB
S=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp $S $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
ls -la $O
