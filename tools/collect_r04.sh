#!/bin/bash
# Round-4 evidence for the contract bench (the whole KmerSetSet constructor on 64 x 1e8, k = 23):
#   1. the bench line (default lanes; the roofline leg inside it is a one-stream build)  -> $O/${TAG}_bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats at ONE lane (per-kernel averages that the
#      roofline leg's HIP-event figure must agree with) and at the default lanes (how the streams overlap:
#      tools/lane_overlap.py, tools/gpu_gaps.py)
#   3. FETCH_SIZE and WRITE_SIZE passes (separate runs, --kernel-trace only, one lane) on the 16 x 1e8 loop,
#      reduced per kernel by tools/pmc_kernel.py
#   4. the N-GPU time model over the dumped merge sequence (tools/owned_schedule.py)
# usage: tools/collect_r04.sh TAG [bench args...]
set -e -o pipefail
TAG=${1:-r04}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --dump-trace $O/${TAG}_trace.json "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cut -c1-700 $O/${TAG}_bench.json; echo
for L in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_l$L -o stats -- python3 $R/bench.py --steps 1 --warmup 1 --lanes $L --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_bench_under_rocprof_l$L.json 2> $O/${TAG}_rocprof_stats_l$L.err
  T="$(find $O/${TAG}_stats_l$L -name '*kernel_trace.csv' | head -1)"
  python3 $R/tools/gpu_gaps.py "$T" > $O/${TAG}_gaps_l$L.txt
  python3 $R/tools/lane_overlap.py "$T" -2 > $O/${TAG}_overlap_l$L.txt   # (the timed build; the last one is the one-stream roofline leg)
  cp "$(find $O/${TAG}_stats_l$L -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats_l$L.csv
  rm -rf $O/${TAG}_stats_l$L
  echo stats l$L done
done
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o fetch -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_fetch.json 2> $O/${TAG}_pmc_fetch.err
echo fetch done
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o write -- python3 $R/bench.py --sets 16 --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-verify --no-pair-merge > $O/${TAG}_pmc_write.json 2> $O/${TAG}_pmc_write.err
echo write done
F=$(find $O/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/${TAG}_pmc_write -name "*counter_collection.csv" | head -1)
STAGE=k_rc_hist,k_rc_columns,k_rc_scatter,k_rc_scatter_l1,k_rc_scatter_l2,k_rc_bounds,k_adj_rc,k_adj_rc1,k_tgt_bounds,k_tgt_split,k_tgt_subcuts,k_adj_fwd_targets
python3 $R/tools/pmc_kernel.py $F $W $STAGE $O/${TAG}_pmc_adjacency_stage.json --units-from k_link_cut
python3 $R/tools/pmc_kernel.py $F $W k_rank_walk,k_rank_heads,k_rank_unset,k_ruler_jump,k_l2_walk,k_l2_jump,k_l2_resolve $O/${TAG}_pmc_ranking_walks.json --units-from k_link_cut
python3 $R/tools/pmc_kernel.py $F $W k_emit_log_rulers,k_emit_log_heads $O/${TAG}_pmc_emit_from_logs.json --units-from k_link_cut
python3 $R/tools/pmc_kernel.py $F $W k_decode,k_decode_l1,k_decode_l2,k_hist_columns,k_bucket_sort $O/${TAG}_pmc_decode.json --units-from k_decode_l2 --units-scale 8 || true
for n in 2 4 8; do python3 $R/tools/owned_schedule.py $O/${TAG}_trace.json --gpus $n > $O/${TAG}_owned_schedule_model_${n}gpu.json; done
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
ls -la $O | tail -30
