#!/bin/bash
# A/B of the neighbour-probe stage on one 1e8-k-mer encode: tools/encode_prof.sh under the knobs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "${@:-staged}"; do
  set -- $v
  echo "== $v"
  $R/tools/encode_prof.sh "$@" 2>&1 | grep -E "k_adj|k_rc|k_fwd|k_adjacency|k_fine" || true
done
